// multiwalker_c4.hip -- the MultiWalker kernels for up to 4 walkers per env, 4 lanes of a wavefront per env (16 envs per wavefront).
// One of the three capacity classes of multiwalker_impl.hpp (see there, and multiwalker.hip for how the C ABI picks one).
#define MW_CAPW 4
#define MW_NLANES 4
#include "multiwalker_impl.hpp"
