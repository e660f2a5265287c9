// multiwalker_c8.hip -- the MultiWalker kernels for up to 8 walkers per env, 8 lanes of a wavefront per env (8 envs per wavefront).
// One of the three capacity classes of multiwalker_impl.hpp (see there, and multiwalker.hip for how the C ABI picks one).
#define MW_CAPW 8
#define MW_NLANES 8
#include "multiwalker_impl.hpp"
