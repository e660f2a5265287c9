// multiwalker_c10.hip -- the MultiWalker kernels for up to 10 walkers per env, 16 lanes of a wavefront per env (4 envs per wavefront).
// One of the three capacity classes of multiwalker_impl.hpp (see there, and multiwalker.hip for how the C ABI picks one).
#define MW_CAPW 10
#define MW_NLANES 16
#include "multiwalker_impl.hpp"
