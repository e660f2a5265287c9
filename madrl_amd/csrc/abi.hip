// abi.hip -- version / error plumbing of the C ABI (include/madrl_hip.h).
#include "common.hpp"

namespace madrl {
char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace madrl

extern "C" {

int madrl_abi_version(void) { return MADRL_ABI_VERSION; }

const char *madrl_last_error(void) { return madrl::last_error_buf(); }

void madrl_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const madrl::u32x4 r = madrl::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

}  // extern "C"
