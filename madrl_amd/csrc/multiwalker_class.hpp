// multiwalker_class.hpp -- what one capacity class of the MultiWalker kernels (multiwalker_impl.hpp compiled with MW_CAPW / MW_NLANES)
// hands to the C ABI in multiwalker.hip: the entry points of include/madrl_hip.h with the handle as void *.
#pragma once
#include <stdint.h>

#include "../../include/madrl_hip.h"

namespace madrl {

struct MwClassApi {
    int cap_walkers;   // n_walkers this class has room for
    int lanes;         // lanes of a wavefront one env is spread over
    int (*obs_dim)(const madrl_multiwalker_config *, int32_t *);
    int (*state_bytes)(const madrl_multiwalker_config *, int64_t, uint64_t *);
    int (*create)(const madrl_multiwalker_config *, int64_t, int32_t, void *, void **);
    void (*destroy)(void *);
    int (*set_mode)(void *, int32_t, int32_t);
    int (*dims)(const void *, int32_t *, int32_t *);
    int (*record_bytes)(const void *, int32_t *, int32_t *);
    int (*reset)(void *, const uint8_t *, float *, void *);
    int (*reset_with)(void *, const uint8_t *, const double *, const double *, float *, void *);
    int (*get_state)(void *, float *, float *, float *, uint8_t *, float *, void *);
    int (*set_state)(void *, const float *, const float *, void *);
    int (*step)(void *, const float *, float *, float *, uint8_t *, void *);
    int (*get_bodies)(void *, float *, uint8_t *, float *, void *);
    int (*debug_read)(unsigned long long *, int *);              // measurement builds only (else NULL)
    int (*debug_read_acc)(unsigned long long *, int);
};

}  // namespace madrl

extern const madrl::MwClassApi madrl_mw_class_c4, madrl_mw_class_c8, madrl_mw_class_c10;
