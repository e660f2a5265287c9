// multiwalker.hip -- the MultiWalkerEnv entry points of include/madrl_hip.h.
//
// The kernels exist in three capacity classes (multiwalker_c4.hip, _c8.hip, _c10.hip: the same source, multiwalker_impl.hpp, with room for
// 4 / 8 / 10 walkers on 4 / 8 / 16 lanes of a wavefront per env): per-env records, LDS blocks and register budgets are sized by the class,
// so three walkers do not pay for ten.  The reference's curriculum runs n_walkers = 2 .. 10 (lessons/multiwalker/env.yaml:1-27,
// runners/curriculum.py:56-91; multi_walker.py:286-301 scales the package and the terrain with n_walkers).  A handle belongs to the
// smallest class that holds its n_walkers; a change of n_walkers is a new handle (and a new state buffer) like in the reference, where
// `update_curriculum` builds a new env.
#include "common.hpp"
#include "multiwalker_class.hpp"

#include <new>
#include <stdlib.h>

struct madrl_multiwalker {
    const madrl::MwClassApi *api;
    void *impl;
};

namespace {

using namespace madrl;

const MwClassApi *mw_class_of(const madrl_multiwalker_config *c) {
    if (!c) { (void)fail(MADRL_EINVAL, "config is NULL"); return nullptr; }
    if (c->struct_size != (int32_t)sizeof(madrl_multiwalker_config)) {
        (void)fail(MADRL_EINVAL, "madrl_multiwalker_config.struct_size=%d, library expects %d", c->struct_size, (int)sizeof(madrl_multiwalker_config));
        return nullptr;
    }
    const MwClassApi *classes[3] = {&madrl_mw_class_c4, &madrl_mw_class_c8, &madrl_mw_class_c10};
#ifdef MADRL_EXPERIMENTS   // measurement builds only (scripts/mw_occupancy.sh): run a walker count on a LARGER class than it needs (fewer envs per wavefront)
    if (const char *e = getenv("MADRL_MW_MIN_CLASS")) {
        const int want = atoi(e);
        for (const MwClassApi *k : classes) if (c->n_walkers >= 1 && c->n_walkers <= k->cap_walkers && k->cap_walkers >= want) return k;
    }
#endif
    if (c->n_walkers >= 1)
        for (const MwClassApi *k : classes) if (c->n_walkers <= k->cap_walkers) return k;
    (void)fail(MADRL_EINVAL, "n_walkers=%d unsupported (1..%d)", c->n_walkers, madrl_mw_class_c10.cap_walkers);
    return nullptr;
}

}  // namespace

extern "C" {

int madrl_multiwalker_obs_dim(const madrl_multiwalker_config *cfg, int32_t *out_dim) {
    const MwClassApi *k = mw_class_of(cfg);
    return k ? k->obs_dim(cfg, out_dim) : MADRL_EINVAL;
}

int madrl_multiwalker_state_bytes(const madrl_multiwalker_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    const MwClassApi *k = mw_class_of(cfg);
    return k ? k->state_bytes(cfg, n_envs, out_bytes) : MADRL_EINVAL;
}

int madrl_multiwalker_create(const madrl_multiwalker_config *cfg, int64_t n_envs, int32_t device, void *state_dev, madrl_multiwalker **out) {
    const MwClassApi *k = mw_class_of(cfg);
    if (!k) return MADRL_EINVAL;
    if (!out) return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    madrl_multiwalker *h = new (std::nothrow) madrl_multiwalker();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->api = k;
    h->impl = nullptr;
    const int rc = k->create(cfg, n_envs, device, state_dev, &h->impl);
    if (rc != MADRL_OK) { delete h; return rc; }
    *out = h;
    return MADRL_OK;
}

void madrl_multiwalker_destroy(madrl_multiwalker *h) {
    if (!h) return;
    h->api->destroy(h->impl);
    delete h;
}

int madrl_multiwalker_set_mode(madrl_multiwalker *h, int32_t fused, int32_t use_spares) {
    if (!h) return fail(MADRL_EINVAL, "set_mode: handle is NULL or a flag is not 0 / 1");
    return h->api->set_mode(h->impl, fused, use_spares);
}

int madrl_multiwalker_dims(const madrl_multiwalker *h, int32_t *n_bodies, int32_t *n_terrain) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    return h->api->dims(h->impl, n_bodies, n_terrain);
}

int madrl_multiwalker_record_bytes(const madrl_multiwalker *h, int32_t *stride_bytes, int32_t *world_bytes) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    return h->api->record_bytes(h->impl, stride_bytes, world_bytes);
}

int madrl_multiwalker_lanes(const madrl_multiwalker *h, int32_t *cap_walkers, int32_t *lanes_per_env) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (cap_walkers) *cap_walkers = h->api->cap_walkers;
    if (lanes_per_env) *lanes_per_env = h->api->lanes;
    return MADRL_OK;
}

int madrl_multiwalker_reset(madrl_multiwalker *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    return h->api->reset(h->impl, mask_dev, obs_dev, stream);
}

int madrl_multiwalker_reset_with(madrl_multiwalker *h, const uint8_t *mask_dev, const double *terrain_dev, const double *push_dev,
                                 float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    return h->api->reset_with(h->impl, mask_dev, terrain_dev, push_dev, obs_dev, stream);
}

int madrl_multiwalker_get_state(madrl_multiwalker *h, float *bodies_dev, float *joints_dev, float *aux_dev, uint8_t *flags_dev,
                                float *terrain_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    return h->api->get_state(h->impl, bodies_dev, joints_dev, aux_dev, flags_dev, terrain_dev, stream);
}

int madrl_multiwalker_set_state(madrl_multiwalker *h, const float *bodies_dev, const float *joints_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    return h->api->set_state(h->impl, bodies_dev, joints_dev, stream);
}

int madrl_multiwalker_step(madrl_multiwalker *h, const float *actions_dev, float *obs_dev, float *rew_dev, uint8_t *done_dev, void *stream) {
    if (!h || !actions_dev || !obs_dev || !rew_dev || !done_dev) return fail(MADRL_EINVAL, "step: NULL argument");
    return h->api->step(h->impl, actions_dev, obs_dev, rew_dev, done_dev, stream);
}

int madrl_multiwalker_get_bodies(madrl_multiwalker *h, float *bodies_dev, uint8_t *flags_dev, float *terrain_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    return h->api->get_bodies(h->impl, bodies_dev, flags_dev, terrain_dev, stream);
}

#if defined(MADRL_MW_TIMING)   // measurement builds (scripts/variants.sh): the four-walker class is the instrumented one
int madrl_multiwalker_debug_read(unsigned long long *stamps_host, int *vals_host) { return madrl_mw_class_c4.debug_read(stamps_host, vals_host); }
int madrl_multiwalker_debug_read_acc(unsigned long long *acc_host, int reset) { return madrl_mw_class_c4.debug_read_acc(acc_host, reset); }
#endif

}  // extern "C"
