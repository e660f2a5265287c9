// multiwalker.hip -- batched MultiWalkerEnv for MI355X (gfx950 / CDNA4), float32.
//
// FOUR ENVS PER WAVEFRONT: a group of 16 lanes owns one env (multiwalker_core.hpp `Par`: bodies and their terrain
// contacts by lane, one lane per leg for the revolute joints, the package / hull contacts one at a time), the four groups
// of a wavefront advance their envs through the same instruction stream.  The step is bound by instruction issue (about
// 1 MFLOP of dependent FP32 work per env-step in 180 + 60 Gauss-Seidel sweeps against ~12 KB of HBM traffic), so sharing
// the stream between four envs is worth almost 4x per wavefront.
//
// Per env, LDS holds only what the sweeps touch: mw::Hot (bodies, joints, flags: 1 KB) and mw::Scratch (active manifolds
// + schedule, sized by n_walkers: 5.3 KB at three walkers); the joint constants and accumulated impulses of a leg sit in
// the registers of the lane that owns it for the whole step.  mw::Cold (manifold cache with the warm-start impulses,
// terrain) is touched once per step by Collide / StoreImpulses / lidar and is read and written in place in HBM (L2).
// Lanes of a group that work concurrently never share a body and every pair of constraints that shares a body keeps its
// serial order, so the result equals the serial sweep of the CPU build bit for bit.
//
// PARITY UNPINNED (Box2D is not available to pin against) -- see multiwalker_core.hpp.
#include "common.hpp"
#include "multiwalker_core.hpp"

#include <new>
#include <stdlib.h>
#include <string.h>

namespace {

using namespace madrl;

struct MwDev {
    mw::EnvCfg cfg;
    uint32_t gid_base;
    int32_t world_dw;      // dwords per env in the state buffer
    int32_t scratch_bytes; // mw::Scratch truncated to Model::max_manifolds manifolds, 16-byte aligned
    int32_t env_lds_bytes; // LDS per env: Hot | Scratch | actions, rewards, done
    int64_t n_envs;
    const mw::Model *model;
    uint32_t *state;
};
struct MwIO {
    const double *inj_terrain;  // reset only, parity hook: [N][NT] terrain heights instead of the Philox walk (or NULL)
    const double *inj_push;     // reset only, parity hook: [N][W] initial pushes (or NULL)
    const uint8_t *mask;
    const float *actions;  // [N][W][4]
    float *obs;            // [N][W][32]
    float *rew;            // [N][W]
    uint8_t *done;         // [N]
};

constexpr int HOT_BYTES = (int)((sizeof(mw::Hot) + 15) / 16 * 16);
constexpr int IO_BYTES = (4 * mw::MAX_WALKERS + mw::MAX_WALKERS + 4) * 4;  // s_act | s_rew | s_done

__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the cooperating lanes of multiwalker_core.hpp's `Par` = one group of 64 / EPW lanes of the wavefront
template <int EPW>
struct GroupPar {
    static constexpr int NL = 64 / EPW;
    static constexpr int JOINTS = (mw::MAXJ + NL - 1) / NL;
    static constexpr int BODIES = (mw::MAXB + NL - 1) / NL;
    int l;
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ int n() const { return NL; }
    __device__ __forceinline__ void sync() const { lds_sync(); }
    __device__ __forceinline__ int alloc(int *counter) const { return atomicAdd(counter, 1); }
    __device__ __forceinline__ void or_bits(uint32_t *p, uint32_t v) const { atomicOr(p, v); }
};

// MODE 0: reset(mask)   MODE 1: step (+ fused auto-reset)
template <int MODE, int EPW>
__global__ __launch_bounds__(64, 2) void multiwalker_kernel(const MwDev d, const MwIO io) {
    const mw::Model &M = *d.model;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NL = 64 / EPW;
    const int g = threadIdx.x / NL, lane = threadIdx.x % NL;
    const GroupPar<EPW> par{lane};
    unsigned char *base = smem + g * d.env_lds_bytes;
    mw::Hot &Wd = *reinterpret_cast<mw::Hot *>(base);
    mw::Scratch &S = *reinterpret_cast<mw::Scratch *>(base + HOT_BYTES);
    float *s_act = reinterpret_cast<float *>(base + HOT_BYTES + d.scratch_bytes);
    float *s_rew = s_act + 4 * mw::MAX_WALKERS;
    uint32_t *s_done = reinterpret_cast<uint32_t *>(s_rew + mw::MAX_WALKERS);
    const int W = M.W;
    for (int64_t e0 = (int64_t)blockIdx.x * EPW; e0 < d.n_envs; e0 += (int64_t)gridDim.x * EPW) {
        const int64_t env = e0 + g;
        bool active = env < d.n_envs;
        if (MODE == 0 && active && io.mask != nullptr && io.mask[env] == 0) active = false;
        if (active) {
            uint32_t *rec = d.state + env * (int64_t)d.world_dw;
            mw::Cold &Cd = *reinterpret_cast<mw::Cold *>(rec + sizeof(mw::Hot) / 4);
            {
                uint32_t *dst = reinterpret_cast<uint32_t *>(&Wd);
                for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) dst[k] = rec[k];
                for (int k = lane; k < 4 * mw::MAX_WALKERS; k += NL) s_act[k] = (MODE == 1 && k < 4 * W) ? io.actions[env * 4 * W + k] : 0.0f;
                if (lane == 0) *s_done = 0;
            }
            lds_sync();
            const uint32_t gid = d.gid_base + (uint32_t)env;
            float *obs_row = io.obs + env * W * mw::obs_dim_of(d.cfg);  // observation rows go straight to HBM
            // pass 0: the step proper (MODE 1); pass 1: MultiWalkerEnv.reset (:330-357), which ends with step(zeros).
            // One call site, so that env_step is inlined here and keeps LDS addressing for Wd / S.
            uint32_t dn = 0;
            for (int pass = (MODE == 1 ? 0 : 1); pass < 2; ++pass) {
                if (pass == 1) {
                    dn = *s_done;  // uniform over the group
                    if (!(MODE == 0 || (dn != 0 && d.cfg.auto_reset))) break;
                    for (int k = lane; k < 4 * mw::MAX_WALKERS; k += NL) s_act[k] = 0.0f;
                    if (lane == 0) mw::env_reset_world(M, d.cfg, Wd, Cd, gid, (MODE == 0 && io.inj_terrain) ? io.inj_terrain + env * M.NT : nullptr,
                                                       (MODE == 0 && io.inj_push) ? io.inj_push + env * W : nullptr);
                    lds_sync();
                }
                mw::env_step(M, d.cfg, Wd, Cd, S, par, gid, s_act, obs_row, pass == 0 ? s_rew : (float *)nullptr,
                             pass == 0 ? reinterpret_cast<uint8_t *>(s_done) : (uint8_t *)nullptr);
                if (pass == 0) { if (lane == 0 && d.cfg.max_steps > 0 && Wd.t >= d.cfg.max_steps) *s_done |= 2; }
                else if (lane == 0) Wd.t = 0;
                lds_sync();
            }
            if (MODE == 1) {
                if (lane < W) io.rew[env * W + lane] = s_rew[lane];
                if (lane == 0) io.done[env] = (uint8_t)dn;
            }
            {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(&Wd);
                for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) rec[k] = src[k];
            }
        }
        // the next env of this group reuses the LDS block; its Cold part is other memory, nothing to wait for
        lds_sync();
    }
}

__global__ void mw_get_bodies_kernel(const MwDev d, float *bodies, uint8_t *flags, float *terrain) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    const mw::World *wr = reinterpret_cast<const mw::World *>(d.state + env * (int64_t)d.world_dw);
    const mw::Hot *w = &wr->h;
    const int NB = d.model->NB, W = d.model->W, NT = d.model->NT;
    if (bodies)
        for (int b = 0; b < NB; ++b) {
            float *p = bodies + (env * NB + b) * 6;
            p[0] = w->b[b].c.x; p[1] = w->b[b].c.y; p[2] = w->b[b].a; p[3] = w->b[b].v.x; p[4] = w->b[b].v.y; p[5] = w->b[b].w;
        }
    if (flags) {
        uint8_t *f = flags + env * (1 + 3 * W);
        f[0] = w->game_over;
        for (int k = 0; k < W; ++k) { f[1 + k] = w->fallen[k]; f[1 + W + 2 * k] = w->ground[k][0]; f[1 + W + 2 * k + 1] = w->ground[k][1]; }
    }
    if (terrain) for (int i = 0; i < NT; ++i) terrain[env * NT + i] = wr->c.ty[i];
}

// unpacked state (checkpoint / teacher-forcing hook); any pointer may be NULL
__global__ void mw_get_state_kernel(const MwDev d, float *bodies, float *joints, float *aux, uint8_t *flags, float *terrain) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    const mw::World *wr = reinterpret_cast<const mw::World *>(d.state + env * (int64_t)d.world_dw);
    const mw::Hot *w = &wr->h;
    const int NB = d.model->NB, W = d.model->W, NT = d.model->NT, NJ = d.model->NJ;
    if (bodies)
        for (int b = 0; b < NB; ++b) {
            float *p = bodies + (env * NB + b) * 6;
            p[0] = w->b[b].c.x; p[1] = w->b[b].c.y; p[2] = w->b[b].a; p[3] = w->b[b].v.x; p[4] = w->b[b].v.y; p[5] = w->b[b].w;
        }
    if (joints)
        for (int j = 0; j < NJ; ++j) {
            const mw::Joint &q = wr->c.j[j];
            float *p = joints + (env * NJ + j) * 6;
            p[0] = q.ix; p[1] = q.iy; p[2] = q.iz; p[3] = q.motor_impulse; p[4] = (float)q.limit_state; p[5] = q.motor_speed;
        }
    if (aux)
        for (int b = 0; b < NB; ++b) {
            float *p = aux + (env * NB + b) * 6;
            for (int k = 0; k < 4; ++k) p[k] = wr->c.fat[b][k];
            p[4] = wr->c.sleep_time[b]; p[5] = (float)((w->awake >> b) & 1u);
        }
    if (flags) {
        uint8_t *f = flags + env * (2 + 3 * W);
        f[0] = w->game_over;
        for (int k = 0; k < W; ++k) { f[1 + k] = w->fallen[k]; f[1 + W + 2 * k] = w->ground[k][0]; f[1 + W + 2 * k + 1] = w->ground[k][1]; }
        f[1 + 3 * W] = w->overflow;
    }
    if (terrain) for (int i = 0; i < NT; ++i) terrain[env * NT + i] = wr->c.ty[i];
}
__global__ void mw_set_state_kernel(const MwDev d, const float *bodies, const float *joints) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    mw::World *wr = reinterpret_cast<mw::World *>(d.state + env * (int64_t)d.world_dw);
    const int NB = d.model->NB, NJ = d.model->NJ;
    if (bodies)
        for (int b = 0; b < NB; ++b) {
            const float *p = bodies + (env * NB + b) * 6;
            mw::Body &q = wr->h.b[b];
            q.c = mw::v2(p[0], p[1]); q.a = p[2]; q.v = mw::v2(p[3], p[4]); q.w = p[5];
        }
    if (joints)
        for (int j = 0; j < NJ; ++j) {
            const float *p = joints + (env * NJ + j) * 6;
            mw::Joint &q = wr->c.j[j];
            q.ix = p[0]; q.iy = p[1]; q.iz = p[2]; q.motor_impulse = p[3]; q.limit_state = (int)p[4]; q.motor_speed = p[5];
        }
}

}  // namespace

struct madrl_multiwalker {
    madrl_multiwalker_config cfg;
    MwDev dev;
    int device;
    int epw;  // envs per wavefront: 4 (default), 2 or 1 (MADRL_MW_EPW at create: experiments)
    int64_t max_blocks;
    void *model_dev;
    int NB, NT;
};

namespace {

int mw_validate(const madrl_multiwalker_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_multiwalker_config))
        return fail(MADRL_EINVAL, "madrl_multiwalker_config.struct_size=%d, library expects %d", c->struct_size,
                    (int)sizeof(madrl_multiwalker_config));
    if (c->n_walkers < 1 || c->n_walkers > mw::MAX_WALKERS)
        return fail(MADRL_EINVAL, "n_walkers=%d unsupported (1..%d)", c->n_walkers, mw::MAX_WALKERS);
    return MADRL_OK;
}

template <int EPW>
void mw_launch_epw(const madrl_multiwalker *h, const MwIO &io, int mode, hipStream_t s) {
    int64_t blocks = (h->dev.n_envs + EPW - 1) / EPW;   // default: every group of EPW envs gets its own wavefront
    if (h->max_blocks > 0 && blocks > h->max_blocks) blocks = h->max_blocks;
    const size_t lds = (size_t)EPW * h->dev.env_lds_bytes;
    if (mode == 0) hipLaunchKernelGGL((multiwalker_kernel<0, EPW>), dim3((unsigned)blocks), dim3(64), lds, s, h->dev, io);
    else hipLaunchKernelGGL((multiwalker_kernel<1, EPW>), dim3((unsigned)blocks), dim3(64), lds, s, h->dev, io);
}

int mw_launch(const madrl_multiwalker *h, const MwIO &io, int mode, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (h->epw == 1) mw_launch_epw<1>(h, io, mode, s);
    else if (h->epw == 2) mw_launch_epw<2>(h, io, mode, s);
    else mw_launch_epw<4>(h, io, mode, s);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // namespace

extern "C" {

int madrl_multiwalker_obs_dim(const madrl_multiwalker_config *cfg, int32_t *out_dim) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (!out_dim) return fail(MADRL_EINVAL, "out_dim is NULL");
    *out_dim = mw::OBS_DIM - 1 + (cfg->one_hot ? mw::MAX_AGENTS_ID : 1);
    return MADRL_OK;
}

int madrl_multiwalker_state_bytes(const madrl_multiwalker_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_bytes) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_bytes non-NULL");
    *out_bytes = (uint64_t)align_up(sizeof(mw::World), 16) * (uint64_t)n_envs;
    return MADRL_OK;
}

int madrl_multiwalker_create(const madrl_multiwalker_config *cfg, int64_t n_envs, int32_t device, void *state_dev,
                             madrl_multiwalker **out) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (!state_dev || !out || n_envs < 1) return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    if (n_envs + cfg->env_id_base > 0xFFFFFFFFll) return fail(MADRL_EINVAL, "global env index must fit 32 bits");
    MADRL_HIP_TRY(hipSetDevice(device));
    madrl_multiwalker *h = new (std::nothrow) madrl_multiwalker();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    h->max_blocks = 0;
    mw::Model M;
    memset(&M, 0, sizeof(M));
    mw::build_model(M, cfg->n_walkers);
    M.continuous = cfg->discrete_only ? 0 : 1;
    if (const char *e = getenv("MADRL_MW_TOI")) M.continuous = atoi(e);  // experiments: 0 = no continuous pass, 2 = candidates only
    h->NB = M.NB; h->NT = M.NT;
    hipError_t e = hipMalloc(&h->model_dev, sizeof(M));
    if (e == hipSuccess) e = hipMemcpy(h->model_dev, &M, sizeof(M), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->model_dev) (void)hipFree(h->model_dev);
        delete h;
        return fail(MADRL_EHIP, "model upload failed: %s", hipGetErrorString(e));
    }
    MwDev &d = h->dev;
    memset(&d, 0, sizeof(d));
    d.cfg.n_walkers = cfg->n_walkers; d.cfg.reward_global = cfg->reward_global; d.cfg.terminate_on_fall = cfg->terminate_on_fall;
    d.cfg.one_hot = cfg->one_hot ? 1 : 0; d.cfg.max_steps = cfg->max_steps; d.cfg.auto_reset = cfg->auto_reset;
    d.cfg.position_noise = (float)cfg->position_noise; d.cfg.angle_noise = (float)cfg->angle_noise;
    d.cfg.forward_reward = (float)cfg->forward_reward; d.cfg.fall_reward = (float)cfg->fall_reward;
    d.cfg.drop_reward = (float)cfg->drop_reward;
    d.cfg.k0 = (uint32_t)cfg->seed; d.cfg.k1 = (uint32_t)(cfg->seed >> 32);
    d.gid_base = (uint32_t)cfg->env_id_base;
    d.world_dw = (int32_t)(align_up(sizeof(mw::World), 16) / 4);
    d.scratch_bytes = (int32_t)align_up(sizeof(mw::Scratch) - (size_t)(mw::MAXM - M.max_manifolds) * sizeof(mw::Manifold), 16);
    d.env_lds_bytes = HOT_BYTES + d.scratch_bytes + (int32_t)align_up(IO_BYTES, 16);
    // 8 resident wavefronts per CU (two per SIMD) need 4 envs x env_lds_bytes <= 20 KB; three walkers: 5 040 bytes per env
    h->epw = 4;
    if (const char *e = getenv("MADRL_MW_EPW")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) h->epw = v; }
    d.n_envs = n_envs;
    d.model = (const mw::Model *)h->model_dev;
    d.state = (uint32_t *)state_dev;
    *out = h;
    return MADRL_OK;
}

void madrl_multiwalker_destroy(madrl_multiwalker *h) {
    if (!h) return;
    if (h->model_dev) (void)hipFree(h->model_dev);
    delete h;
}

int madrl_multiwalker_set_launch(madrl_multiwalker *h, int64_t max_blocks) {
    if (!h || max_blocks < 0) return fail(MADRL_EINVAL, "set_launch: bad argument");
    h->max_blocks = max_blocks;
    return MADRL_OK;
}

int madrl_multiwalker_dims(const madrl_multiwalker *h, int32_t *n_bodies, int32_t *n_terrain) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (n_bodies) *n_bodies = h->NB;
    if (n_terrain) *n_terrain = h->NT;
    return MADRL_OK;
}

int madrl_multiwalker_reset(madrl_multiwalker *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    MwIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.obs = obs_dev;
    return mw_launch(h, io, 0, stream);
}

int madrl_multiwalker_reset_with(madrl_multiwalker *h, const uint8_t *mask_dev, const double *terrain_dev, const double *push_dev,
                                 float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    MwIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.inj_terrain = terrain_dev;
    io.inj_push = push_dev;
    io.obs = obs_dev;
    return mw_launch(h, io, 0, stream);
}

int madrl_multiwalker_get_state(madrl_multiwalker *h, float *bodies_dev, float *joints_dev, float *aux_dev, uint8_t *flags_dev,
                                float *terrain_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_get_state_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, joints_dev, aux_dev,
                       flags_dev, terrain_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_multiwalker_set_state(madrl_multiwalker *h, const float *bodies_dev, const float *joints_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_set_state_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, joints_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_multiwalker_step(madrl_multiwalker *h, const float *actions_dev, float *obs_dev, float *rew_dev,
                           uint8_t *done_dev, void *stream) {
    if (!h || !actions_dev || !obs_dev || !rew_dev || !done_dev) return fail(MADRL_EINVAL, "step: NULL argument");
    MwIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions_dev;
    io.obs = obs_dev;
    io.rew = rew_dev;
    io.done = done_dev;
    return mw_launch(h, io, 1, stream);
}

int madrl_multiwalker_get_bodies(madrl_multiwalker *h, float *bodies_dev, uint8_t *flags_dev, float *terrain_dev,
                                 void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_get_bodies_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, flags_dev,
                       terrain_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // extern "C"
