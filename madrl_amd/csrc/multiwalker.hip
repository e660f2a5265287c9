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
    int32_t world_dw;      // dwords per env in the state buffer: mw::World, then the step's Scratch
    int32_t scratch_off_dw; // where the Scratch starts inside an env's block
    uint8_t *pending;      // [n_envs] at the end of the state buffer: this env runs the trailing step of a reset in pass 1
    int32_t scratch_bytes; // mw::Scratch truncated to Model::max_manifolds manifolds, 16-byte aligned
    int32_t env_lds_bytes; // LDS per env outside the solver launch, first part: Hot | Scratch | actions, rewards, done
    int32_t env_lds_bytes_staged;  // ... all of it: | the used part of mw::Cold | mw::ToiWork
    int32_t env_lds_bytes_solve;   // LDS per env in the solver launch: Hot | the solver's part of Scratch
    int32_t toi_lane0_bytes;       // time-of-impact cache of lane 0 (the package's contact slots)
    int32_t cold_dw;       // dwords of mw::Cold in use (up to the last slot of this walker count)
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

#ifndef MADRL_MW_SOLVE_WAVES
#define MADRL_MW_SOLVE_WAVES 1   // resident wavefronts per SIMD the solver launch's registers are allocated for
#endif
#ifndef MADRL_MW_SOLVE_MREG
#define MADRL_MW_SOLVE_MREG 3    // manifolds a solver lane holds in registers for the whole solve
#endif
constexpr int SOLVE_EPW = 64 / mw::SOLVE_LANES;   // envs per wavefront in the solver launch: one lane per walker
constexpr int SOLVE_HDR_BYTES = (int)((offsetof(mw::Scratch, dyn_midx) + 15) / 16 * 16);   // the part of Scratch the solver works on
#ifndef MADRL_MW_SOLVE_OVERFLOW
#define MADRL_MW_SOLVE_OVERFLOW 8   // manifolds per env the solver launch can hold in LDS on top of the lanes' register copies
#endif
constexpr int TOI_WORK_BYTES = (int)((sizeof(mw::ToiWork) + 15) / 16 * 16);
constexpr int TOI_LANE_BYTES = (mw::EDGE_SLOTS_HULL * 5 + 15) / 16 * 16;   // time-of-impact cache of a walker's body: 4 + 1 bytes per contact slot
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
    static constexpr int SOLVE_EMU = 1;                    // mw::step_solve: this lane IS one solver lane
    static constexpr int MREG = MADRL_MW_SOLVE_MREG;
    int l;
    __device__ __forceinline__ int solve_lane(int) const { return l; }
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ int n() const { return NL; }
    __device__ __forceinline__ void sync() const { lds_sync(); }
    __device__ __forceinline__ int alloc(int *counter) const { return atomicAdd(counter, 1); }
    __device__ __forceinline__ void or_bits(uint32_t *p, uint32_t v) const { atomicOr(p, v); }
};

// One API call is a SEQUENCE of launches over the same per-env records (Hot in LDS for the duration of a launch, Cold in place in
// HBM, the step's Scratch -- manifolds and schedule -- handed from launch to launch through the caller's state buffer):
//   PH_COLLIDE  apply_action, b2ContactManager::Collide, islands + level schedule        (mw::step_collide)
//   PH_SOLVE    b2Island::Solve by levels, sleeping, SynchronizeFixtures, FindNewContacts (mw::step_solve)
//   PH_TOI      b2World::SolveTOI, then the observation / reward / done of the step       (mw::solve_toi, mw::env_observe)
//   PH_RESET    MultiWalkerEnv.reset (:330-357) without its trailing step                 (mw::env_reset_world)
// Three kernels instead of one because a kernel's register allocation is the maximum over its phases: the narrow phase and the
// time-of-impact root finder (GJK) need 250+ VGPRs, and in one kernel the 180-sweep solver loop ran at two wavefronts per SIMD with
// its joint state spilled to scratch memory.
// pass 0 = the step proper (every env, the caller's actions, rewards / done written); pass 1 = the trailing zero-action step of a
// reset (:357), only for the envs whose byte in `pending` is set -- by PH_RESET (reset(mask)) or by PH_TOI of pass 0 (auto-reset).
enum { PH_RESET = 0, PH_COLLIDE = 1, PH_SOLVE = 2, PH_TOI = 3 };

// Collide, the continuous pass and reset walk the contacts, the terrain and the broad phase's boxes with data-dependent, mostly
// single-lane access chains: from HBM / L2 every link costs a memory round trip.  Those launches copy the env's Cold record into LDS
// (coalesced), work there and copy it back; the solver launch touches Cold once per step and leaves it in HBM.
template <int PHASE> struct PhaseStage { static constexpr bool value = true; };
template <> struct PhaseStage<PH_SOLVE> { static constexpr bool value = false; };
template <int PHASE> struct PhaseOcc { static constexpr int value = 2; };
template <> struct PhaseOcc<PH_SOLVE> { static constexpr int value = MADRL_MW_SOLVE_WAVES; };
template <> struct PhaseOcc<PH_TOI> { static constexpr int value = 1; };   // (LDS allows one wavefront per SIMD anyway; a mini island's manifolds live in registers)

template <int PHASE, int EPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PhaseOcc<PHASE>::value, PhaseOcc<PHASE>::value)))
void mw_phase_kernel(const MwDev d, const MwIO io, const int pass) {
    const mw::Model &M = *d.model;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NL = 64 / EPW;
    const int g = threadIdx.x / NL, lane = threadIdx.x % NL;
    const GroupPar<EPW> par{lane};
    constexpr bool STAGE = PhaseStage<PHASE>::value;
    static_assert(PHASE != PH_SOLVE || NL == mw::SOLVE_LANES, "the solver launch runs one lane per walker");
    unsigned char *base = smem + g * (STAGE ? d.env_lds_bytes_staged : d.env_lds_bytes_solve);
    mw::Hot &Wd = *reinterpret_cast<mw::Hot *>(base);
    mw::Scratch &S = *reinterpret_cast<mw::Scratch *>(base + HOT_BYTES);
    float *s_act = reinterpret_cast<float *>(base + HOT_BYTES + d.scratch_bytes);
    float *s_rew = s_act + 4 * mw::MAX_WALKERS;
    uint32_t *s_done = reinterpret_cast<uint32_t *>(s_rew + mw::MAX_WALKERS);
    const int W = M.W;
    constexpr int SCR_HDR_DW = (int)(offsetof(mw::Scratch, m) / 4);
    for (int64_t e0 = (int64_t)blockIdx.x * EPW; e0 < d.n_envs; e0 += (int64_t)gridDim.x * EPW) {
        const int64_t env = e0 + g;
        bool active = env < d.n_envs;
        if (active) {
            if (PHASE == PH_RESET) active = io.mask ? io.mask[env] != 0 : (pass == 0 || d.pending[env] != 0);   // reset(mask) / auto-reset
            else if (pass == 1) active = d.pending[env] != 0;
        }
        if (active) {
            uint32_t *rec = d.state + env * (int64_t)d.world_dw;
            uint32_t *cold_g = rec + sizeof(mw::Hot) / 4;
            uint32_t *cold_l = reinterpret_cast<uint32_t *>(base + d.env_lds_bytes);   // staged copy (launches with STAGE)
            mw::Cold &Cd = *reinterpret_cast<mw::Cold *>(STAGE ? cold_l : cold_g);
            uint32_t *scr = rec + d.scratch_off_dw;   // the step's Scratch between launches
            {
                uint32_t *dst = reinterpret_cast<uint32_t *>(&Wd);
                for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) dst[k] = rec[k];
                if (STAGE) for (int k = lane; k < d.cold_dw; k += NL) cold_l[k] = cold_g[k];
                if (PHASE == PH_SOLVE || PHASE == PH_TOI) {   // the schedule the collide launch built (the solver: only its part)
                    uint32_t *sd = reinterpret_cast<uint32_t *>(&S);
                    for (int k = lane; k < (PHASE == PH_SOLVE ? SOLVE_HDR_BYTES / 4 : SCR_HDR_DW); k += NL) sd[k] = scr[k];
                }
            }
            lds_sync();
            const uint32_t gid = d.gid_base + (uint32_t)env;
            if (PHASE == PH_RESET) {
                if (lane == 0) {
                    mw::env_reset_world(M, d.cfg, Wd, Cd, gid, io.inj_terrain ? io.inj_terrain + env * M.NT : nullptr, io.inj_push ? io.inj_push + env * W : nullptr);
                    d.pending[env] = 1;
                }
            } else if (PHASE == PH_COLLIDE) {
                for (int k = lane; k < 4 * mw::MAX_WALKERS; k += NL) s_act[k] = (pass == 0 && k < 4 * W) ? io.actions[env * 4 * W + k] : 0.0f;
                lds_sync();
                mw::env_apply_actions(M, Wd, Cd, par, s_act);
                mw::step_collide(M, Wd, Cd, S, par);
                const int nm = S.nm < M.max_manifolds ? S.nm : M.max_manifolds;
                const uint32_t *sd = reinterpret_cast<const uint32_t *>(&S);
                for (int k = lane; k < SCR_HDR_DW + nm * (int)(sizeof(mw::Manifold) / 4); k += NL) scr[k] = sd[k];
            } else if (PHASE == PH_SOLVE) {
                // the manifolds the collide launch emitted stay in the state buffer: every lane copies the ones it owns into registers
                mw::step_solve(M, Wd, Cd, S, reinterpret_cast<mw::Manifold *>(scr + SCR_HDR_DW),
                               reinterpret_cast<mw::Manifold *>(base + HOT_BYTES + SOLVE_HDR_BYTES), MADRL_MW_SOLVE_OVERFLOW, par);
            } else {
                mw::step_post(M, Wd, Cd, S, par);
                if (M.continuous) {
                    // per lane: the time-of-impact cache of the body it works on (lane 0 may hold the package: the largest contact cache) and
                    // room for the manifolds of a mini island past the four in registers -- in the manifold pool of the state buffer,
                    // free during this launch: most of it for lane 0, a few entries for every other lane
                    unsigned char *tw = base + d.env_lds_bytes + d.cold_dw * 4;
                    mw::ToiLaneWork TL;
                    unsigned char *lc = tw + TOI_WORK_BYTES + (lane == 0 ? 0 : d.toi_lane0_bytes + (lane - 1) * TOI_LANE_BYTES);
                    const int lcap = lane == 0 ? d.toi_lane0_bytes / 5 : TOI_LANE_BYTES / 5;
                    TL.alpha = reinterpret_cast<float *>(lc); TL.meta = lc + 4 * lcap;
                    mw::Manifold *pool = reinterpret_cast<mw::Manifold *>(scr + SCR_HDR_DW);
                    constexpr int CO = NL <= 4 ? 4 : 1;
                    const int c0 = M.max_manifolds - CO * ((NL < M.NB ? NL : M.NB) - 1);   // (lanes past the last body own nothing)
                    TL.ovf = lane == 0 ? pool : pool + c0 + CO * (lane - 1);
                    TL.ovf_cap = lane == 0 ? c0 : CO;
                    mw::solve_toi(M, Wd, Cd, S, *reinterpret_cast<mw::ToiWork *>(tw), TL, par, 1.0f / mw::FPS);
                }
                float *obs_row = io.obs + env * W * mw::obs_dim_of(d.cfg);  // observation rows go straight to HBM
                if (lane == 0) {
                    *s_done = 0;
                    mw::env_observe(M, d.cfg, Wd, Cd, gid, obs_row, pass == 0 ? s_rew : (float *)nullptr, pass == 0 ? reinterpret_cast<uint8_t *>(s_done) : (uint8_t *)nullptr);
                    Wd.t += 1;
                    Wd.tick += 1;
                    if (pass == 0) {
                        if (d.cfg.max_steps > 0 && Wd.t >= d.cfg.max_steps) *s_done |= 2;
                        d.pending[env] = (d.cfg.auto_reset && *s_done != 0) ? 1 : 0;
                    } else {
                        Wd.t = 0;   // the reset's trailing step does not count (:357)
                        d.pending[env] = 0;
                    }
                }
                lds_sync();
                if (pass == 0) {
                    if (lane < W) io.rew[env * W + lane] = s_rew[lane];
                    if (lane == 0) io.done[env] = (uint8_t)*s_done;
                }
            }
            lds_sync();
            {
                const uint32_t *src = reinterpret_cast<const uint32_t *>(&Wd);
                for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) rec[k] = src[k];
                if (STAGE) for (int k = lane; k < d.cold_dw; k += NL) cold_g[k] = cold_l[k];
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
    int epw_staged;  // envs per wavefront in the launches that stage Cold in LDS: 2 (default) or 4 (MADRL_MW_EPW_STAGED at create: experiments)
    int64_t max_blocks;
    void *model_dev;
    int NB, NT;
};

namespace {

size_t mw_scratch_bytes(const mw::Model &M) {   // mw::Scratch truncated to Model::max_manifolds manifolds, 16-byte aligned
    return align_up(sizeof(mw::Scratch) - (size_t)(mw::MAXM - M.max_manifolds) * sizeof(mw::Manifold), 16);
}

int mw_validate(const madrl_multiwalker_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_multiwalker_config))
        return fail(MADRL_EINVAL, "madrl_multiwalker_config.struct_size=%d, library expects %d", c->struct_size,
                    (int)sizeof(madrl_multiwalker_config));
    if (c->n_walkers < 1 || c->n_walkers > mw::MAX_WALKERS)
        return fail(MADRL_EINVAL, "n_walkers=%d unsupported (1..%d)", c->n_walkers, mw::MAX_WALKERS);
    return MADRL_OK;
}

template <int PH, int EPW>
void mw_launch_phase(const madrl_multiwalker *h, const MwIO &io, int pass, hipStream_t s) {
    int64_t blocks = (h->dev.n_envs + EPW - 1) / EPW;   // default: every group of EPW envs gets its own wavefront
    if (h->max_blocks > 0 && blocks > h->max_blocks) blocks = h->max_blocks;
    const size_t lds = (size_t)EPW * (PhaseStage<PH>::value ? h->dev.env_lds_bytes_staged : h->dev.env_lds_bytes_solve);
    hipLaunchKernelGGL((mw_phase_kernel<PH, EPW>), dim3((unsigned)blocks), dim3(64), lds, s, h->dev, io, pass);
}
// EPW envs per wavefront in the solver launch (16 lanes per env: one per joint / body), ES in the launches that stage Cold in LDS
template <int EPW, int ES>
void mw_launch_epw(const madrl_multiwalker *h, const MwIO &io, int mode, hipStream_t s) {
#define MW_PHASE(PH, PASS) mw_launch_phase<PH, (PH == PH_SOLVE ? EPW : ES)>(h, io, PASS, s)
    if (mode == 1) {   // MultiWalkerEnv.step
        MW_PHASE(PH_COLLIDE, 0); MW_PHASE(PH_SOLVE, 0); MW_PHASE(PH_TOI, 0);
        if (!h->cfg.auto_reset) return;
    }
    // MultiWalkerEnv.reset(mask), or the fused auto-reset of the envs whose step just ended their episode: reset, then step(zeros)
    MwIO r = io;
    if (mode == 1) { r.mask = nullptr; r.inj_terrain = nullptr; r.inj_push = nullptr; }
    mw_launch_phase<PH_RESET, ES>(h, r, mode == 1 ? 1 : 0, s);
    MW_PHASE(PH_COLLIDE, 1); MW_PHASE(PH_SOLVE, 1); MW_PHASE(PH_TOI, 1);
#undef MW_PHASE
}

int mw_launch(const madrl_multiwalker *h, const MwIO &io, int mode, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (h->epw_staged == 4) mw_launch_epw<SOLVE_EPW, 4>(h, io, mode, s);
    else mw_launch_epw<SOLVE_EPW, 2>(h, io, mode, s);
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
    mw::Model M;
    memset(&M, 0, sizeof(M));
    mw::build_model(M, cfg->n_walkers);
    // per env: the world record, then the step's Scratch (manifolds + schedule handed from launch to launch); then one byte per env
    *out_bytes = (uint64_t)(align_up(sizeof(mw::World), 16) + mw_scratch_bytes(M)) * (uint64_t)n_envs + align_up((uint64_t)n_envs, 16);
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
    d.scratch_bytes = (int32_t)mw_scratch_bytes(M);
    d.scratch_off_dw = (int32_t)(align_up(sizeof(mw::World), 16) / 4);
    d.world_dw = d.scratch_off_dw + d.scratch_bytes / 4;
    d.pending = (uint8_t *)state_dev + (size_t)d.world_dw * 4 * (size_t)n_envs;
    d.env_lds_bytes = HOT_BYTES + d.scratch_bytes + (int32_t)align_up(IO_BYTES, 16);
    d.cold_dw = (int32_t)(align_up(offsetof(mw::Cold, slot) + (size_t)M.n_slots * sizeof(mw::Slot), 16) / 4);
    d.env_lds_bytes_solve = HOT_BYTES + SOLVE_HDR_BYTES + MADRL_MW_SOLVE_OVERFLOW * (int32_t)sizeof(mw::Manifold);
    d.env_lds_bytes_solve = (d.env_lds_bytes_solve + 255 - 16) / 256 * 256 + 16;   // env g's block starts 4 LDS banks after env g-1's: the lanes
                                                                                    // of a wavefront (16 envs) then hit disjoint banks with 16-byte accesses
    d.toi_lane0_bytes = (int32_t)align_up((size_t)(M.slot_cap[0] > mw::EDGE_SLOTS_HULL ? M.slot_cap[0] : mw::EDGE_SLOTS_HULL) * 5, 16);
    d.env_lds_bytes_staged = d.env_lds_bytes + d.cold_dw * 4 + TOI_WORK_BYTES + d.toi_lane0_bytes + (mw::MAXB - 1) * TOI_LANE_BYTES;
    // 8 resident wavefronts per CU (two per SIMD) need 4 envs x env_lds_bytes <= 20 KB; three walkers: 5 040 bytes per env
    h->epw_staged = 2;
    if (const char *e = getenv("MADRL_MW_EPW_STAGED")) { const int v = atoi(e); if (v == 2 || v == 4) h->epw_staged = v; }
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

int madrl_multiwalker_record_bytes(const madrl_multiwalker *h, int32_t *stride_bytes, int32_t *world_bytes) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (stride_bytes) *stride_bytes = h->dev.world_dw * 4;
    if (world_bytes) *world_bytes = (int32_t)sizeof(mw::World);
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
