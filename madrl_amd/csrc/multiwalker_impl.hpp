// multiwalker_impl.hpp -- batched MultiWalkerEnv for MI355X (gfx950 / CDNA4), float32: the kernels and the host side of ONE capacity
// class.  Included by multiwalker_c4.hip / _c8.hip / _c10.hip, which set MW_CAPW (walkers the class has room for) and MW_NLANES (lanes
// per env); multiwalker.hip holds the C ABI and picks the class by n_walkers.
//
// SIXTEEN ENVS PER WAVEFRONT, FOUR LANES PER ENV (up to four walkers; eight envs of eight lanes for 5 .. 8 walkers, four envs of sixteen
// lanes for 9 and 10 -- the reference's curriculum, lessons/multiwalker/env.yaml:1-27).  A b2World::Step of this env is a long chain of
// short, dependent float32 updates (180 + 60 Gauss-Seidel sweeps over 4 joints per walker and a handful of contacts, then the continuous
// pass): there is no data parallelism inside an env beyond its walkers, so the lanes of a wavefront are filled with ENVS.  Lane w of an
// env's group owns walker w: its four joints live in that lane's registers for the whole solve; contacts are dealt out to the env's lanes
// by a list schedule; in the continuous pass every body's chain of time-of-impact events runs on its own lane.  16 384 envs of three
// walkers are 1 024 wavefronts -- one per SIMD of the chip, all resident at once.
//
// One API call is a sequence of launches (collide | solve | continuous pass + observe, see below) over the per-env records in the
// caller's state buffer: mw::Hot (bodies, flags: 0.6 KB) is in LDS for the duration of a launch, the terrain heights too; the contact
// cache with its warm-start impulses (mw::Cold::slot, 7 KB) and the step's manifold pool stay in HBM / L2 and are read and written in
// place, by the one lane that owns the body.  Constraints that share a body keep their serial order on every path, so the result equals
// the serial CPU build of the same source bit for bit.
//
// PARITY UNPINNED (Box2D is not available to pin against) -- see multiwalker_core.hpp.
#include "common.hpp"
#include "multiwalker_class.hpp"
#if defined(MADRL_MW_TIMING) && MW_CAPW != 4
#undef MADRL_MW_TIMING   // the measurement build instruments the four-walker class only
#endif
#if defined(MADRL_MW_TIMING)
// measurement build (scripts/variants.sh, never the shipped library): s_memtime stamps of a wavefront's first lane and a few per-env
// counters, per block, for the solver launch (p = 0) and the continuous-pass launch (p = 1); read back by madrl_multiwalker_debug_read
#define MW_DBG_BLOCKS 4096
static __device__ unsigned long long g_mw_stamp[2][MW_DBG_BLOCKS][8];
static __device__ int g_mw_val[2][MW_DBG_BLOCKS][16][4];
#define MW_TSTAMP(p, k) do { if (threadIdx.x == 0 && blockIdx.x < MW_DBG_BLOCKS) g_mw_stamp[p][blockIdx.x][k] = __builtin_amdgcn_s_memtime(); } while (0)
#define MW_TVAL(p, k, v) do { if ((threadIdx.x & 3) == 0 && blockIdx.x < MW_DBG_BLOCKS) g_mw_val[p][blockIdx.x][threadIdx.x >> 2][k] = (int)(v); } while (0)   // (four lanes per env)
// shader clocks a wavefront spends in region k of the continuous pass's chains, summed over the step (the first ACTIVE lane adds them up:
// the regions run under exec masks)
static __device__ unsigned long long g_mw_acc[MW_DBG_BLOCKS][8];
#define MW_TACC_T0(v) unsigned long long v = __builtin_amdgcn_s_memtime()
#define MW_TACC(k, v) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime();                                              \
        if ((int)threadIdx.x == __ffsll((long long)__ballot(1)) - 1 && blockIdx.x < MW_DBG_BLOCKS) atomicAdd(&g_mw_acc[blockIdx.x][k], n_ - v); \
        v = n_; } while (0)
#endif
#define MW_LDS __attribute__((address_space(3)))   // mw::step_solve: its working copies of manifolds past the registers are in LDS
#include "multiwalker_core.hpp"

#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

// (a named namespace per class: the kernels of the three classes carry their class in their names -- profiles tell them apart)
#define MW_KNS MW_CAT_(mwk_c, MW_CAPW)
namespace MW_KNS {

using namespace madrl;

struct MwDev {
    mw::EnvCfg cfg;
    uint32_t gid_base;
    int32_t world_dw;      // dwords per env in the state buffer: mw::World, then the step's manifold pool
    int32_t scratch_off_dw; // where the pool starts inside an env's block
    uint8_t *pending;      // [n_envs] at the end of the state buffer: this env runs the trailing step of a reset in pass 1
    int32_t scratch_bytes; // the pool: Model::max_manifolds manifolds, 16-byte aligned
    int32_t ty_bytes;      // the terrain heights of one env, 16-byte aligned
    int32_t toi_lane0_bytes;       // time-of-impact cache of lane 0 (the package's contact slots)
    int32_t cold_q;        // 16-byte words of mw::Cold in use (up to the last contact slot of this walker count)
    int32_t lds_stride[4]; // LDS per env: Hot | terrain | the solver's part of Scratch | work area of the phase; [0] all phases in one launch, [1..3] collide, solve, continuous pass
    int64_t n_envs;
    const mw::Model *model;
    uint32_t *state;
    // the NEXT episode of every env, prepared ahead of time (see the launch sequence below)
    uint32_t *spare_state; // [n_envs] records like `state`
    float *spare_obs;      // [n_envs][W][D]: the observation MultiWalkerEnv.reset returns for that episode
    uint32_t *ready_step;  // [n_envs] 0: the spare is stale / consumed, SPARE_BUSY: being prepared, else the step() call that finished it
    uint32_t *dirty_list;  // [2][n_envs] envs whose spare is stale: list (step_id & 1) is rebuilt by step() call step_id, which appends to the other
    uint32_t *n_dirty;     // [2]
    uint32_t step_id;      // this step() call (starts at 1; reset calls carry the last one)
    int32_t spare_blocks;  // leading blocks of a step launch that work on spares
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
#define MADRL_MW_SOLVE_WAVES 1   // resident wavefronts per SIMD the launches' registers are allocated for
#endif
#ifndef MADRL_MW_SOLVE_MREG
#define MADRL_MW_SOLVE_MREG 3    // manifolds a solver lane holds in registers for the whole solve
#endif
#ifndef MADRL_MW_SOLVE_OVERFLOW
#define MADRL_MW_SOLVE_OVERFLOW 5   // manifolds per env the solver launch can hold in LDS on top of the lanes' register copies
#endif
// which phases a launch runs (see mw_step_kernel)
enum { PH_COLLIDE = 1, PH_SOLVE = 2, PH_TOI = 4, PH_ALL = 7 };
// THE SOLVER LAUNCH OF THE SIXTEEN-LANE CLASS RUNS TWO WAVEFRONTS PER SIMD (round 6).  With 9 / 10 walkers a wavefront holds 4 envs: 16 384
// envs are 4 096 wavefronts, four rounds over the chip at one wavefront per SIMD -- the launch is bound by how much of a SIMD's time ONE
// wavefront of dependent float32 updates can use (about half: the rest waits for LDS round trips and branches), so a second resident
// wavefront is worth more than manifolds in registers.  That launch therefore keeps NO manifold in registers (228 VGPRs instead of 385, no
// scratch): every position of a sweep works on a register copy fetched from the LDS working copies in one go (mw::step_solve,
// MW_POOL_VELOCITY), of which it holds 14 per env instead of 5 (20 KB of LDS per wavefront: eight wavefronts per CU).  The other classes
// have one wavefront per SIMD's worth of work at the BASELINE batch (16 envs per wavefront) and keep three manifolds per lane in registers.
#ifndef MADRL_MW_SOLVE_2W
#define MADRL_MW_SOLVE_2W (MW_NLANES >= 16)
#endif
constexpr int waves_of(int ph) { return (ph == PH_SOLVE && MADRL_MW_SOLVE_2W) ? 2 : MADRL_MW_SOLVE_WAVES; }
constexpr int mreg_of(int ph) { return (ph == PH_SOLVE && MADRL_MW_SOLVE_2W) ? 0 : MADRL_MW_SOLVE_MREG; }
#ifndef MADRL_MW_SOLVE_2W_COPIES
#define MADRL_MW_SOLVE_2W_COPIES 14   // LDS working copies per env of that launch
#endif
constexpr int overflow_of(int ph) { return (ph == PH_SOLVE && MADRL_MW_SOLVE_2W) ? MADRL_MW_SOLVE_2W_COPIES : MADRL_MW_SOLVE_OVERFLOW; }
constexpr int EPW = 64 / mw::SOLVE_LANES;   // envs per wavefront: one lane per walker
constexpr int NL = mw::SOLVE_LANES;
constexpr int HOT_BYTES = (int)((sizeof(mw::Hot) + 15) / 16 * 16);
constexpr int SCR_HDR_BYTES = (int)((offsetof(mw::Scratch, m) + 15) / 16 * 16);              // Scratch without the pool
static_assert(offsetof(mw::Scratch, m) % 16 == 0, "the manifold pool follows the header at a 16-byte boundary");
constexpr int SOLVE_HDR_BYTES = (int)((offsetof(mw::Scratch, m_bA) + 15) / 16 * 16);         // the part of Scratch the solver and the continuous pass work on
constexpr int TOI_WORK_BYTES = (int)((sizeof(mw::ToiWork) + 15) / 16 * 16);
constexpr int TOI_LANE_BYTES = (mw::EDGE_SLOTS_HULL * 5 + 15) / 16 * 16;   // time-of-impact cache of a walker's body: 4 + 1 bytes per contact slot

__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the cooperating lanes of multiwalker_core.hpp's `Par` = one group of NL (4, 8 or 16) neighbouring lanes of the wavefront
template <int MREG_>
struct GroupPar {
    static constexpr int SOLVE_EMU = 1;                    // mw::step_solve: this lane IS one solver lane
    static constexpr int MREG = MREG_;
    int l;
    // Wave-uniform values from which a lane finds its env's record again with nothing but its lane id (sixteen-lane class: no per-lane
    // pointer has to stay live across the solver's sweeps, see HAVE_FUSED below): the records (live or spare), the list that maps a
    // spare slot to its env (or nullptr), the first env of this wavefront, the record stride and where the manifold pool starts in it
    const uint32_t *rec_base;
    const uint32_t *slot_env;
    int64_t first_env;
    int32_t world_dw, pool_off_dw;
    __device__ __forceinline__ int64_t env_again() const {
        const int lid = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // the lane id, whatever exec is
        int64_t env = first_env + lid / NL;
        if (slot_env) env = slot_env[env];
        return env;
    }
    __device__ __forceinline__ uint32_t *rec_again() const { return const_cast<uint32_t *>(rec_base) + env_again() * (int64_t)world_dw; }
    __device__ __forceinline__ mw::ColdView cold_again(const mw::ColdView &c) const {
        if constexpr (NL < 16) return c;
        else { mw::ColdView v = mw::cold_view(*reinterpret_cast<mw::Cold *>(rec_again() + sizeof(mw::Hot) / 4)); v.ty = c.ty; return v; }
    }
    __device__ __forceinline__ mw::Manifold *pool_again(mw::Manifold *p) const {
        if constexpr (NL < 16) return p;
        else return reinterpret_cast<mw::Manifold *>(rec_again() + pool_off_dw);
    }
    __device__ __forceinline__ int solve_lane(int) const { return l; }
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ int n() const { return NL; }
    __device__ __forceinline__ void sync() const { lds_sync(); }
    __device__ __forceinline__ int alloc(int *counter) const { return atomicAdd(counter, 1); }
    __device__ __forceinline__ void or_bits(uint32_t *p, uint32_t v) const { atomicOr(p, v); }
    __device__ __forceinline__ void or_bits(mw::Bits32 *p, mw::Bits32 v) const { atomicOr(&p->v, v.v); }
    __device__ __forceinline__ void or_bits(mw::Bits64 *p, mw::Bits64 v) const { atomicOr(reinterpret_cast<unsigned long long *>(&p->v), (unsigned long long)v.v); }
    // OR over the lanes of the env -- a quad, half a row or a row of the wavefront: DPP permutes, no LDS.  After the two quad permutes
    // the four lanes of a quad agree, so mirroring half a row (lane i <-> 7 - i) brings in the other quad, and mirroring the row
    // (i <-> 15 - i) the other half.
    __device__ __forceinline__ uint32_t reduce_or(uint32_t v) const {
        static_assert(NL == 4 || NL == 8 || NL == 16, "one env = a quad, half a row or a row of lanes");
        v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
        v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
        if (NL >= 8) v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);   // row_half_mirror
        if (NL >= 16) v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
        return v;
    }
};

// ONE LAUNCH PER b2World::Step.  A wavefront takes its 16 envs through the whole step:
//   apply_action, b2ContactManager::Collide, islands + solver schedule                                (mw::step_collide)
//   b2Island::Solve, sleeping                                                                        (mw::step_solve)
//   SynchronizeFixtures + FindNewContacts, b2World::SolveTOI, observation / reward / done             (mw::step_post, solve_toi, env_observe)
// with mw::Hot, the terrain heights and the step's schedule in LDS from beginning to end (the manifold pool of the step lives in the
// state buffer next to the env's record).  How long a phase takes varies a lot between envs -- a lying package, a leg arriving at the
// ground -- and a launch ends with its slowest wavefront: one launch pays that tail once, three launches paid it three times.
//
// AUTO-RESET WITHOUT A SECOND PASS.  With the Philox contract a reset's world depends on the env and on how many episodes it has had,
// not on when the previous episode ended -- so it can be built BEFORE it is needed.  Every env has a spare record holding its next
// episode (world after reset + the trailing step, and the observation reset returns).  When a step ends an env's episode, the end of
// the launch copies the spare over the live record and hands out its observation; the env goes on a list, and the NEXT step() call
// rebuilds the listed spares in the leading blocks of its own launch (16 per wavefront).  Only an env whose spare is not ready (two
// episodes ending within two steps) takes the second launch: reset + trailing step for the envs marked `pending`, which is also what
// reset(mask) runs.  A spare finished by the current call is never taken (ready_step == step_id): which order the hardware runs the
// blocks in must not decide which path an env takes.
constexpr uint32_t SPARE_BUSY = 0xFFFFFFFFu;

// mode 0: step() -- blocks [0, spare_blocks) rebuild spares (reset + trailing step), the others step the live envs;
// mode 1: reset + trailing step of the live envs selected by io.mask (reset(mask)) or, without a mask, by `pending` (auto-reset)
// PH: which phases this launch runs (all of them, or one: the step as three launches -- every wavefront of a launch then runs the same
// loops, which is what the instruction caches, shared by the eight wavefronts of two CUs, are sized for).  Between the launches of a
// split step the schedule (the solver's part of mw::Scratch) waits in front of the manifold pool in the state buffer.
// The one-launch form and hipcc 7.2.  Up to round 5 it was not built for the sixteen-lane class: the compiler emitted faulting code for that
// kernel -- and only that one.  Root cause (round 6, profiles/r06_multiwalker/c10_fused_masked_spill.txt, readable off the emitted code
// without a GPU): the register allocator split the live range of the per-lane record pointers at the control-flow join that follows the
// joints' InitVelocityConstraints and emitted its copies into accumulation registers at the head of the join block, AHEAD of the
// `s_or_b64 exec` that re-enables the lanes the region had masked off -- the lanes without joints, lane >= n_walkers.  Those lanes never
// wrote their copy; after the velocity iterations the pointer was read back under the full mask and step_post's body loop, which runs on
// all sixteen lanes, loaded through the stale registers (the fault rocgdb showed: profiles/r05_multiwalker/rocgdb_c10_fused.txt).  The
// source read nothing uninitialised: every lane computes the pointer at the top of the kernel.
// What is built now keeps no per-lane pointer (and no joint id) live across that join: after the sweeps every lane finds its env's record
// again from its lane id and wave-uniform values (GroupPar::rec_again, mw::step_solve's cold_again / pool_again).  The one-launch kernel of
// every class is then clean under scripts/find_masked_spills.py -- tests/test_kernel_metadata.py compiles the three classes and runs that
// scan, so a later change that brings such a copy back fails the CPU suite -- and matches the CPU build byte for byte in every lane
// (tests/test_multiwalker_gpu.py, the "fused" cases at 3, 8, 9 and 10 walkers).
constexpr bool HAVE_FUSED = true;
template <int PH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(waves_of(PH), waves_of(PH))))
void mw_step_kernel(const MwDev d, const MwIO io, const int mode, const int pending_only) {
    const mw::Model &M = *d.model;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = threadIdx.x / NL, lane = threadIdx.x % NL;
    constexpr int PHI = PH == PH_ALL ? 0 : (PH == PH_COLLIDE ? 1 : (PH == PH_SOLVE ? 2 : 3));
    constexpr bool TERRAIN = PH != PH_SOLVE;   // the solver never looks at the terrain
    const int tyb = TERRAIN ? d.ty_bytes : 0;
    unsigned char *base = smem + g * d.lds_stride[PHI];
    mw::Hot &Wd = *reinterpret_cast<mw::Hot *>(base);
    float *ty_l = reinterpret_cast<float *>(base + HOT_BYTES);
    mw::Scratch &S = *reinterpret_cast<mw::Scratch *>(base + HOT_BYTES + tyb);   // its header; the pool stays in the state buffer
    // after the solver's part of the header: the collide phase's manifold summaries and the actions | the solver's LDS copies of
    // manifolds | the continuous pass's work areas -- one after the other in the same bytes
    unsigned char *work = base + HOT_BYTES + tyb + SOLVE_HDR_BYTES;
    float *s_rew = reinterpret_cast<float *>(work);   // (written after the continuous pass is done with the work area)
    uint32_t *s_done = reinterpret_cast<uint32_t *>(s_rew + mw::MAX_WALKERS);
    constexpr int RW_OFF = (4 * mw::MAX_WALKERS + 4 + 7) / 8 * 8 < 32 ? 32 : (4 * mw::MAX_WALKERS + 4 + 7) / 8 * 8;   // env_observe's doubles, after the two above
    const int W = M.W;
    const uint32_t cur = d.step_id & 1u, nxt = cur ^ 1u;
    const bool spare = mode == 0 && (int)blockIdx.x < d.spare_blocks;
    int64_t env = ((int64_t)blockIdx.x - (mode == 0 && !spare ? d.spare_blocks : 0)) * EPW + g;
    bool active;
    if (spare) {   // slot -> the env whose spare is to be rebuilt
        active = env < (int64_t)d.n_dirty[cur];
        if (active) env = d.dirty_list[cur * d.n_envs + env];
    } else {
        active = env < d.n_envs;
        if (active && mode == 1) active = pending_only ? d.pending[env] != 0 : (io.mask ? io.mask[env] != 0 : true);
    }
    if (!active) return;   // (a whole group: the lanes that stay only ever synchronise inside their wavefront)
    const GroupPar<mreg_of(PH)> par{lane, spare ? d.spare_state : d.state, spare ? d.dirty_list + cur * d.n_envs : nullptr,
                       ((int64_t)blockIdx.x - (mode == 0 && !spare ? d.spare_blocks : 0)) * EPW, d.world_dw, d.scratch_off_dw + SOLVE_HDR_BYTES / 4};
    const bool fresh = spare || mode == 1;   // reset first, then the trailing zero-action step (:357)
    if (PH == PH_SOLVE) MW_TSTAMP(0, 0);
    if (PH == PH_TOI) MW_TSTAMP(1, 0);
    uint32_t *rec = (spare ? d.spare_state : d.state) + env * (int64_t)d.world_dw;
    mw::Cold *cold_g = reinterpret_cast<mw::Cold *>(rec + sizeof(mw::Hot) / 4);
    mw::ColdView Cd = mw::cold_view(*cold_g);
    uint32_t *sched_g = rec + d.scratch_off_dw;   // the schedule between the launches of a split step
    mw::Manifold *pool = reinterpret_cast<mw::Manifold *>(sched_g + SOLVE_HDR_BYTES / 4);
    const uint32_t gid = d.gid_base + (uint32_t)env;
    {
        uint32_t *dst = reinterpret_cast<uint32_t *>(&Wd);
        for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) dst[k] = rec[k];
    }
    if (!(PH & PH_COLLIDE)) {
        uint32_t *sd = reinterpret_cast<uint32_t *>(&S);
        for (int k = lane; k < SOLVE_HDR_BYTES / 4; k += NL) sd[k] = sched_g[k];
    }
    lds_sync();
    if ((PH & PH_COLLIDE) && fresh) {
        if (lane == 0) {
            if (spare) {   // the episode the live env will start next
                Wd.episode = reinterpret_cast<const mw::Hot *>(d.state + env * (int64_t)d.world_dw)->episode;
                mw::env_reset_world(M, d.cfg, Wd, Cd, gid);
                d.ready_step[env] = SPARE_BUSY;
            } else {
                mw::env_reset_world(M, d.cfg, Wd, Cd, gid, io.inj_terrain ? io.inj_terrain + env * M.NT : nullptr, io.inj_push ? io.inj_push + env * W : nullptr);
                if (d.ready_step[env] != 0) {   // the spare held this episode: stale now, to be rebuilt by the next step() call
                    d.ready_step[env] = 0;
                    d.dirty_list[nxt * d.n_envs + atomicAdd(&d.n_dirty[nxt], 1u)] = (uint32_t)env;
                }
            }
        }
        lds_sync();
    }
    if (TERRAIN) {   // read over and over by the narrow phase, the root finder, the lidar
        for (int k = lane; k < M.NT; k += NL) ty_l[k] = cold_g->ty[k];
        Cd.ty = ty_l;
    }
    if (PH & PH_COLLIDE) {
        float *s_act = reinterpret_cast<float *>(work + SCR_HDR_BYTES - SOLVE_HDR_BYTES);
        for (int k = lane; k < 4 * mw::MAX_WALKERS; k += NL) s_act[k] = (!fresh && k < 4 * W) ? io.actions[env * 4 * W + k] : 0.0f;
        lds_sync();
        mw::env_apply_actions(M, Wd, Cd, par, s_act);
        mw::step_collide(M, Wd, Cd, S, pool, par);
        lds_sync();
    }
    // every lane copies the manifolds it owns from the pool into registers (and LDS: the bytes the collide phase's summaries were in)
    if (PH & PH_SOLVE) { mw::step_solve(M, Wd, Cd, S, pool, reinterpret_cast<mw::Manifold *>(work), overflow_of(PH), par); lds_sync(); }
    if constexpr (NL >= 16 && PH == PH_ALL) {   // the record found again from the lane id: nothing of the above stays live over the solver (GroupPar)
        env = par.env_again();
        rec = par.rec_again();
        cold_g = reinterpret_cast<mw::Cold *>(rec + sizeof(mw::Hot) / 4);
        Cd = par.cold_again(Cd);
        sched_g = rec + d.scratch_off_dw;
        pool = par.pool_again(pool);
    }
    if (PH != PH_ALL && !(PH & PH_TOI)) {   // hand the step on to the next launch
        if (PH & PH_COLLIDE) { const uint32_t *sd = reinterpret_cast<const uint32_t *>(&S); for (int k = lane; k < SOLVE_HDR_BYTES / 4; k += NL) sched_g[k] = sd[k]; }
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&Wd);
        for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) rec[k] = src[k];
        if (PH == PH_SOLVE) MW_TSTAMP(0, 6);
        return;
    }
    mw::step_post(M, Wd, Cd, S, par);
    if (PH == PH_TOI) MW_TSTAMP(1, 1);
    if (M.continuous) {
        // per lane: the time-of-impact cache of the body it works on (lane 0 may hold the package: the largest contact cache) and room
        // for the manifolds of a mini island past the four in registers -- in the manifold pool, free by now: most of it for lane 0,
        // a few entries for every other lane
        mw::ToiLaneWork TL;
        unsigned char *lc = work + TOI_WORK_BYTES + (lane == 0 ? 0 : d.toi_lane0_bytes + (lane - 1) * TOI_LANE_BYTES);
        const int lcap = lane == 0 ? d.toi_lane0_bytes / 5 : TOI_LANE_BYTES / 5;
        TL.alpha = reinterpret_cast<float *>(lc); TL.meta = lc + 4 * lcap;
        constexpr int CO = NL >= 16 ? 2 : 4;   // (a leg or a hull touches two or three edges; the four in registers come first)
        const int c0 = M.max_manifolds - CO * ((NL < M.NB ? NL : M.NB) - 1);   // (lanes past the last body own nothing)
        TL.ovf = lane == 0 ? pool : pool + c0 + CO * (lane - 1);
        TL.ovf_cap = lane == 0 ? c0 : CO;
        mw::solve_toi(M, Wd, Cd, S, *reinterpret_cast<mw::ToiWork *>(work), TL, par, 1.0f / mw::FPS);
    }
    if (PH == PH_TOI) MW_TSTAMP(1, 4);
    const int OD = W * mw::obs_dim_of(d.cfg);
    float *obs_row = (spare ? d.spare_obs : io.obs) + env * OD;  // observation rows go straight to HBM
    if (lane == 0) *s_done = 0;
    mw::env_observe(M, d.cfg, Wd, Cd, par, gid, obs_row, !fresh ? s_rew : (float *)nullptr, !fresh ? reinterpret_cast<uint8_t *>(s_done) : (uint8_t *)nullptr,
                    reinterpret_cast<double *>(work + RW_OFF));   // walker w's row by lane w
    if (lane == 0) {
        Wd.t += 1;
        Wd.tick += 1;
        if (!fresh) {
            if (d.cfg.max_steps > 0 && Wd.t >= d.cfg.max_steps) *s_done |= 2;
            uint32_t take = 0;
            if (d.cfg.auto_reset && *s_done != 0) {
                const uint32_t rs = d.ready_step[env];
                // the spare must hold THIS record's next episode: a caller that restored or teacher-forced the live records through
                // state_buffer (Hot::episode included) leaves spares built for another episode behind -- those take the second launch
                const uint32_t sp_episode = reinterpret_cast<const mw::Hot *>(d.spare_state + env * (int64_t)d.world_dw)->episode;
                take = (rs != 0 && rs != SPARE_BUSY && rs != d.step_id && sp_episode == Wd.episode + 1u) ? 1u : 2u;   // 1: the spare is ready, 2: the second launch
            }
            d.pending[env] = take == 2 ? 1 : 0;
            *s_done |= take << 8;
        } else {
            Wd.t = 0;   // the reset's trailing step does not count (:357)
            if (spare) d.ready_step[env] = d.step_id;
            else d.pending[env] = 0;
        }
    }
    lds_sync();
    if (!fresh) {
        if (lane < W) io.rew[env * W + lane] = s_rew[lane];
        if (lane == 0) io.done[env] = (uint8_t)(*s_done | (Wd.overflow ? 0x80u : 0u));   // bit 7: this episode ran out of a capacity (sticky, Hot::overflow)
        if ((*s_done >> 8) == 1) {   // the episode ended and the next one is ready: it becomes the live record
            const uint32_t *sp = d.spare_state + env * (int64_t)d.world_dw;
            uint32_t *dst = reinterpret_cast<uint32_t *>(&Wd);
            for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) dst[k] = sp[k];   // (written back to the live record below)
            const uint4 *cs = reinterpret_cast<const uint4 *>(sp + sizeof(mw::Hot) / 4);
            uint4 *cdst = reinterpret_cast<uint4 *>(rec + sizeof(mw::Hot) / 4);
            for (int k = lane; k < d.cold_q; k += NL) cdst[k] = cs[k];
            const float *so = d.spare_obs + env * OD;
            for (int k = lane; k < OD; k += NL) obs_row[k] = so[k];
            if (lane == 0) {
                d.ready_step[env] = 0;
                d.dirty_list[nxt * d.n_envs + atomicAdd(&d.n_dirty[nxt], 1u)] = (uint32_t)env;
            }
        }
    }
    lds_sync();
    {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&Wd);
        for (int k = lane; k < (int)(sizeof(mw::Hot) / 4); k += NL) rec[k] = src[k];
    }
    if (PH == PH_TOI) MW_TSTAMP(1, 5);
}

// first thing in a step() call: the list the call's own launch will append to starts empty
__global__ void mw_begin_step_kernel(const MwDev d) { if (threadIdx.x == 0 && blockIdx.x == 0) d.n_dirty[(d.step_id & 1u) ^ 1u] = 0; }
__global__ void mw_init_spares_kernel(const MwDev d) {   // every spare is stale: all envs on the list of the first step() call (step_id 1)
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env < d.n_envs) { d.ready_step[env] = 0; d.dirty_list[d.n_envs + env] = (uint32_t)env; }
    if (env == 0) { d.n_dirty[0] = 0; d.n_dirty[1] = (uint32_t)d.n_envs; }
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
            p[4] = wr->c.sleep_time[b]; p[5] = w->awake.test(b) ? 1.0f : 0.0f;
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

struct MwHandle {
    madrl_multiwalker_config cfg;
    MwDev dev;
    int device;
    int64_t max_blocks;
    uint32_t step_id;
    int use_spares, fused;
    void *model_dev;
    int NB, NT;
};

size_t mw_scratch_bytes(const mw::Model &M) {   // the schedule between launches, then the manifold pool of a step (Model::max_manifolds entries)
    return (size_t)SOLVE_HDR_BYTES + align_up((size_t)M.max_manifolds * sizeof(mw::Manifold), 16);
}

int mw_validate(const madrl_multiwalker_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_multiwalker_config))
        return fail(MADRL_EINVAL, "madrl_multiwalker_config.struct_size=%d, library expects %d", c->struct_size,
                    (int)sizeof(madrl_multiwalker_config));
    if (c->n_walkers < 1 || c->n_walkers > mw::MAX_WALKERS)   // (the C ABI picked this class by n_walkers: multiwalker.hip)
        return fail(MADRL_EINVAL, "n_walkers=%d unsupported by the %d-walker class", c->n_walkers, mw::MAX_WALKERS);
    return MADRL_OK;
}

void mw_launch_step(const MwHandle *h, const MwDev &d, const MwIO &io, int mode, int pending_only, hipStream_t s) {
    const unsigned blocks = (unsigned)((d.n_envs + EPW - 1) / EPW + d.spare_blocks);   // every group of EPW envs gets its own wavefront
    if (h->fused && HAVE_FUSED) {
        hipLaunchKernelGGL(mw_step_kernel<HAVE_FUSED ? PH_ALL : PH_COLLIDE>, dim3(blocks), dim3(64), (size_t)EPW * d.lds_stride[0], s, d, io, mode, pending_only);
    } else {
        hipLaunchKernelGGL(mw_step_kernel<PH_COLLIDE>, dim3(blocks), dim3(64), (size_t)EPW * d.lds_stride[1], s, d, io, mode, pending_only);
        hipLaunchKernelGGL(mw_step_kernel<PH_SOLVE>, dim3(blocks), dim3(64), (size_t)EPW * d.lds_stride[2], s, d, io, mode, pending_only);
        hipLaunchKernelGGL(mw_step_kernel<PH_TOI>, dim3(blocks), dim3(64), (size_t)EPW * d.lds_stride[3], s, d, io, mode, pending_only);
    }
}
void mw_launch_all(MwHandle *h, const MwIO &io, int mode, hipStream_t s) {
    MwDev d = h->dev;
    d.spare_blocks = 0;
    d.step_id = h->step_id;
    if (mode == 1) {   // MultiWalkerEnv.step
        d.step_id = ++h->step_id;
        if (h->cfg.auto_reset && h->use_spares) {
            hipLaunchKernelGGL(mw_begin_step_kernel, dim3(1), dim3(64), 0, s, d);
            d.spare_blocks = (int32_t)((d.n_envs + EPW - 1) / EPW);   // room for every spare (a common horizon ends all episodes at once); mostly a handful do something
        }
        mw_launch_step(h, d, io, 0, 0, s);
        if (!h->cfg.auto_reset) return;
        d.spare_blocks = 0;
        mw_launch_step(h, d, io, 1, 1, s);   // envs without a ready spare
        return;
    }
    mw_launch_step(h, d, io, 1, 0, s);
}

int mw_launch(MwHandle *h, const MwIO &io, int mode, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    mw_launch_all(h, io, mode, s);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int k_obs_dim(const madrl_multiwalker_config *cfg, int32_t *out_dim) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (!out_dim) return fail(MADRL_EINVAL, "out_dim is NULL");
    *out_dim = mw::OBS_DIM - 1 + (cfg->one_hot ? mw::MAX_AGENTS_ID : 1);
    return MADRL_OK;
}

int k_state_bytes(const madrl_multiwalker_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_bytes) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_bytes non-NULL");
    mw::Model M;
    memset(&M, 0, sizeof(M));
    mw::build_model(M, cfg->n_walkers);
    // per env: the world record, then the step's Scratch (manifolds + schedule handed from launch to launch); one byte per env; then the
    // spares: a second record per env, its observation, three dwords per env (state, two lists), the lists' lengths
    const uint64_t rec = (uint64_t)(align_up(sizeof(mw::World), 16) + mw_scratch_bytes(M));
    const uint64_t od = (uint64_t)cfg->n_walkers * (uint64_t)(mw::OBS_DIM - 1 + (cfg->one_hot ? mw::MAX_AGENTS_ID : 1));
    *out_bytes = rec * (uint64_t)n_envs + align_up((uint64_t)n_envs, 16) + rec * (uint64_t)n_envs + align_up(od * 4 * (uint64_t)n_envs, 16) + 12 * (uint64_t)n_envs + 16;
    return MADRL_OK;
}

int k_create(const madrl_multiwalker_config *cfg, int64_t n_envs, int32_t device, void *state_dev,
                             void **out) {
    int rc = mw_validate(cfg);
    if (rc) return rc;
    if (!state_dev || !out || n_envs < 1) return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    if (n_envs + cfg->env_id_base > 0xFFFFFFFFll) return fail(MADRL_EINVAL, "global env index must fit 32 bits");
    MADRL_HIP_TRY(hipSetDevice(device));
    MwHandle *h = new (std::nothrow) MwHandle();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    h->max_blocks = 0;
    mw::Model M;
    memset(&M, 0, sizeof(M));
    mw::build_model(M, cfg->n_walkers);
    M.continuous = cfg->discrete_only ? 0 : 1;
    M.poly_rev = cfg->polygon_revision ? 1 : 0;
#ifdef MADRL_EXPERIMENTS   // measurement builds only (scripts/variants.sh): the production library takes nothing from the process environment
    if (const char *e = getenv("MADRL_MW_TOI")) M.continuous = atoi(e);  // 0 = no continuous pass, 2 = candidates only
#endif
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
    {
        unsigned char *p = d.pending + align_up((size_t)n_envs, 16);
        d.spare_state = (uint32_t *)p; p += (size_t)d.world_dw * 4 * (size_t)n_envs;
        d.spare_obs = (float *)p; p += align_up((size_t)cfg->n_walkers * (size_t)mw::obs_dim_of(d.cfg) * 4 * (size_t)n_envs, 16);
        d.ready_step = (uint32_t *)p; p += 4 * (size_t)n_envs;
        d.dirty_list = (uint32_t *)p; p += 8 * (size_t)n_envs;
        d.n_dirty = (uint32_t *)p;
    }
    d.cold_q = (int32_t)(align_up(offsetof(mw::Cold, slot) + (size_t)M.n_slots * sizeof(mw::Slot), 16) / 16);
    h->step_id = 0;
    h->use_spares = 1;
    h->fused = 0;
    d.ty_bytes = (int32_t)align_up((size_t)M.NT * 4, 16);
    d.toi_lane0_bytes = (int32_t)align_up((size_t)(M.slot_cap[0] > mw::EDGE_SLOTS_HULL ? M.slot_cap[0] : mw::EDGE_SLOTS_HULL) * 5, 16);
    {
        const int wc = SCR_HDR_BYTES - SOLVE_HDR_BYTES + (int32_t)align_up(4 * 4 * mw::MAX_WALKERS, 16);            // collide
        const int ws = overflow_of(PH_SOLVE) * (int32_t)sizeof(mw::Manifold), wsa = overflow_of(PH_ALL) * (int32_t)sizeof(mw::Manifold);   // solve: its own launch, inside the one-launch kernel
        const int wt = TOI_WORK_BYTES + d.toi_lane0_bytes + (NL - 1) * TOI_LANE_BYTES;                                // continuous pass
        const int wa = wc > wsa ? (wc > wt ? wc : wt) : (wsa > wt ? wsa : wt);
        const int common = HOT_BYTES + SOLVE_HDR_BYTES;
        d.lds_stride[0] = common + d.ty_bytes + (int32_t)align_up((size_t)wa, 16);
        d.lds_stride[1] = common + d.ty_bytes + (int32_t)align_up((size_t)wc, 16);
        d.lds_stride[2] = common + (int32_t)align_up((size_t)ws, 16);
        d.lds_stride[3] = common + d.ty_bytes + (int32_t)align_up((size_t)wt, 16);
        // env g's block starts 4 LDS banks after env g-1's (mod 32 banks): neighbouring envs of a wavefront hit different banks
        for (int k = 0; k < 4; ++k) d.lds_stride[k] = (d.lds_stride[k] + 127 - 16) / 128 * 128 + 16;
#ifdef MADRL_EXPERIMENTS
        if (const char *e = getenv("MADRL_MW_LDS_EXTRA")) { const int x = atoi(e); if (x > 0 && x <= 32768) for (int k = 0; k < 4; ++k) d.lds_stride[k] += x / 16 * 16; }   // occupancy vs LDS
#endif
    }
    d.n_envs = n_envs;
    d.model = (const mw::Model *)h->model_dev;
    d.state = (uint32_t *)state_dev;
#ifdef MADRL_EXPERIMENTS
    if (getenv("MADRL_MW_VERBOSE")) fprintf(stderr, "multiwalker: LDS per env %d (one launch) | %d %d %d (collide, solve, continuous pass)\n", d.lds_stride[0], d.lds_stride[1], d.lds_stride[2], d.lds_stride[3]);
#endif
    hipLaunchKernelGGL(mw_init_spares_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { (void)hipFree(h->model_dev); delete h; return fail(MADRL_EHIP, "state buffer too small or not device memory"); }
    *out = h;
    return MADRL_OK;
}

void k_destroy(MwHandle *h) {
    if (!h) return;
    if (h->model_dev) (void)hipFree(h->model_dev);
    delete h;
}

int k_set_mode(MwHandle *h, int32_t fused, int32_t use_spares) {
    if (!h || (fused & ~1) || (use_spares & ~1)) return fail(MADRL_EINVAL, "set_mode: handle is NULL or a flag is not 0 / 1");
    h->fused = fused;
    h->use_spares = use_spares;
    return MADRL_OK;
}

#if defined(MADRL_MW_TIMING)
int k_debug_read(unsigned long long *stamps_host, int *vals_host) {   // [2][4096][8], [2][4096][16][4]
    MADRL_HIP_TRY(hipDeviceSynchronize());
    MADRL_HIP_TRY(hipMemcpyFromSymbol(stamps_host, HIP_SYMBOL(g_mw_stamp), sizeof(g_mw_stamp)));
    MADRL_HIP_TRY(hipMemcpyFromSymbol(vals_host, HIP_SYMBOL(g_mw_val), sizeof(g_mw_val)));
    return MADRL_OK;
}
int k_debug_read_acc(unsigned long long *acc_host, int reset) {   // [4096][8]; reset != 0: zero the accumulators afterwards
    MADRL_HIP_TRY(hipDeviceSynchronize());
    MADRL_HIP_TRY(hipMemcpyFromSymbol(acc_host, HIP_SYMBOL(g_mw_acc), sizeof(g_mw_acc)));
    if (reset) { static unsigned long long zero[MW_DBG_BLOCKS][8]; MADRL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mw_acc), zero, sizeof(zero))); }
    return MADRL_OK;
}
#endif

int k_dims(const MwHandle *h, int32_t *n_bodies, int32_t *n_terrain) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (n_bodies) *n_bodies = h->NB;
    if (n_terrain) *n_terrain = h->NT;
    return MADRL_OK;
}

int k_record_bytes(const MwHandle *h, int32_t *stride_bytes, int32_t *world_bytes) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (stride_bytes) *stride_bytes = h->dev.world_dw * 4;
    if (world_bytes) *world_bytes = (int32_t)sizeof(mw::World);
    return MADRL_OK;
}

int k_reset(MwHandle *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    MwIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.obs = obs_dev;
    return mw_launch(h, io, 0, stream);
}

int k_reset_with(MwHandle *h, const uint8_t *mask_dev, const double *terrain_dev, const double *push_dev,
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

int k_get_state(MwHandle *h, float *bodies_dev, float *joints_dev, float *aux_dev, uint8_t *flags_dev,
                                float *terrain_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_get_state_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, joints_dev, aux_dev,
                       flags_dev, terrain_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int k_set_state(MwHandle *h, const float *bodies_dev, const float *joints_dev, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_set_state_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, joints_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int k_step(MwHandle *h, const float *actions_dev, float *obs_dev, float *rew_dev,
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

int k_get_bodies(MwHandle *h, float *bodies_dev, uint8_t *flags_dev, float *terrain_dev,
                                 void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 63) / 64);
    hipLaunchKernelGGL(mw_get_bodies_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, h->dev, bodies_dev, flags_dev,
                       terrain_dev);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

// the handle crosses the class boundary as void *
int a_create(const madrl_multiwalker_config *cfg, int64_t n_envs, int32_t device, void *state_dev, void **out) { return k_create(cfg, n_envs, device, state_dev, out); }
void a_destroy(void *h) { k_destroy((MwHandle *)h); }
int a_set_mode(void *h, int32_t fused, int32_t use_spares) { return k_set_mode((MwHandle *)h, fused, use_spares); }
int a_dims(const void *h, int32_t *nb, int32_t *nt) { return k_dims((const MwHandle *)h, nb, nt); }
int a_record_bytes(const void *h, int32_t *stride, int32_t *world) { return k_record_bytes((const MwHandle *)h, stride, world); }
int a_reset(void *h, const uint8_t *mask, float *obs, void *stream) { return k_reset((MwHandle *)h, mask, obs, stream); }
int a_reset_with(void *h, const uint8_t *mask, const double *terrain, const double *push, float *obs, void *stream) { return k_reset_with((MwHandle *)h, mask, terrain, push, obs, stream); }
int a_get_state(void *h, float *bodies, float *joints, float *aux, uint8_t *flags, float *terrain, void *stream) { return k_get_state((MwHandle *)h, bodies, joints, aux, flags, terrain, stream); }
int a_set_state(void *h, const float *bodies, const float *joints, void *stream) { return k_set_state((MwHandle *)h, bodies, joints, stream); }
int a_step(void *h, const float *actions, float *obs, float *rew, uint8_t *done, void *stream) { return k_step((MwHandle *)h, actions, obs, rew, done, stream); }
int a_get_bodies(void *h, float *bodies, uint8_t *flags, float *terrain, void *stream) { return k_get_bodies((MwHandle *)h, bodies, flags, terrain, stream); }

}  // namespace MW_KNS
using namespace MW_KNS;

#if !defined(__HIP_DEVICE_COMPILE__)   // (a table of host functions: nothing for the device pass, which would emit any const global)
extern const madrl::MwClassApi MW_CAT_(madrl_mw_class_c, MW_CAPW);
const madrl::MwClassApi MW_CAT_(madrl_mw_class_c, MW_CAPW) = {
    mw::MAX_WALKERS, NL, k_obs_dim, k_state_bytes, a_create, a_destroy, a_set_mode, a_dims, a_record_bytes, a_reset, a_reset_with, a_get_state,
    a_set_state, a_step, a_get_bodies,
#if defined(MADRL_MW_TIMING)
    k_debug_read, k_debug_read_acc,
#else
    nullptr, nullptr,
#endif
};
#endif
