// waterworld.hip -- batched MAWaterWorld for MI355X (gfx950 / CDNA4), float32.
//
// One wavefront owns one env at a time (64-thread workgroups, persistent, striding over
// envs).  The env's particles (pursuers | evaders | poisons: position + velocity), the
// obstacle and the assembled observation rows live in LDS; HBM sees one packed state record
// in / out, the action row in and observation / reward / done / info rows out.
//
// Lane roles change per phase:
//   particle phases   lane j < NP owns particle j (integration, walls, obstacle rebound,
//                     respawn, evader/poison motion);
//   collision phase   lane = (pursuer, evader) / (pursuer, poison) pair;
//   sensing phase     lane = (pursuer, sensor) pair, three passes of 64 pairs held in registers; the objects are
//                     broadcast once each from the owning lane's registers (v_readlane -> SGPR operands) and a
//                     conservative reach mask skips, per pass, the objects none of its pursuers can sense --
//                     the only O(Np*K*N) part (~4k ray tests before the cull).
// No dense contraction -> no MFMA.  ~40 kFLOP and 5.2 KB of HBM traffic per env-step: the
// kernel is VALU-issue bound, not HBM bound.
//
// Reference semantics (file:line under /root/reference/madrl_environments/pursuit/waterworld.py):
//   step phases ........ MAWaterWorld.step :220-436      sensing ...... Archea.sensed :64-72
//   catch rule ......... _caught :180-193                 respawn ...... _respawn :139-142, :355-374
//   reset .............. :144-172 (ends with a zero-action step, W11)
// Arithmetic is float32 (north_star tolerance 1e-5 against the float64 reference); every
// expression keeps the statement order of the reference's step() so that a float32 CPU restatement agrees bit for bit.
#include "common.hpp"

#include <math.h>
#include <new>
#include <string.h>
#include <vector>

namespace {

using namespace madrl;

enum : uint32_t { WW_TAG_RESPAWN = 16, WW_TAG_RESET = 17, WW_TAG_OBSTACLE = 18 };

struct WwDev {
    int32_t Np, Ne, Npo, NP, K, D, nfeat;
    int32_t n_coop, addid, speed_features, reward_global, obstacle_fixed, max_steps, auto_reset;
    int32_t rec_dw;  // dwords per packed state record: pos[NP][2] vel[NP][2] obst[2] t tick
    uint32_t k0, k1, gid_base;
    float r_pu, r_ev, r_po, obst_r, ev_speed, poison_speed, sensor_range, action_scale;
    float poison_reward, food_reward, encounter_reward, control_penalty;
    float obst_x, obst_y;
    // sq_*: the largest float32 x with sqrtf(x) <= threshold, so that "distance <= threshold" is the single compare "dx*dx + dy*dy <= sq"
    // with the same truth value for every input (sqrtf is monotonic and correctly rounded); the correctly rounded sqrtf itself is a
    // 20-instruction sequence.  Thresholds: obstacle rebound per particle kind (:247-270), pursuer-evader / pursuer-poison contact (:272-293).
    float sq_obst_pu, sq_obst_ev, sq_obst_po, sq_hit_ev, sq_hit_po;
    int64_t n_envs;
    const float *sensors;  // [K][2]
    float *state;
};

// Fused StandardizedEnv (madrl_environments/__init__.py:204-311): when `obs_out` is set, the observation row is normalised as
// it leaves LDS -- per env, per agent, per element exponential running mean / variance in float64, exactly the arithmetic of the
// stand-alone epilogue kernel (wrappers.hip obsnorm_kernel / rewnorm_kernel) -- instead of being stored raw and read back by a
// second launch: 36 instead of 44 bytes of HBM traffic per observation element.
struct WwStd {
    double *obs_mean, *obs_var;   // [N][Np][D]
    float *obs_out;               // [N][Np][D] normalised observations; NULL = not fused
    double *rew_mean, *rew_var;   // [N][Np]
    float *rew_out;               // [N][Np] scale * (reward / (sqrt(var) + eps)); NULL = rewards are not touched
    double obs_alpha, rew_alpha, eps, scale;
    int32_t enable_obsnorm, enable_rewnorm;
};

struct WwIO {
    const uint8_t *mask;    // reset mode
    const float *actions;   // [N][Np][2]
    const float *inj_resp;  // [N][NP][4] or NULL
    float *obs;             // [N][Np][D]
    float *rew;             // [N][Np]
    uint8_t *done;          // [N]
    int32_t *info;          // [N][2]  evcatches, pocatches
    const WwStd *st;        // device copy of the fused-wrapper arguments, or NULL
};

// Launch parameters are read from the kernel-argument segment (scalar loads) at the phase that needs them instead of being held in
// SGPRs across the whole env loop: the loop's scalar live set (broadcast masks, reach sets, counters) is already at the SGPR limit,
// and what does not fit is parked in VGPR lanes at two VALU issue slots (v_writelane / v_readlane) per value and use.
struct WwKArgs {
    WwDev d;
    WwIO io;
};
typedef const __attribute__((address_space(4))) WwKArgs *WwKArgsPtr;
__device__ __forceinline__ WwKArgsPtr ww_args() {
    WwKArgsPtr p = (WwKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // a fresh pointer at every call: loads are neither merged with earlier ones nor hoisted out of the loop
    return p;
}

// A wave-uniform pointer pinned to an SGPR pair (global address space): per-lane accesses become "SGPR base + 32-bit VGPR offset"
// instead of a 64-bit VGPR address pair per array kept live across the env loop.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *uniform_ptr(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (__attribute__((address_space(1))) T *)(((uint64_t)hi << 32) | lo);
}

// Hides the loop-invariance of a lane predicate: fresh(lane) < n is one v_cmp where it is used and dies there, instead of an SGPR
// pair hoisted out of the env loop (and, past the SGPR budget, parked in a VGPR lane).
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// The same for the SPECIALISED shapes, one instruction cheaper (round 6): fresh() costs a v_mov before its v_cmp; here the compare is written
// out (volatile: it stays where it is used and its mask dies there), its right-hand side an inline constant of the shape (<= 64), and the
// wave-uniform mask becomes the lane predicate without an instruction (inverse ballot).
__device__ __forceinline__ bool lane_lt_imm(int lane, int n) {   // lane < n; n must fold to a constant in -16 .. 64
    unsigned long long m;
    asm volatile("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(m) : "i"(n), "v"(lane));
    return __builtin_amdgcn_inverse_ballot_w64(m);
}
__device__ __forceinline__ bool lane_eq_imm(int lane, int n) {
    unsigned long long m;
    asm volatile("v_cmp_eq_i32_e64 %0, %1, %2" : "=s"(m) : "i"(n), "v"(lane));
    return __builtin_amdgcn_inverse_ballot_w64(m);
}

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float u24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

__device__ __forceinline__ float dist2d(float ax, float ay, float bx, float by) {
    const float dx = ax - bx, dy = ay - by;
    return sqrtf(dx * dx + dy * dy);  // scipy cdist 'euclidean'
}

// dist2d(a, b) <= thr, with sq = sq_threshold(thr) (host side): same truth value, no square root
__device__ __forceinline__ bool dist2_le(float ax, float ay, float bx, float by, float sq) {
    const float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy <= sq;
}

// Profiling aid (scripts/variants.sh, never the shipped library): 1 no sensing loop, 2 no observation store, 4 no collisions
#ifndef MADRL_WW_ABLATE
#define MADRL_WW_ABLATE 0
#endif

// Resident wavefronts per SIMD the SPECIALISED kernel's registers are allocated for (the generic one is left to the compiler).
// Measured at BASELINE C3, 32 768 envs: 4 waves (104 VGPRs) 77 us, 5 waves (86) 60.4 us, 6 waves (80, no scratch) 57.2 us, 7 waves
// (72 + 32 B of scratch) 57.4 us.
#ifndef MADRL_WW_WAVES
#define MADRL_WW_WAVES 6
#endif
// (the fused-wrapper variant holds float64 statistics: one wave fewer, or it spills)
#define MADRL_WW_OCC_N (TNp > 0 ? (FUSED ? MADRL_WW_WAVES - 1 : MADRL_WW_WAVES) - (TNp > 6 ? 1 : 0) : 0)  // (> 6 pursuers: two words per collision matrix, more live pass state)
#define MADRL_WW_OCC __attribute__((amdgpu_waves_per_eu(MADRL_WW_OCC_N > 0 ? MADRL_WW_OCC_N : 1, MADRL_WW_OCC_N > 0 ? MADRL_WW_OCC_N : 8)))

// MODE 0: reset(mask)   MODE 1: step (+ fused auto-reset)
// TNp..TK > 0: the particle / sensor counts are compile-time constants (loops unroll, the index divisions fold); 0: generic.
// FUSED: the StandardizedEnv epilogue (WwStd) is compiled in; a template parameter because its float64 code would otherwise cost the
// plain kernel a wavefront per SIMD (132 instead of 119 VGPRs: 96 instead of 78 us per step)
template <int MODE, int TNp, int TNe, int TNpo, int TK, bool FUSED = false, int TD = 0>
__global__ __launch_bounds__(64) MADRL_WW_OCC void waterworld_kernel(const WwDev d, const WwIO io) {
    // The specialised shape has a compile-time LDS layout in a STATIC array (launched with 0 dynamic bytes): every LDS address is
    // "lane-dependent register + immediate offset".  With the dynamic array the base is a link-time symbol the compiler adds in
    // registers, hoists out of the env loop per access pattern and -- at 5 waves per SIMD -- spills.
    constexpr int SPEC_DW = TNp > 0 ? ((4 * (TNp + TNe + TNpo) + 4 + 3) / 4 * 4 + ((TNp + 1) * (TD > 0 ? TD : 1) + 3) / 4 * 4 + (2 * TK + 3) / 4 * 4) : 0;   // (TNp + 1: the spare row)
    constexpr int SPEC_BYTES = TNp > 0 ? (SPEC_DW * 4 + 8 * TNp + TNp * (TNe + TNpo) + 2 * TNe + TNpo + 15) / 16 * 16 : 16;
    static_assert(TNp == 0 || TD > 0, "a specialised shape fixes the observation width too");
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ __attribute__((aligned(16))) float smem_static[SPEC_BYTES / 4];
    float *const smem = TNp > 0 ? smem_static : smem_dyn;
    const int lane = threadIdx.x;
    const uint32_t ulane = threadIdx.x;
#define DA (ww_args()->d)
#define IOA (ww_args()->io)
    static_assert(TNp == 0 || TNp + TNe + TNpo + 1 <= 64, "lane predicates of a specialised shape compare against inline constants");
#define LANE_LT(n) (TNp > 0 ? lane_lt_imm(lane, (n)) : (fresh(lane) < (n)))
#define LANE_EQ(n) (TNp > 0 ? lane_eq_imm(lane, (n)) : (fresh(lane) == (n)))
    const int Np = TNp > 0 ? TNp : d.Np, Ne = TNp > 0 ? TNe : d.Ne, Npo = TNp > 0 ? TNpo : d.Npo, K = TNp > 0 ? TK : d.K;
    const int NP = Np + Ne + Npo, D = TD > 0 ? TD : d.D;  // TD: the observation width of the specialised shape (7 K + 3)
    // ---- LDS carve
    float *S = smem;                                    // packed record: X[NP][2] | V[NP][2] | obst[2] | t | tick
    float *X = S, *V = S + 2 * NP;
    float *OB = S + 4 * NP;
    float *O = S + (((TNp > 0 ? 4 * (TNp + TNe + TNpo) + 4 : d.rec_dw) + 3) & ~3);  // observation staging [Np][D]
    float *const O_SPARE = O + Np * D;                  // one more row: where the sensing lanes without a (pursuer, sensor) pair write
    float *SEN = O + (((Np + 1) * D + 3) & ~3);         // sensor unit vectors [K][2]
    uint64_t *NEAR = reinterpret_cast<uint64_t *>(SEN + ((2 * K + 3) & ~3));  // per pursuer: particles (bit j) / obstacle (bit NP) in sensing reach
    uint8_t *COL = reinterpret_cast<uint8_t *>(NEAR + Np);  // col_ev[Np][Ne] | col_po[Np][Npo]
    uint8_t *COLP = COL + Np * Ne;
    uint8_t *FLG = COLP + Np * Npo;                     // caught_ev[Ne] | enc_ev[Ne] | caught_po[Npo]

    for (int k = lane; k < 2 * K; k += 64) SEN[k] = d.sensors[k];
    const int rec_dw = TNp > 0 ? (4 * (TNp + TNe + TNpo) + 4 + 3) / 4 * 4 : d.rec_dw;
    const int nreg = (rec_dw + 63) >> 6;  // <= 4 (NP <= 62)

    // ---- software pipeline: next env's record + action row are fetched one env ahead
    uint32_t cur[4] = {0, 0, 0, 0};
    float cur_act = 0.0f;
    auto fetch = [&](int64_t env, uint32_t (&r)[4], float &a) {
        const auto src = uniform_ptr(reinterpret_cast<const uint32_t *>(DA.state) + env * (int64_t)rec_dw);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t k = ulane + 64u * q;
            r[q] = (q < nreg && (int)k < rec_dw) ? src[k] : 0u;
        }
        if constexpr (MODE == 1) a = (lane < 2 * Np) ? uniform_ptr(IOA.actions + env * 2 * Np)[ulane] : 0.0f;
        else a = 0.0f;
    };
    const int n_envs = (int)d.n_envs;
    if ((int)blockIdx.x < n_envs) fetch(blockIdx.x, cur, cur_act);
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur_act));
    wave_sync();

    for (int e32 = blockIdx.x; e32 < n_envs; e32 += (int)gridDim.x) {  // env indices are 32-bit (n_envs < 2^31 - grid), byte offsets 64-bit
        const int64_t env = e32;
        const int n32 = e32 + (int)gridDim.x;
        uint32_t nxt[4] = {0, 0, 0, 0};
        float nxt_act = 0.0f;
        if (n32 < n_envs) fetch(n32, nxt, nxt_act);
        bool skip = false;
        if constexpr (MODE == 0) skip = (IOA.mask != nullptr && IOA.mask[env] == 0);
        if (!skip) {
            // record -> LDS
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = lane + 64 * q;
                if (q < nreg && k < rec_dw) reinterpret_cast<uint32_t *>(S)[k] = cur[q];
            }
            wave_sync();
            int32_t tstep = reinterpret_cast<int32_t *>(S)[4 * NP + 2];
            uint32_t tick = reinterpret_cast<uint32_t *>(S)[4 * NP + 3];
            const uint32_t gid = DA.gid_base + (uint32_t)env;
            float act_lane = cur_act;  // lane 2i / 2i+1 hold pursuer i's action components

            bool do_init = (MODE == 0);
            int npass = 1;
            for (int pass = 0; pass < npass; ++pass) {
                if (do_init) {
                    // ------------------------------------------------ reset (:144-172)
                    tstep = 0;
                    if (fresh(lane) == 0) {
                        float ox = DA.obst_x, oy = DA.obst_y;
                        if (!DA.obstacle_fixed) {  // :147-148
                            const u32x4 r = philox4x32_10(gid, tick, 0u, WW_TAG_OBSTACLE, DA.k0, DA.k1);
                            ox = u24(r.x);
                            oy = u24(r.y);
                        }
                        OB[0] = ox;
                        OB[1] = oy;
                    }
                    wave_sync();
                    if (fresh(lane) < NP) {  // :153-170 each particle: uniform position, redrawn while too close to the obstacle
                        const float pr = fresh(lane) < Np ? DA.r_pu : (fresh(lane) < Np + Ne ? DA.r_ev : DA.r_po);
                        const float thr = pr * 2.0f + DA.obst_r;
                        const float ox = OB[0], oy = OB[1];
                        float x = 0.f, y = 0.f, u0 = 0.f, u1 = 0.f;
                        for (uint32_t att = 0; att < 1024u; ++att) {
                            const u32x4 r = philox4x32_10(gid, tick, (uint32_t)lane, WW_TAG_RESET | (att << 8), DA.k0, DA.k1);
                            x = u24(r.x);
                            y = u24(r.y);
                            if (att == 0) { u0 = u24(r.z); u1 = u24(r.w); }
                            if (!(dist2d(x, y, ox, oy) <= thr)) break;
                        }
                        X[2 * lane] = x;
                        X[2 * lane + 1] = y;
                        V[2 * lane] = LANE_LT(Np) ? 0.0f : (u0 - 0.5f) * DA.ev_speed;      // :164, :170 (W9)
                        V[2 * lane + 1] = LANE_LT(Np) ? 0.0f : (u1 - 0.5f) * DA.ev_speed;
                    }
                    tick += 1;
                    act_lane = 0.0f;  // reset ends with step(zeros) (:172, W11)
                    wave_sync();
                }
                // ---------------------------------------------------- step (:220-436)
                const float ox = OB[0], oy = OB[1];
                // phase A: particles
                float reward = 0.0f;
                {
                    const float a_raw0 = __shfl(act_lane, 2 * (LANE_LT(Np) ? lane : 0));
                    const float a_raw1 = __shfl(act_lane, 2 * (LANE_LT(Np) ? lane : 0) + 1);
                    const float a0 = a_raw0 * DA.action_scale, a1 = a_raw1 * DA.action_scale;  // :224
                    float pen_local = DA.control_penalty * (a0 * a0 + a1 * a1);
                    if (DA.reward_global) {  // (actions**2).sum(), row-major (:234-235, W12)
                        float s = 0.0f;
                        for (int i = 0; i < Np; ++i) {
                            const float b0 = __shfl(a0, i), b1 = __shfl(a1, i);
                            s += b0 * b0;
                            s += b1 * b1;
                        }
                        pen_local = DA.control_penalty * s;
                    }
                    if (LANE_LT(NP)) {
                        float x = X[2 * lane], y = X[2 * lane + 1], vx = V[2 * lane], vy = V[2 * lane + 1];
                        float sq_obst = DA.sq_obst_po, f = -1.0f;
                        if (LANE_LT(Np)) {
                            vx = vx + a0; vy = vy + a1;  // :229-231
                            x = x + vx; y = y + vy;
                            reward = 0.0f + pen_local;   // :233-237
                            const float cx = x < 0.f ? 0.f : (x > 1.f ? 1.f : x);  // :239-245
                            const float cy = y < 0.f ? 0.f : (y > 1.f ? 1.f : y);
                            if (x != cx) vx = 0.f;
                            if (y != cy) vy = 0.f;
                            x = cx; y = cy;
                            sq_obst = DA.sq_obst_pu; f = -0.5f;
                        } else if (LANE_LT(Np + Ne)) {
                            sq_obst = DA.sq_obst_ev; f = -0.5f;
                        }
                        if (dist2_le(x, y, ox, oy, sq_obst)) {  // dist <= pr + obst_r, :247-270 (W1, W2)
                            vx = f * vx;
                            vy = f * vy;
                        }
                        X[2 * lane] = x; X[2 * lane + 1] = y; V[2 * lane] = vx; V[2 * lane + 1] = vy;
                    }
                }
                wave_sync();
                // phase B: collisions (:272-293)
                // BITROWS (specialised shapes): a collision matrix is a few wave-uniform 64-bit masks (bit r * n + m of word w = pursuer
                // w * G + r touches particle m) made by ballots; columns are counted and rows tested with bit operations.  No byte
                // matrices in LDS, no loop over the other side of the pair.
                // A 64-bit word holds GE = floor(64 / Ne) whole rows; shapes with more pursuers use up to 4 words per matrix.
                constexpr int GE = (TNe > 0 && TNe < 64) ? 64 / TNe : 1, GP = (TNpo > 0 && TNpo < 64) ? 64 / TNpo : 1;  // rows per word
                constexpr int WE = TNp > 0 ? (TNp + GE - 1) / GE : 1, WP = TNp > 0 ? (TNp + GP - 1) / GP : 1;              // words per matrix
                constexpr bool BITROWS = TNp > 0 && TNe < 64 && TNpo < 64 && WE <= 4 && WP <= 4;
                uint64_t col_ev[WE], col_po[WP];
#pragma unroll
                for (int w = 0; w < WE; ++w) col_ev[w] = 0ull;
#pragma unroll
                for (int w = 0; w < WP; ++w) col_po[w] = 0ull;
                bool my_caught = false, my_enc = false;
                if constexpr (BITROWS) {
#pragma unroll
                    for (int w = 0; w < WE; ++w) {
                        const int li = lane / Ne, i0 = w * GE + li;
                        const bool in = LANE_LT(GE * Ne) && i0 < Np;
                        const int i = in ? i0 : 0, m = in ? lane - li * Ne : 0, j = Np + m;
                        col_ev[w] = __ballot(in && dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], DA.sq_hit_ev));
                    }
#pragma unroll
                    for (int w = 0; w < WP; ++w) {
                        const int li = lane / Npo, i0 = w * GP + li;
                        const bool in = LANE_LT(GP * Npo) && i0 < Np;
                        const int i = in ? i0 : 0, m = in ? lane - li * Npo : 0, j = Np + Ne + m;
                        col_po[w] = __ballot(in && dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], DA.sq_hit_po));
                    }
#if MADRL_WW_ABLATE & 4
                    for (int w = 0; w < WE; ++w) col_ev[w] = 0ull;
                    for (int w = 0; w < WP; ++w) col_po[w] = 0ull;
#endif
                    // _caught (:180-193): evader lanes / poison lanes count their column
                    uint64_t cm_ev = 0ull, cm_po = 0ull;  // bit r * n of every row of a word
#pragma unroll
                    for (int r = 0; r < GE; ++r) cm_ev |= 1ull << (r * Ne);
#pragma unroll
                    for (int r = 0; r < GP; ++r) cm_po |= 1ull << (r * Npo);
                    if ((!LANE_LT(Np) && LANE_LT(NP))) {
                        const bool is_ev = LANE_LT(Np + Ne);
                        const int m = is_ev ? lane - Np : lane - Np - Ne;
                        int sc = 0;
                        if (is_ev) {
#pragma unroll
                            for (int w = 0; w < WE; ++w) sc += __popcll(col_ev[w] & (cm_ev << m));
                        } else {
#pragma unroll
                            for (int w = 0; w < WP; ++w) sc += __popcll(col_po[w] & (cm_po << m));
                        }
                        my_caught = sc >= (is_ev ? DA.n_coop : 1);
                        my_enc = is_ev && sc >= 1;
                    }
                } else {
#if MADRL_WW_ABLATE & 4
                if (DA.n_envs < 0)
#endif
                for (int idx = lane; idx < Np * (Ne + Npo); idx += 64) {
                    const bool is_ev = idx < Np * Ne;
                    const int r = is_ev ? idx : idx - Np * Ne;
                    const int n2 = is_ev ? Ne : Npo;
                    const int i = r / n2, m = r % n2;
                    const int j = (is_ev ? Np : Np + Ne) + m;
                    COL[idx] = dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], is_ev ? DA.sq_hit_ev : DA.sq_hit_po);
                }
                wave_sync();
                // _caught (:180-193): evader lanes / poison lanes count their column
                if ((!LANE_LT(Np) && LANE_LT(NP))) {
                    const bool is_ev = LANE_LT(Np + Ne);
                    const int m = is_ev ? lane - Np : lane - Np - Ne;
                    const uint8_t *col = is_ev ? COL : COLP;
                    const int n2 = is_ev ? Ne : Npo;
                    int s = 0;
                    for (int i = 0; i < Np; ++i) s += col[i * n2 + m];
                    my_caught = s >= (is_ev ? DA.n_coop : 1);
                    my_enc = is_ev && s >= 1;
                    if (is_ev) { FLG[m] = my_caught; FLG[Ne + m] = my_enc; }
                    else FLG[2 * Ne + m] = my_caught;
                }
                }
                const uint64_t ev_lanes = ((Ne >= 64) ? ~0ull : ((1ull << Ne) - 1ull)) << Np;
                const uint64_t caught_mask = __ballot(my_caught);
                const uint64_t enc_mask = __ballot(my_enc);
                const int n_evc = __popcll(caught_mask & ev_lanes);
                const int n_poc = __popcll(caught_mask & ~ev_lanes);
                const int n_enc = __popcll(enc_mask);
                wave_sync();
                // phase C: sensing (:295-353).  lane = (pursuer i, sensor k)
                const float srange = DA.sensor_range, rad2 = DA.r_pu * DA.r_pu;  // W3
                // The (pursuer, sensor) pairs are spread over the lanes, PCH passes of 64 at a time; the objects they are tested
                // against are wave-uniform, so each object's position is broadcast ONCE from the register of the lane that
                // owns the particle (v_readlane -> SGPR operand) and reused by all passes: the inner loop is pure VALU, no
                // LDS round trip per (pair, object).  Arithmetic and comparison order per pair are those of the reference loop.
                // passes of 64 (pursuer, sensor) pairs held in registers at a time: no more than the specialised shape needs
                // Lane layout of a pass.  ALIGNED (compile-time K <= 64): a pass holds floor(64 / K) WHOLE pursuers (the last lanes idle), so
                // the objects a pass must visit are those in reach of 2 pursuers at BASELINE C3 instead of the 2.1-3 a pass of 64
                // consecutive (pursuer, sensor) pairs straddles: about a quarter fewer (object, pass) visits.  Otherwise: consecutive pairs.
                constexpr bool ALIGNED = TK > 0 && TK <= 64;
                constexpr int PPP = ALIGNED ? 64 / (TK > 0 ? TK : 1) : 1;  // pursuers per pass
                const int n_pass = ALIGNED ? (Np + PPP - 1) / PPP : (Np * K + 63) / 64;
                constexpr int N_PASS_T = TNp > 0 ? (ALIGNED ? (TNp + PPP - 1) / PPP : (TNp * TK + 63) / 64) : 3;
                const float part_x = LANE_LT(NP) ? X[2 * lane] : 0.f, part_y = LANE_LT(NP) ? X[2 * lane + 1] : 0.f;
                // Conservative cull: a sensor of pursuer i can only return a finite value for an object with
                // d2 <= rad2 + sv^2 <= rad2 + range^2; NEAR[i] marks the objects within that reach plus a 1e-4 relative margin
                // (d2 is computed exactly as in the test below), everything else would yield INFINITY and is skipped per pass.
                {
                    const float thr2 = (rad2 + srange * srange) * 1.0001f + 1e-9f;
                    const float mx = LANE_EQ(NP) ? ox : part_x, my = LANE_EQ(NP) ? oy : part_y;
                    for (int i = 0; i < Np; ++i) {
                        const float rx = mx - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part_x), i));
                        const float ry = my - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part_y), i));
                        const uint64_t mk = __ballot((LANE_LT(NP + 1)) && (rx * rx + ry * ry <= thr2));
                        if (LANE_EQ(0)) NEAR[i] = mk;
                    }
                    wave_sync();
                }
#if MADRL_WW_ABLATE & 1
                if (DA.n_envs < 0)
#endif
                // ONE PASS AT A TIME (round 6).  A pass walks the set bits of ITS OWN reach mask, class by class -- ascending = the reference's
                // index order: the first minimum wins as in np.argmin.  Round 5 held three passes in registers, walked the union of their
                // masks once and tested per object which of the passes it concerns: 17 scalar instructions per object (loop control +
                // three test-and-skip branches) on the CU's single scalar pipe, the resource this kernel is bound by.  Per (object, pass)
                // visit the walk now costs 6 (32-bit class masks where a class has at most 32 members), nothing is tested and skipped, and
                // one pass's lane constants and ONE running minimum are all that is live in the object loop.
#pragma unroll
                for (int pass_q = 0; pass_q < (TNp > 0 ? N_PASS_T : n_pass); ++pass_q) {
                    int i_first, i_last;  // pursuers of this pass
                    bool okq;
                    int iq, kq;
                    if constexpr (ALIGNED) {
                        const int li = lane / K;
                        i_first = pass_q * PPP; i_last = min(i_first + PPP, Np) - 1;
                        okq = li < PPP && i_first + li <= i_last;
                        iq = okq ? i_first + li : 0;
                        kq = okq ? lane - li * K : 0;
                    } else {
                        const int idx = 64 * pass_q + lane;
                        okq = idx < Np * K;
                        iq = okq ? idx / K : 0;
                        kq = okq ? idx - iq * K : 0;
                        i_first = 64 * pass_q / K; i_last = min(64 * pass_q + 63, Np * K - 1) / K;
                    }
                    const float sxq = SEN[2 * kq], syq = SEN[2 * kq + 1];
                    const float pxq = X[2 * iq], pyq = X[2 * iq + 1];
                    uint64_t u = 0ull;
                    for (int i = i_first; i <= i_last; ++i) u |= NEAR[i];
                    // wave-uniform: objects in reach of any pursuer of this pass
                    const uint64_t reach = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u)) |
                                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32);
                    // (a specialised shape fixes the row width, and with it whether the speed features are in the row: 7 K + 2 (+ 1) against 4 K + 2 (+ 1))
                    const bool speed = TNp > 0 ? (TD >= 7 * TK + 2) : (bool)DA.speed_features;
                    // lanes without a (pursuer, sensor) pair -- 4 of 64 in a pass of two pursuers, 34 in the last pass of C3 -- write their features
                    // to a spare row behind the staging rows instead of branching around the stores (an exec-mask round trip per (pass, class))
                    float *const orow_l = okq ? O + iq * D : O_SPARE;
#pragma unroll
                    for (int cls = 0; cls < 4; ++cls) {
                        const int lo = cls == 0 ? NP : (cls == 1 ? Np : (cls == 2 ? Np + Ne : 0));
                        const int cnt = cls == 0 ? 1 : (cls == 1 ? Ne : (cls == 2 ? Npo : Np));
                        float b = INFINITY;
                        int bi = 0;
                        auto visit = [&](int m, float qx, float qy) {
                            const float rx = qx - pxq, ry = qy - pyq;
                            const float sv = sxq * rx + syq * ry;
                            const float d2 = rx * rx + ry * ry;
                            // branch-free (bitwise |, selects): no exec-mask round trips in the inner loop
                            // sv < 0 || sv > srange as ONE compare: the median of (sv, 0, srange) is sv exactly when 0 <= sv <= srange (sv is finite;
                            // -0.0 compares equal to the +0.0 the median may return, as it passes `sv < 0`)
                            const bool out = (__builtin_amdgcn_fmed3f(sv, 0.f, srange) != sv) | (d2 - sv * sv > rad2) | ((cls == 3) & (m == iq));
                            // (the reference sets an excluded ray to +inf and takes the first minimum: an excluded ray is never "better", a kept one
                            // is when it is smaller -- the same minimum and the same first index without materialising the +inf)
                            const bool better = !out & (sv < b);
                            b = better ? sv : b;
                            bi = better ? m : bi;
                        };
                        if (cls == 0) {
                            if ((reach >> NP) & 1ull) visit(0, ox, oy);
                        } else if (TNp > 0 && cnt <= 32) {
                            uint32_t todo = (uint32_t)(reach >> lo) & (cnt >= 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u));
#pragma nounroll
                            while (todo != 0u) {
                                const int m = __builtin_ctz(todo);
                                todo &= todo - 1u;
                                // the object's position: ONE uniform-address LDS read (a broadcast) instead of two v_readlane + their wait states --
                                // the LDS pipe has room, the VALU port is what this kernel is bound by since the scalar work went
                                const float2 qp = *reinterpret_cast<const float2 *>(&X[2 * (lo + m)]);
                                visit(m, qp.x, qp.y);
                            }
                        } else {
                            uint64_t todo = reach & ((((cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull))) << lo);
#pragma nounroll
                            while (todo != 0ull) {
                                const int bit = __builtin_ctzll(todo);
                                todo &= todo - 1ull;
                                visit(bit - lo, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part_x), bit)),
                                      __int_as_float(__builtin_amdgcn_readlane(__float_as_int(part_y), bit)));
                            }
                        }
                        // the features of (pass, class) go to the staging row now: nothing but ONE running minimum is held in registers
                        {
                            float *o = orow_l;
                            const bool fin = b < INFINITY;
                            const float fd = fin ? b : 0.f;  // W4: raw distance or 0
                            if (cls == 0) {
                                o[kq] = fd;
                            } else {
                                const int j = lo + bi;   // (bi = 0 without a hit: a valid particle, its value is not used)
                                const float raw = sxq * (V[2 * j] - V[2 * iq]) + syq * (V[2 * j + 1] - V[2 * iq + 1]);   // loaded and computed
                                const float fs = fin ? raw : 0.f;  // W5                                                  unconditionally: a select, no branch
                                if (speed) { o[(2 * cls - 1) * K + kq] = fd; o[2 * cls * K + kq] = fs; }
                                else o[cls * K + kq] = fd;
                            }
                        }
                    }
                }
                // pursuer lanes: collision flags, id, who-caught tests for the local rewards
                bool wc = false, wp = false, we = false;
                if (LANE_LT(Np)) {
                    bool tev = false, tpo = false;
                    if constexpr (BITROWS) {
                        uint64_t we_ = col_ev[0], wp_ = col_po[0];  // the word that holds this pursuer's row
#pragma unroll
                        for (int w = 1; w < WE; ++w) we_ = (lane / GE == w) ? col_ev[w] : we_;
#pragma unroll
                        for (int w = 1; w < WP; ++w) wp_ = (lane / GP == w) ? col_po[w] : wp_;
                        const uint64_t row_ev = (we_ >> ((lane % GE) * Ne)) & ((1ull << Ne) - 1ull);
                        const uint64_t row_po = (wp_ >> ((lane % GP) * Npo)) & ((1ull << Npo) - 1ull);
                        tev = row_ev != 0ull;
                        tpo = row_po != 0ull;
                        wc = (row_ev & (caught_mask >> Np)) != 0ull;           // touches a caught evader
                        we = (row_ev & (enc_mask >> Np)) != 0ull;              // touches an encountered evader
                        wp = (row_po & (caught_mask >> (Np + Ne))) != 0ull;    // touches a caught poison
                    } else {
                    for (int e = 0; e < Ne; ++e) {
                        const bool c = COL[lane * Ne + e];
                        tev |= c;
                        wc |= c && FLG[e];
                        we |= c && FLG[Ne + e];
                    }
                    for (int p = 0; p < Npo; ++p) {
                        const bool c = COLP[lane * Npo + p];
                        tpo |= c;
                        wp |= c && FLG[2 * Ne + p];
                    }
                    }
                    float *o = O + lane * D + DA.nfeat * K;  // :411-428
                    o[0] = tev ? 1.f : 0.f;
                    o[1] = tpo ? 1.f : 0.f;
                    if (DA.addid) o[2] = (float)(lane + 1);  // W10
                }
                wave_sync();
                // phase E: respawn caught evaders / poisons (:355-374)
                if ((!LANE_LT(Np) && LANE_LT(NP)) && my_caught) {
                    const bool is_ev = LANE_LT(Np + Ne);
                    float x, y, u0, u1;
                    if (MODE == 1 && IOA.inj_resp != nullptr && !do_init) {
                        const float *r = IOA.inj_resp + (env * NP + lane) * 4;
                        x = r[0]; y = r[1]; u0 = r[2]; u1 = r[3];
                    } else {
                        const float thr = (is_ev ? DA.r_ev : DA.r_po) * 2.0f + DA.obst_r;
                        x = y = u0 = u1 = 0.f;
                        for (uint32_t att = 0; att < 1024u; ++att) {
                            const u32x4 r = philox4x32_10(gid, tick, (uint32_t)lane, WW_TAG_RESPAWN | (att << 8), DA.k0, DA.k1);
                            x = u24(r.x);
                            y = u24(r.y);
                            if (att == 0) { u0 = u24(r.z); u1 = u24(r.w); }
                            if (!(dist2d(x, y, ox, oy) <= thr)) break;
                        }
                    }
                    const float sp = is_ev ? DA.ev_speed : DA.poison_speed;  // W9
                    X[2 * lane] = x; X[2 * lane + 1] = y;
                    V[2 * lane] = (u0 - 0.5f) * sp;
                    V[2 * lane + 1] = (u1 - 0.5f) * sp;
                }
                tick += 1;
                // phase F: rewards (:376-385)
                if (LANE_LT(Np)) {
                    if (DA.reward_global) {
                        reward += ((float)n_evc * DA.food_reward) + ((float)n_poc * DA.poison_reward) +
                                  ((float)n_enc * DA.encounter_reward);
                    } else {  // fancy-index += pays a pursuer once per kind (W7)
                        if (wc) reward += DA.food_reward;
                        if (wp) reward += DA.poison_reward;
                        if (we) reward += DA.encounter_reward;
                    }
                }
                wave_sync();
                // phase G: evaders / poisons move; velocity flips only if BOTH coordinates left [0,1] (W6)
                if ((!LANE_LT(Np) && LANE_LT(NP))) {
                    float x = X[2 * lane], y = X[2 * lane + 1], vx = V[2 * lane], vy = V[2 * lane + 1];
                    x = x + vx; y = y + vy;
                    const bool outx = !(x >= 0.f && x <= 1.f), outy = !(y >= 0.f && y <= 1.f);
                    if (outx && outy) { vx = -1.0f * vx; vy = -1.0f * vy; }
                    X[2 * lane] = x; X[2 * lane + 1] = y; V[2 * lane] = vx; V[2 * lane + 1] = vy;
                }
                tstep += 1;  // :433
                const int limit = DA.max_steps > 0 ? DA.max_steps : 1000;  // timestep_limit :124-126
                const bool is_done = tstep >= limit;                     // :174-178
                wave_sync();

                if (pass == 0) asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt_act));  // pipeline hinge
                // ---------------------------------------------------- outputs
                if (MODE == 1 && !do_init) {
                    if (LANE_LT(Np)) uniform_ptr(IOA.rew + env * Np)[ulane] = reward;
                    if (FUSED && IOA.st->rew_out != nullptr && LANE_LT(Np)) {  // StandardizedEnv.step :283-291
                        const WwStd &st = *IOA.st;
                        const int64_t i = env * Np + lane;
                        double r = (double)reward;
                        if (st.enable_rewnorm) {
                            const double m = (1.0 - st.rew_alpha) * st.rew_mean[i] + st.rew_alpha * r;      // :253-254
                            const double dd = r - m;
                            const double v = (1.0 - st.rew_alpha) * st.rew_var[i] + st.rew_alpha * (dd * dd);  // :255-257
                            st.rew_mean[i] = m;
                            st.rew_var[i] = v;
                            r = r / (sqrt(v) + st.eps);                                                   // :268-271
                        }
                        st.rew_out[i] = (float)(st.scale * r);                                           // :290
                    }
                    if (LANE_EQ(0)) {
                        IOA.done[env] = (uint8_t)is_done;
                        IOA.info[2 * env] = n_evc;
                        IOA.info[2 * env + 1] = n_poc;
                    }
                    if (is_done && DA.auto_reset) {  // wave-uniform: run the reset pass next
                        npass = 2;
                        do_init = true;
                    }
                }
                if (pass == npass - 1) {
                    float *const obs_p = IOA.obs;
                    const auto orow = uniform_ptr(obs_p + env * (int64_t)(Np * D));
#if MADRL_WW_ABLATE & 2
                    if (DA.n_envs < 0)
#endif
                    if (obs_p != nullptr) {  // the raw row may be dropped when the fused wrapper output is all the caller reads
                        // 16 bytes per lane (ds_read_b128 + one 16-byte store; an env's rows start on a 4-byte boundary only -- gfx950 under HSA runs
                        // global accesses in unaligned mode -- and the rows of neighbouring envs are contiguous, so whole lines leave the chip anyway)
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        typedef float f4a __attribute__((ext_vector_type(4), aligned(16)));
                        const uint32_t n4 = (uint32_t)(Np * D) / 4u;
                        for (uint32_t e = ulane; e < n4; e += 64u)
                            *reinterpret_cast<__attribute__((address_space(1))) f4u *>(orow + 4 * e) = *reinterpret_cast<const f4a *>(O + 4 * e);
                        for (uint32_t e = 4u * n4 + ulane; e < (uint32_t)(Np * D); e += 64u) orow[e] = O[e];
                    }
                    if (FUSED) {  // StandardizedEnv.standardize_obs :242-263
                        const WwStd &st = *IOA.st;
                        const int64_t base = env * (int64_t)(Np * D);
                        if (st.enable_obsnorm) {
                            // batches of 4 elements per lane: all 8 statistics loads of a batch are in flight before the first
                            // dependent float64 operation (element by element the loop pays one HBM round trip each: 421 instead
                            // of 357 us per wrapped step; 16-byte pair accesses on top measured no further gain)
                            const int n_el = Np * D;
                            const double *__restrict__ gm = st.obs_mean + base;
                            const double *__restrict__ gv = st.obs_var + base;
                            for (int e0 = lane; e0 < n_el; e0 += 256) {
                                double m[4], v[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int e = e0 + 64 * u;
                                    m[u] = e < n_el ? __builtin_nontemporal_load(&gm[e]) : 0.0;
                                    v[u] = e < n_el ? __builtin_nontemporal_load(&gv[e]) : 1.0;
                                }
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const int e = e0 + 64 * u;
                                    if (e < n_el) {
                                        const double x = (double)O[e];
                                        const double mm = (1.0 - st.obs_alpha) * m[u] + st.obs_alpha * x;      // :245-246
                                        const double dd = x - mm;
                                        const double vv = (1.0 - st.obs_alpha) * v[u] + st.obs_alpha * (dd * dd);  // :247-249
                                        __builtin_nontemporal_store(mm, &st.obs_mean[base + e]);
                                        __builtin_nontemporal_store(vv, &st.obs_var[base + e]);
                                        __builtin_nontemporal_store((float)((x - mm) / (sqrt(vv) + st.eps)), &st.obs_out[base + e]);  // :262-263
                                    }
                                }
                            }
                        } else {
                            for (int e = lane; e < Np * D; e += 64) st.obs_out[base + e] = O[e];
                        }
                    }
                }
                wave_sync();
            }
            // ---------------------------------------------------------- LDS -> record
            if (fresh(lane) == 0) {
                reinterpret_cast<int32_t *>(S)[4 * NP + 2] = tstep;
                reinterpret_cast<uint32_t *>(S)[4 * NP + 3] = tick;
            }
            wave_sync();
            {
                const auto dst = uniform_ptr(reinterpret_cast<uint32_t *>(DA.state) + env * (int64_t)rec_dw);
                for (uint32_t k = ulane; k < (uint32_t)rec_dw; k += 64u) dst[k] = reinterpret_cast<const uint32_t *>(S)[k];
            }
            wave_sync();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        cur_act = nxt_act;
    }
}
#undef DA
#undef IOA
#undef LANE_LT
#undef LANE_EQ

}  // namespace
// =================================================================== host side / C ABI
struct madrl_waterworld {
    madrl_waterworld_config cfg;
    WwDev dev;
    int device;
    int64_t max_blocks;
    size_t lds_bytes;
    void *tables;
    WwStd *std_dev;   // device copy of the bound StandardizedEnv arguments (madrl_waterworld_set_standardize)
    bool std_bound;
};

namespace {

int ww_validate(const madrl_waterworld_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_waterworld_config))
        return fail(MADRL_EINVAL, "madrl_waterworld_config.struct_size=%d, library expects %d", c->struct_size,
                    (int)sizeof(madrl_waterworld_config));
    if (c->n_pursuers < 1 || c->n_evaders < 1 || c->n_poison < 1)
        return fail(MADRL_EINVAL, "n_pursuers, n_evaders, n_poison must be >= 1");
    if (c->n_pursuers + c->n_evaders + c->n_poison > 62)
        return fail(MADRL_EINVAL, "at most 62 particles per env (one wavefront per env)");
    if (2 * c->n_pursuers > 64) return fail(MADRL_EINVAL, "n_pursuers must be <= 32");
    if (c->n_sensors < 1 || c->n_sensors > 256) return fail(MADRL_EINVAL, "n_sensors must be in 1..256");
    if (c->n_coop < 1) return fail(MADRL_EINVAL, "n_coop must be >= 1");
    return MADRL_OK;
}

int ww_obs_dim_of(const madrl_waterworld_config *c) {
    return c->n_sensors * (c->speed_features ? 7 : 4) + 2 + (c->addid ? 1 : 0);  // Archea.__init__ :18-24
}

// largest float32 x with sqrtf(x) <= thr (thr >= 0 finite); see WwDev::sq_*
float sq_threshold(float thr) {
    float x = thr * thr;
    while (x > 0.0f && sqrtf(x) > thr) x = nextafterf(x, 0.0f);
    for (;;) {
        const float up = nextafterf(x, INFINITY);
        if (!(sqrtf(up) <= thr)) break;
        x = up;
    }
    return x;
}

void ww_layout(const madrl_waterworld_config *c, WwDev *d) {
    memset(d, 0, sizeof(*d));
    d->Np = c->n_pursuers; d->Ne = c->n_evaders; d->Npo = c->n_poison; d->NP = d->Np + d->Ne + d->Npo;
    d->K = c->n_sensors; d->D = ww_obs_dim_of(c); d->nfeat = c->speed_features ? 7 : 4;
    d->n_coop = c->n_coop; d->addid = c->addid; d->speed_features = c->speed_features;
    d->reward_global = c->reward_global; d->obstacle_fixed = c->obstacle_fixed; d->max_steps = c->max_steps;
    d->auto_reset = c->auto_reset;
    d->rec_dw = (int)align_up((size_t)4 * d->NP + 4, 4);
    d->k0 = (uint32_t)c->seed; d->k1 = (uint32_t)(c->seed >> 32); d->gid_base = (uint32_t)c->env_id_base;
    // radii: pursuer r, evader 2r, poison 3r/4, evaluated in float64 like the reference (:108-118)
    d->r_pu = (float)c->radius; d->r_ev = (float)(c->radius * 2); d->r_po = (float)(c->radius * 3 / 4);
    d->obst_r = (float)c->obstacle_radius; d->ev_speed = (float)c->ev_speed; d->poison_speed = (float)c->poison_speed;
    d->sensor_range = (float)c->sensor_range; d->action_scale = (float)c->action_scale;
    d->poison_reward = (float)c->poison_reward; d->food_reward = (float)c->food_reward;
    d->encounter_reward = (float)c->encounter_reward; d->control_penalty = (float)c->control_penalty;
    d->obst_x = (float)c->obstacle_loc[0]; d->obst_y = (float)c->obstacle_loc[1];
    // the float32 sums are the ones the kernel used to form before comparing (pr + obst_r, r_pu + r_ev, r_pu + r_po)
    d->sq_obst_pu = sq_threshold(d->r_pu + d->obst_r); d->sq_obst_ev = sq_threshold(d->r_ev + d->obst_r);
    d->sq_obst_po = sq_threshold(d->r_po + d->obst_r);
    d->sq_hit_ev = sq_threshold(d->r_pu + d->r_ev); d->sq_hit_po = sq_threshold(d->r_pu + d->r_po);
}

size_t ww_lds_bytes(const WwDev &d) {
    size_t f = align_up((size_t)d.rec_dw, 4) + align_up((size_t)(d.Np + 1) * d.D, 4) + align_up((size_t)2 * d.K, 4);   // (Np + 1: the spare row of the sensing phase)
    size_t b = f * 4 + 8 * (size_t)d.Np + (size_t)d.Np * (d.Ne + d.Npo) + 2 * (size_t)d.Ne + d.Npo;
    return align_up(b, 16);
}

struct WwSpec {
    int Np, Ne, Npo, K, D;
    void (*launch)(const WwDev &, const WwIO &, int mode, bool fused, dim3 g, hipStream_t s);
};

template <int TNp, int TNe, int TNpo, int TK, int TD>
void ww_launch_spec(const WwDev &d, const WwIO &io, int mode, bool fused, dim3 g, hipStream_t s) {
    const dim3 b(64);  // static LDS layout: 0 dynamic bytes
    if (mode == 0) {
        if (fused) hipLaunchKernelGGL((waterworld_kernel<0, TNp, TNe, TNpo, TK, true, TD>), g, b, 0, s, d, io);
        else hipLaunchKernelGGL((waterworld_kernel<0, TNp, TNe, TNpo, TK, false, TD>), g, b, 0, s, d, io);
    } else {
        if (fused) hipLaunchKernelGGL((waterworld_kernel<1, TNp, TNe, TNpo, TK, true, TD>), g, b, 0, s, d, io);
        else hipLaunchKernelGGL((waterworld_kernel<1, TNp, TNe, TNpo, TK, false, TD>), g, b, 0, s, d, io);
    }
}

#define X(NP_, NE_, NPO_, K_, D_) {NP_, NE_, NPO_, K_, D_, ww_launch_spec<NP_, NE_, NPO_, K_, D_>},
const WwSpec WW_SPECS[] = {
#include "waterworld_specializations.def"
#if __has_include("waterworld_specializations.local.def")   // shapes added on this machine by `python -m madrl_amd.build --waterworld-shape ...` (git-ignored)
#include "waterworld_specializations.local.def"
#endif
};
#undef X

int ww_launch(const madrl_waterworld *h, const WwIO &io, int mode, void *stream) {
    int64_t blocks = h->max_blocks > 0 ? h->max_blocks : 256 * 64;
    if (blocks > h->dev.n_envs) blocks = h->dev.n_envs;
    hipStream_t s = (hipStream_t)stream;
    const WwDev &d = h->dev;
    const dim3 g((unsigned)blocks), b(64);
    const bool fused = io.st != nullptr;
    const WwSpec *spec = nullptr;
    for (const WwSpec &w : WW_SPECS)
        if (d.Np == w.Np && d.Ne == w.Ne && d.Npo == w.Npo && d.K == w.K && d.D == w.D) spec = &w;
    if (spec) {
        spec->launch(h->dev, io, mode, fused, g, s);
    } else if (mode == 0) {
        if (fused) hipLaunchKernelGGL((waterworld_kernel<0, 0, 0, 0, 0, true>), g, b, h->lds_bytes, s, h->dev, io);
        else hipLaunchKernelGGL((waterworld_kernel<0, 0, 0, 0, 0, false>), g, b, h->lds_bytes, s, h->dev, io);
    } else {
        if (fused) hipLaunchKernelGGL((waterworld_kernel<1, 0, 0, 0, 0, true>), g, b, h->lds_bytes, s, h->dev, io);
        else hipLaunchKernelGGL((waterworld_kernel<1, 0, 0, 0, 0, false>), g, b, h->lds_bytes, s, h->dev, io);
    }
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

__global__ void ww_state_copy_kernel(const WwDev d, float *pos, float *vel, float *obst, int32_t *t, uint32_t *tick,
                                     const int to_state) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    float *rec = d.state + env * (int64_t)d.rec_dw;
    const int NP = d.NP;
    for (int k = 0; k < 2 * NP; ++k) {
        if (pos) { if (to_state) rec[k] = pos[env * 2 * NP + k]; else pos[env * 2 * NP + k] = rec[k]; }
        if (vel) { if (to_state) rec[2 * NP + k] = vel[env * 2 * NP + k]; else vel[env * 2 * NP + k] = rec[2 * NP + k]; }
    }
    for (int k = 0; k < 2; ++k)
        if (obst) { if (to_state) rec[4 * NP + k] = obst[env * 2 + k]; else obst[env * 2 + k] = rec[4 * NP + k]; }
    int32_t *ti = reinterpret_cast<int32_t *>(rec) + 4 * NP + 2;
    uint32_t *tk = reinterpret_cast<uint32_t *>(rec) + 4 * NP + 3;
    if (t) { if (to_state) *ti = t[env]; else t[env] = *ti; }
    if (tick) { if (to_state) *tk = tick[env]; else tick[env] = *tk; }
}

}  // namespace

extern "C" {

int madrl_waterworld_obs_dim(const madrl_waterworld_config *cfg, int32_t *out_dim) {
    int rc = ww_validate(cfg);
    if (rc) return rc;
    if (!out_dim) return fail(MADRL_EINVAL, "out_dim is NULL");
    *out_dim = ww_obs_dim_of(cfg);
    return MADRL_OK;
}

int madrl_waterworld_state_bytes(const madrl_waterworld_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    int rc = ww_validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_bytes) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_bytes non-NULL");
    WwDev d;
    ww_layout(cfg, &d);
    *out_bytes = (uint64_t)d.rec_dw * 4u * (uint64_t)n_envs;
    return MADRL_OK;
}

int madrl_waterworld_create(const madrl_waterworld_config *cfg, const double *sensors_host, int64_t n_envs,
                            int32_t device, void *state_dev, madrl_waterworld **out) {
    int rc = ww_validate(cfg);
    if (rc) return rc;
    if (!sensors_host || !state_dev || !out || n_envs < 1) return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    if (n_envs >= 0x7FF00000ll)  // the kernel indexes envs with 32-bit integers (index + workgroup count must stay below 2^31)
        return fail(MADRL_EINVAL, "n_envs=%lld is too large for one handle (limit 2146435071); shard the batch", (long long)n_envs);
    if (n_envs + cfg->env_id_base > 0xFFFFFFFFll) return fail(MADRL_EINVAL, "global env index must fit 32 bits");
    MADRL_HIP_TRY(hipSetDevice(device));
    madrl_waterworld *h = new (std::nothrow) madrl_waterworld();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    ww_layout(cfg, &h->dev);
    h->dev.n_envs = n_envs;
    h->dev.state = (float *)state_dev;
    h->lds_bytes = ww_lds_bytes(h->dev);
    h->max_blocks = 0;
    if (h->lds_bytes > 64 * 1024) {
        delete h;
        return fail(MADRL_EINVAL, "configuration needs %zu B of LDS (> 64 KiB)", h->lds_bytes);
    }
    std::vector<float> sens(2 * (size_t)cfg->n_sensors);
    for (size_t k = 0; k < sens.size(); ++k) sens[k] = (float)sensors_host[k];  // float64 cos/sin rounded once
    hipError_t e = hipMalloc(&h->tables, sens.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(h->tables, sens.data(), sens.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (h->tables) (void)hipFree(h->tables);
        delete h;
        return fail(MADRL_EHIP, "sensor table upload failed: %s", hipGetErrorString(e));
    }
    h->dev.sensors = (const float *)h->tables;
    *out = h;
    return MADRL_OK;
}

void madrl_waterworld_destroy(madrl_waterworld *h) {
    if (!h) return;
    if (h->tables) (void)hipFree(h->tables);
    if (h->std_dev) (void)hipFree(h->std_dev);
    delete h;
}

int madrl_waterworld_set_standardize(madrl_waterworld *h, const madrl_standardize_args *a) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (!a) { h->std_bound = false; return MADRL_OK; }
    if (a->struct_size != (int32_t)sizeof(madrl_standardize_args))
        return fail(MADRL_EINVAL, "madrl_standardize_args.struct_size=%d, library expects %d", a->struct_size, (int)sizeof(madrl_standardize_args));
    if (!a->obs_out || (a->enable_obsnorm && (!a->obs_mean || !a->obs_var)) || (a->rew_out && a->enable_rewnorm && (!a->rew_mean || !a->rew_var)))
        return fail(MADRL_EINVAL, "set_standardize: obs_out and the running statistics of every enabled normalisation are required");
    WwStd st;
    st.obs_mean = a->obs_mean; st.obs_var = a->obs_var; st.obs_out = a->obs_out;
    st.rew_mean = a->rew_mean; st.rew_var = a->rew_var; st.rew_out = a->rew_out;
    st.obs_alpha = a->obs_alpha; st.rew_alpha = a->rew_alpha; st.eps = a->eps; st.scale = a->scale_reward;
    st.enable_obsnorm = a->enable_obsnorm; st.enable_rewnorm = a->enable_rewnorm;
    MADRL_HIP_TRY(hipSetDevice(h->device));
    if (!h->std_dev) MADRL_HIP_TRY(hipMalloc((void **)&h->std_dev, sizeof(WwStd)));
    MADRL_HIP_TRY(hipMemcpy(h->std_dev, &st, sizeof(WwStd), hipMemcpyHostToDevice));
    h->std_bound = true;
    return MADRL_OK;
}

int madrl_waterworld_set_launch(madrl_waterworld *h, int64_t max_blocks) {
    if (!h || max_blocks < 0) return fail(MADRL_EINVAL, "set_launch: bad argument");
    h->max_blocks = max_blocks;
    return MADRL_OK;
}

int madrl_waterworld_reset(madrl_waterworld *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || (!obs_dev && !h->std_bound)) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    WwIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.obs = obs_dev;
    io.st = h->std_bound ? h->std_dev : nullptr;
    return ww_launch(h, io, 0, stream);
}

int madrl_waterworld_step(madrl_waterworld *h, const float *actions_dev, const float *inj_respawn_dev, float *obs_dev,
                          float *rew_dev, uint8_t *done_dev, int32_t *info_dev, void *stream) {
    if (!h || !actions_dev || (!obs_dev && !h->std_bound) || !rew_dev || !done_dev || !info_dev) return fail(MADRL_EINVAL, "step: NULL argument");
    WwIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions_dev;
    io.inj_resp = inj_respawn_dev;
    io.obs = obs_dev;
    io.st = h->std_bound ? h->std_dev : nullptr;
    io.rew = rew_dev;
    io.done = done_dev;
    io.info = info_dev;
    return ww_launch(h, io, 1, stream);
}

int madrl_waterworld_get_state(madrl_waterworld *h, float *pos, float *vel, float *obst, int32_t *t, uint32_t *tick,
                               void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(ww_state_copy_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, pos, vel, obst, t,
                       tick, 0);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_waterworld_set_state(madrl_waterworld *h, const float *pos, const float *vel, const float *obst,
                               const int32_t *t, const uint32_t *tick, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(ww_state_copy_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, (float *)pos,
                       (float *)vel, (float *)obst, (int32_t *)t, (uint32_t *)tick, 1);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // extern "C"
