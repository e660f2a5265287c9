// hostage.hip -- batched ContinuousHostageWorld for MI355X (gfx950 / CDNA4), float32.
//
// Same execution model as waterworld.hip: one wavefront owns one env at a time (64-thread persistent workgroups striding
// over envs); the particles (rescuers | hostages | criminals: position + velocity), key, bomb, flags and the assembled
// observation rows live in LDS; HBM sees one packed state record in / out, the action row in and observation / reward /
// done / info rows out.  Lane roles per phase: particle lanes (integration, walls, gate, respawn, motion), (rescuer,
// object) collision pairs, (rescuer, sensor) sensing pairs -- the objects a sensor is tested against are broadcast once
// from the owning lane's registers (v_readlane -> SGPR operands) and reused by three passes of pairs held in registers.
//
// Reference semantics (file:line under /root/reference/madrl_environments/hostage.py):
//   sensing ...... CircAgent.sensed :62-71     reset ...... ContinuousHostageWorld.reset :137-177 (ends with a zero-action step)
//   catch rule ... _caught :184-198             step ....... :228-430
// Quirks kept (G1..G9) are listed where they occur.  Arithmetic is float32, every expression keeps the statement order of the
// reference's step() so that a float32 CPU restatement agrees bit for bit.
#include "common.hpp"

#include <math.h>
#include <new>
#include <string.h>
#include <vector>

// Profiling aid (scripts/variants.sh builds one library per value, never the shipped one):
//   1 no observation store   2 non-temporal observation store   4 no record store   8 no reward / done / info stores
#ifndef MADRL_HW_ABLATE
#define MADRL_HW_ABLATE 0
#endif
#ifndef MADRL_HW_WAVES
#define MADRL_HW_WAVES 7   // resident wavefronts per SIMD the SPECIALISED kernel's register allocation aims at (generic: compiler's choice).  Round 1 kernel: 117 VGPRs, 4 waves (at 5 it spilled
                           // 60 B per lane and the spill stores reached HBM).  With the launch parameters out of the SGPR file
                           // and the static LDS layout: 72 VGPRs, no scratch at 7 waves.  32 768 envs: 4 waves 55.2 us, 5 50.7, 6 48.0, 7 46.9.
#endif

namespace {

using namespace madrl;

enum : uint32_t { HW_TAG_RESPAWN = 48, HW_TAG_RESET = 49 };

struct HwDev {
    int32_t Nr, Nh, Nc, NP, K, D;
    int32_t n_coop_save, addid, reward_global, key_fixed, max_steps, auto_reset;
    int32_t rec_dw;  // dwords per packed state record: pos[NP][2] vel[NP][2] key[2] bomb[2] saved_lo saved_hi flags t tick
    uint32_t k0, k1, gid_base;
    float radius, r_ho, bad_speed, sensor_range, action_scale, gate_lo;
    float save_reward, hit_reward, encounter_reward, not_saved_reward, bomb_reward, bomb_radius, key_radius, control_penalty;
    float key_x, key_y;
    // sq_*: largest float32 x with sqrtf(x) <= threshold ("distance <= threshold" as one compare of the squared distance, same truth
    // value for every input; waterworld.hip): rescuer-hostage / rescuer-criminal contact, bomb and key radii
    float sq_hit_ho, sq_hit_cr, sq_bomb, sq_key;
    int64_t n_envs;
    const float *sensors;  // [K][2]
    float *state;
};

struct HwIO {
    const uint8_t *mask;    // reset mode
    const float *actions;   // [N][Nr][2]
    const float *inj_resp;  // [N][Nc][4] or NULL
    float *obs;             // [N][Nr][D]
    float *rew;             // [N][Nr]
    uint8_t *done;          // [N]
    int32_t *info;          // [N][2]  ho_saved, cr_encs
};

// Same register discipline as waterworld.hip: launch parameters are read from the kernel-argument segment (scalar loads) where a
// phase needs them instead of being held in SGPRs across the env loop (what does not fit in SGPRs is parked in VGPR lanes at two
// VALU issue slots per value and use); per-lane global accesses go through an SGPR base + 32-bit VGPR offset; a lane predicate
// is recomputed at its use (fresh) instead of being hoisted out of the env loop as an SGPR pair.
struct HwKArgs {
    HwDev d;
    HwIO io;
};
typedef const __attribute__((address_space(4))) HwKArgs *HwKArgsPtr;
__device__ __forceinline__ HwKArgsPtr hw_args() {
    HwKArgsPtr p = (HwKArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *uniform_ptr(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (__attribute__((address_space(1))) T *)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
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
__device__ __forceinline__ bool dist2_le(float ax, float ay, float bx, float by, float sq) {  // dist2d(a, b) <= thr with sq = sq_threshold(thr)
    const float dx = ax - bx, dy = ay - by;
    return dx * dx + dy * dy <= sq;
}
__device__ __forceinline__ float clipf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float bcast(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }

// MODE 0: reset(mask)   MODE 1: step (+ fused auto-reset)
// TNr..TK > 0: the particle / sensor counts are compile-time constants (small loops unroll, the index divisions fold); 0: generic.
template <int MODE, int TNr, int TNh, int TNc, int TK, int TD = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TNr > 0 ? MADRL_HW_WAVES : 1, TNr > 0 ? MADRL_HW_WAVES : 8))) void hostage_kernel(const HwDev d, const HwIO io) {
    // specialised shape: compile-time LDS layout in a static array, launched with 0 dynamic bytes (see waterworld.hip)
    constexpr int SPEC_DW = TNr > 0 ? ((4 * (TNr + TNh + TNc) + 9 + 3) / 4 * 4 + ((TNr + 1) * (TD > 0 ? TD : 1) + 3) / 4 * 4 + (2 * TK + 3) / 4 * 4) : 0;   // (TNr + 1: the spare row of the sensing phase)
    constexpr int SPEC_BYTES = TNr > 0 ? (SPEC_DW * 4 + 8 * TNr + TNr * (TNh + TNc) + 2 * TNh + TNc + 15) / 16 * 16 : 16;
    static_assert(TNr == 0 || TD > 0, "a specialised shape fixes the observation width too");
    extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
    __shared__ __attribute__((aligned(16))) float smem_static[SPEC_BYTES / 4];
    float *const smem = TNr > 0 ? smem_static : smem_dyn;
    const int lane = threadIdx.x;
    const uint32_t ulane = threadIdx.x;
#define DA (hw_args()->d)
#define IOA (hw_args()->io)
    const int Nr = TNr > 0 ? TNr : d.Nr, Nh = TNr > 0 ? TNh : d.Nh, Nc = TNr > 0 ? TNc : d.Nc, K = TNr > 0 ? TK : d.K;
    const int NP = Nr + Nh + Nc, D = TD > 0 ? TD : d.D;
    float *S = smem;                                   // packed record
    float *X = S, *V = S + 2 * NP;
    uint32_t *SU = reinterpret_cast<uint32_t *>(S);
    const int OFF_KEY = 4 * NP, OFF_BOMB = 4 * NP + 2, OFF_SAVED = 4 * NP + 4, OFF_FLAGS = 4 * NP + 6, OFF_T = 4 * NP + 7, OFF_TICK = 4 * NP + 8;
    const int rec_dw = TNr > 0 ? (4 * (TNr + TNh + TNc) + 9 + 3) / 4 * 4 : d.rec_dw;
    float *O = S + ((rec_dw + 3) & ~3);                // observation staging [Nr][D]
    float *const O_SPARE = O + Nr * D;                 // one more row: where the sensing lanes without a (rescuer, sensor) pair write
    float *SEN = O + (((Nr + 1) * D + 3) & ~3);        // sensor unit vectors [K][2]
    uint64_t *NEAR = reinterpret_cast<uint64_t *>(SEN + ((2 * K + 3) & ~3));  // per rescuer: particles (bit j), key (bit NP), bomb (bit NP + 1) in sensing reach
    uint8_t *COLH = reinterpret_cast<uint8_t *>(NEAR + Nr);  // [Nr][Nh]
    uint8_t *COLC = COLH + Nr * Nh;                    // [Nr][Nc]
    uint8_t *FLG = COLC + Nr * Nc;                     // ho_caught[Nh] | ho_enc[Nh] | cr_caught[Nc]

    for (int k = lane; k < 2 * K; k += 64) SEN[k] = d.sensors[k];
    const int nreg = (rec_dw + 63) >> 6;  // <= 4

    uint32_t cur[4] = {0, 0, 0, 0};
    float cur_act = 0.0f;
    auto fetch = [&](int64_t env, uint32_t (&r)[4], float &a) {
        const auto src = uniform_ptr(reinterpret_cast<const uint32_t *>(DA.state) + env * (int64_t)rec_dw);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t k = ulane + 64u * q;
            r[q] = (q < nreg && (int)k < rec_dw) ? src[k] : 0u;
        }
        if constexpr (MODE == 1) a = (lane < 2 * Nr) ? uniform_ptr(IOA.actions + env * 2 * Nr)[ulane] : 0.0f;
        else a = 0.0f;
    };
    const EnvWalk walk = env_walk(d.n_envs);  // XCD-aware: neighbouring envs share an L2 (common.hpp)
    if (walk.first < walk.lim) fetch(walk.base + walk.first, cur, cur_act);
    asm volatile("" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur_act));
    wave_sync();

    const int w_base = (int)walk.base, w_stride = (int)walk.stride, w_lim = (int)walk.lim;  // env indices are 32-bit, byte offsets 64-bit
    for (int li = (int)walk.first; li < w_lim; li += w_stride) {
        const int64_t env = w_base + li;
        const int64_t nenv = env + w_stride;
        uint32_t nxt[4] = {0, 0, 0, 0};
        float nxt_act = 0.0f;
        if (li + w_stride < w_lim) fetch(nenv, nxt, nxt_act);
        bool skip = false;
        if constexpr (MODE == 0) skip = (IOA.mask != nullptr && IOA.mask[env] == 0);
        if (!skip) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = lane + 64 * q;
                if (q < nreg && k < rec_dw) SU[k] = cur[q];
            }
            wave_sync();
            int32_t tstep = (int32_t)SU[OFF_T];
            uint32_t tick = SU[OFF_TICK];
            uint32_t flags = SU[OFF_FLAGS];  // bit0 gate_open, bit1 bombed, bit2 key sampled
            uint64_t saved = (uint64_t)SU[OFF_SAVED] | ((uint64_t)SU[OFF_SAVED + 1] << 32);
            const uint32_t gid = DA.gid_base + (uint32_t)env;
            float act_lane = cur_act;

            bool do_init = (MODE == 0);
            int npass = 1;
            for (int pass = 0; pass < npass; ++pass) {
                if (do_init) {
                    // ------------------------------------------------ reset (:137-177); draw index: key 0, particle j -> 1 + j, bomb 1 + NP
                    tstep = 0;
                    if (fresh(lane) < NP + 2) {
                        const uint32_t di = fresh(lane) < NP ? 1u + (uint32_t)lane : (fresh(lane) == NP ? 0u : 1u + (uint32_t)NP);
                        const u32x4 r = philox4x32_10(gid, tick, di, HW_TAG_RESET, DA.k0, DA.k1);
                        const float u0 = u24(r.x), u1 = u24(r.y), u2 = u24(r.z), u3 = u24(r.w);
                        if (fresh(lane) < Nr) {  // :149-153
                            X[2 * lane] = u0; X[2 * lane + 1] = clipf(u1, 0.55f, 0.95f);
                            V[2 * lane] = 0.f; V[2 * lane + 1] = 0.f;
                        } else if (fresh(lane) < Nr + Nh) {  // :156-160
                            X[2 * lane] = u0; X[2 * lane + 1] = clipf(u1, 0.f, 0.35f + u2 * 0.01f);
                            V[2 * lane] = 0.f; V[2 * lane + 1] = 0.f;
                        } else if (fresh(lane) < NP) {  // :165-168 (velocity not centred here)
                            X[2 * lane] = u0; X[2 * lane + 1] = u1;
                            V[2 * lane] = u2 * DA.bad_speed; V[2 * lane + 1] = u3 * DA.bad_speed;
                        } else if (fresh(lane) == NP) {  // key: the first reset of the env's life only (G2, :143-146)
                            if (!(flags & 4u)) {
                                S[OFF_KEY] = DA.key_fixed ? DA.key_x : 1.f - u0 * 0.1f;
                                S[OFF_KEY + 1] = DA.key_fixed ? DA.key_y : 1.f - u1 * 0.1f;
                            }
                        } else {  // bomb :171
                            S[OFF_BOMB] = clipf(u0, 0.f, 0.25f); S[OFF_BOMB + 1] = clipf(u1, 0.f, 0.25f);
                        }
                    }
                    saved = 0ull;
                    flags = 4u;
                    tick += 1;
                    act_lane = 0.0f;  // reset ends with step(zeros) (:173)
                    wave_sync();
                }
                // ---------------------------------------------------- step (:228-430)
                const float kx = S[OFF_KEY], ky = S[OFF_KEY + 1], bx = S[OFF_BOMB], by = S[OFF_BOMB + 1];
                const bool gate0 = flags & 1u;  // gate state before this step's key processing (G5)
                float reward = 0.0f;
                bool col_bo = false, col_ke = false;
                {   // phase A: rescuers (:231-260)
                    const float a_raw0 = __shfl(act_lane, 2 * (fresh(lane) < Nr ? lane : 0));
                    const float a_raw1 = __shfl(act_lane, 2 * (fresh(lane) < Nr ? lane : 0) + 1);
                    const float a0 = a_raw0 * DA.action_scale, a1 = a_raw1 * DA.action_scale;
                    float pen = DA.control_penalty * (a0 * a0 + a1 * a1);
                    if (DA.reward_global) {  // (actions**2).sum(), row-major (:241-242)
                        float s = 0.0f;
                        for (int i = 0; i < Nr; ++i) {
                            const float b0 = __shfl(a0, i), b1 = __shfl(a1, i);
                            s += b0 * b0;
                            s += b1 * b1;
                        }
                        pen = DA.control_penalty * s;
                    }
                    if (fresh(lane) < Nr) {
                        float x = X[2 * lane], y = X[2 * lane + 1], vx = V[2 * lane], vy = V[2 * lane + 1];
                        vx = vx + a0; vy = vy + a1;
                        x = x + vx; y = y + vy;
                        reward = 0.0f + pen;
                        float cx = clipf(x, 0.f, 1.f), cy = clipf(y, 0.f, 1.f);  // walls :247-252
                        if (x != cx) vx = 0.f;
                        if (y != cy) vy = 0.f;
                        x = cx; y = cy;
                        if (!gate0) {  // G3: both coordinates, velocity component flipped (:255-260)
                            cx = clipf(x, DA.gate_lo, 1.f); cy = clipf(y, DA.gate_lo, 1.f);
                            if (x != cx) vx *= -1.f;
                            if (y != cy) vy *= -1.f;
                            x = cx; y = cy;
                        }
                        X[2 * lane] = x; X[2 * lane + 1] = y; V[2 * lane] = vx; V[2 * lane + 1] = vy;
                        col_bo = dist2_le(x, y, bx, by, DA.sq_bomb);  // dist <= radius + bomb_radius, :281-291
                        col_ke = dist2_le(x, y, kx, ky, DA.sq_key);   // dist <= radius + key_radius
                    }
                }
                wave_sync();
                // phase B: collisions (:263-279), no saved mask here (G4)
                // BITROWS (specialised shapes with at most 64 rescuer x hostage and rescuer x criminal pairs): a collision matrix is one
                // wave-uniform 64-bit ballot (bit i * n + m), columns counted and rows tested with bit operations (waterworld.hip)
                constexpr bool BITROWS = TNr > 0 && TNr * TNh <= 64 && TNr * TNc <= 64 && TNh < 64 && TNc < 64;
                uint64_t col_ho = 0ull, col_cr = 0ull;
                bool my_caught = false, my_enc = false;  // hostage / criminal lanes count their column (_caught :184-198)
                if constexpr (BITROWS) {
                    {
                        const bool in = fresh(lane) < Nr * Nh;
                        const int i = in ? lane / Nh : 0, m = in ? lane - i * Nh : 0, j = Nr + m;
                        col_ho = __ballot(in && dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], DA.sq_hit_ho));
                    }
                    {
                        const bool in = fresh(lane) < Nr * Nc;
                        const int i = in ? lane / Nc : 0, m = in ? lane - i * Nc : 0, j = Nr + Nh + m;
                        col_cr = __ballot(in && dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], DA.sq_hit_cr));
                    }
                    uint64_t cm_ho = 0ull, cm_cr = 0ull;  // bit i * n of every row
#pragma unroll
                    for (int i = 0; i < (TNr > 0 ? TNr : 1); ++i) { cm_ho |= 1ull << (i * Nh); cm_cr |= 1ull << (i * Nc); }
                    if (fresh(lane) >= Nr && fresh(lane) < NP) {
                        const bool is_ho = fresh(lane) < Nr + Nh;
                        const int m = is_ho ? lane - Nr : lane - Nr - Nh;
                        const int sc = __popcll((is_ho ? col_ho : col_cr) & ((is_ho ? cm_ho : cm_cr) << m));
                        my_caught = sc >= (is_ho ? DA.n_coop_save : 1);
                        my_enc = is_ho && sc >= 1;
                    }
                } else {
                for (int idx = lane; idx < Nr * (Nh + Nc); idx += 64) {
                    const bool is_ho = idx < Nr * Nh;
                    const int r = is_ho ? idx : idx - Nr * Nh;
                    const int n2 = is_ho ? Nh : Nc;
                    const int i = r / n2, m = r % n2;
                    const int j = (is_ho ? Nr : Nr + Nh) + m;
                    (is_ho ? COLH : COLC)[r] = dist2_le(X[2 * i], X[2 * i + 1], X[2 * j], X[2 * j + 1], is_ho ? DA.sq_hit_ho : DA.sq_hit_cr);
                }
                wave_sync();
                if (fresh(lane) >= Nr && fresh(lane) < NP) {
                    const bool is_ho = fresh(lane) < Nr + Nh;
                    const int m = is_ho ? lane - Nr : lane - Nr - Nh;
                    const uint8_t *col = is_ho ? COLH : COLC;
                    const int n2 = is_ho ? Nh : Nc;
                    int s = 0;
                    for (int i = 0; i < Nr; ++i) s += col[i * n2 + m];
                    my_caught = s >= (is_ho ? DA.n_coop_save : 1);
                    my_enc = is_ho && s >= 1;
                    if (is_ho) { FLG[m] = my_caught; FLG[Nh + m] = my_enc; }
                    else FLG[2 * Nh + m] = my_caught;
                }
                }
                const uint64_t ho_lanes = ((Nh >= 64) ? ~0ull : ((1ull << Nh) - 1ull)) << Nr;
                const uint64_t caught_mask = __ballot(my_caught);
                const int n_ho_caught = __popcll(caught_mask & ho_lanes);
                const int n_cr_caught = __popcll(caught_mask & ~ho_lanes);
                const uint64_t enc_mask = __ballot(my_enc);
                const int n_ho_enc = __popcll(enc_mask);
                const bool bo_caught = __ballot(col_bo) != 0ull, ke_caught = __ballot(col_ke) != 0ull;
                wave_sync();
                // phase C: sensing (:295-362).  Rows: [criminal dist | criminal speed | hostage dist | key dist | bomb dist] (:398-400)
                {
                    // passes of 64 (rescuer, sensor) pairs held in registers at a time: no more than the specialised shape needs
                    // ALIGNED (compile-time K <= 64): a pass holds floor(64 / K) whole rescuers, so its reach set is theirs alone (waterworld.hip)
                    constexpr bool ALIGNED = TK > 0 && TK <= 64;
                    constexpr int PPP = ALIGNED ? 64 / (TK > 0 ? TK : 1) : 1;
                    const int n_pass = ALIGNED ? (Nr + PPP - 1) / PPP : (Nr * K + 63) / 64;
                    constexpr int N_PASS_T = TNr > 0 ? (ALIGNED ? (TNr + PPP - 1) / PPP : (TNr * TK + 63) / 64) : 3;
                    const float srange = DA.sensor_range, rad2 = DA.radius * DA.radius;  // G1
                    const float part_x = fresh(lane) < NP ? X[2 * lane] : 0.f, part_y = fresh(lane) < NP ? X[2 * lane + 1] : 0.f;
                    // Conservative cull (as in waterworld.hip): NEAR[i] = objects with d2 <= (rad2 + range^2) * (1 + 1e-4); all others
                    // would yield INFINITY for every sensor of rescuer i and are skipped per pass.
                    {
                        const float thr2 = (rad2 + srange * srange) * 1.0001f + 1e-9f;
                        const float mx = fresh(lane) == NP ? kx : (fresh(lane) == NP + 1 ? bx : part_x), my = fresh(lane) == NP ? ky : (fresh(lane) == NP + 1 ? by : part_y);
                        for (int i = 0; i < Nr; ++i) {
                            const float rx = mx - bcast(part_x, i), ry = my - bcast(part_y, i);
                            const uint64_t mk = __ballot((fresh(lane) <= NP + 1) && (rx * rx + ry * ry <= thr2));
                            if (fresh(lane) == 0) NEAR[i] = mk;
                        }
                        wave_sync();
                    }
                    // ONE PASS AT A TIME (round 6, as in waterworld.hip): a pass walks the set bits of ITS OWN reach mask (ascending = the
                    // reference's index order: the first minimum wins as in np.argmin) instead of the union of the passes' masks with a
                    // test-and-skip per pass and object -- scalar work on the CU's one scalar pipe; lanes without a (rescuer, sensor) pair write
                    // to a spare row instead of branching around the stores.
#pragma unroll
                    for (int pass_q = 0; pass_q < (TNr > 0 ? N_PASS_T : n_pass); ++pass_q) {
                        int i_first, i_last;  // rescuers of this pass
                        bool okq;
                        int iq, kq;
                        if constexpr (ALIGNED) {
                            const int li = lane / K;
                            i_first = pass_q * PPP; i_last = min(i_first + PPP, Nr) - 1;
                            okq = li < PPP && i_first + li <= i_last;
                            iq = okq ? i_first + li : 0;
                            kq = okq ? lane - li * K : 0;
                        } else {
                            const int idx = 64 * pass_q + lane;
                            okq = idx < Nr * K;
                            iq = okq ? idx / K : 0;
                            kq = okq ? idx - iq * K : 0;
                            i_first = 64 * pass_q / K; i_last = min(64 * pass_q + 63, Nr * K - 1) / K;
                        }
                        const float sxq = SEN[2 * kq], syq = SEN[2 * kq + 1];
                        const float pxq = X[2 * iq], pyq = X[2 * iq + 1];
                        uint64_t u = 0ull;
                        for (int i = i_first; i <= i_last; ++i) u |= NEAR[i];
                        // wave-uniform: objects in reach of any rescuer of this pass
                        const uint64_t reach = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u)) |
                                               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32)) << 32);
                        auto sense = [&](float qx, float qy) -> float {
                            const float rx = qx - pxq, ry = qy - pyq;
                            const float sv = sxq * rx + syq * ry;
                            const float d2 = rx * rx + ry * ry;
                            // sv < 0 || sv > srange as ONE compare: the median of (sv, 0, srange) is sv exactly when 0 <= sv <= srange (waterworld.hip)
                            const bool out = (__builtin_amdgcn_fmed3f(sv, 0.f, srange) != sv) | (d2 - sv * sv > rad2);
                            return out ? INFINITY : sv;
                        };
                        float b_cr = INFINITY, b_ho = INFINITY;
                        int a_cr = 0;
                        if (TNr > 0 && Nc <= 32 && Nh <= 32) {   // 32-bit class masks: half the scalar work of the walk
                            uint32_t todo = (uint32_t)(reach >> (Nr + Nh)) & (Nc >= 32 ? 0xFFFFFFFFu : ((1u << Nc) - 1u));
#pragma nounroll
                            while (todo != 0u) {
                                const int m = __builtin_ctz(todo);
                                todo &= todo - 1u;
                                const float sv = sense(bcast(part_x, Nr + Nh + m), bcast(part_y, Nr + Nh + m));
                                const bool better = sv < b_cr;
                                b_cr = better ? sv : b_cr;
                                a_cr = better ? m : a_cr;
                            }
                            // hostages: the saved ones (mask from before this step's processing, G5, :296) are not sensed
                            todo = (uint32_t)(reach >> Nr) & (Nh >= 32 ? 0xFFFFFFFFu : ((1u << Nh) - 1u)) & ~(uint32_t)saved;
#pragma nounroll
                            while (todo != 0u) {
                                const int m = __builtin_ctz(todo);
                                todo &= todo - 1u;
                                const float sv = sense(bcast(part_x, Nr + m), bcast(part_y, Nr + m));
                                b_ho = sv < b_ho ? sv : b_ho;
                            }
                        } else {
                            uint64_t todo = reach & ((((Nc >= 64) ? ~0ull : ((1ull << Nc) - 1ull))) << (Nr + Nh));
#pragma nounroll
                            while (todo != 0ull) {
                                const int bit = __builtin_ctzll(todo);
                                todo &= todo - 1ull;
                                const float sv = sense(bcast(part_x, bit), bcast(part_y, bit));
                                const bool better = sv < b_cr;
                                b_cr = better ? sv : b_cr;
                                a_cr = better ? bit - (Nr + Nh) : a_cr;
                            }
                            todo = reach & (((((Nh >= 64) ? ~0ull : ((1ull << Nh) - 1ull))) & ~saved) << Nr);
#pragma nounroll
                            while (todo != 0ull) {
                                const int bit = __builtin_ctzll(todo);
                                todo &= todo - 1ull;
                                const float sv = sense(bcast(part_x, bit), bcast(part_y, bit));
                                b_ho = sv < b_ho ? sv : b_ho;
                            }
                        }
                        const float b_ke = ((reach >> NP) & 1ull) ? sense(kx, ky) : INFINITY;
                        const float b_bo = ((reach >> (NP + 1)) & 1ull) ? sense(bx, by) : INFINITY;
                        {
                            float *o = okq ? O + iq * D : O_SPARE;
                            const bool fin = b_cr < INFINITY;
                            const int j = Nr + Nh + a_cr;   // (a_cr = 0 without a hit: a valid particle, its value is not used)
                            const float raw = sxq * (V[2 * j] - V[2 * iq]) + syq * (V[2 * j + 1] - V[2 * iq + 1]);   // :204-226; loaded and computed unconditionally: a select, no branch
                            o[kq] = fin ? b_cr : 0.f;
                            o[K + kq] = fin ? raw : 0.f;
                            o[2 * K + kq] = (gate0 && b_ho < INFINITY) ? b_ho : 0.f;   // :320-322
                            o[3 * K + kq] = (!gate0 && b_ke < INFINITY) ? b_ke : 0.f;  // :338-340
                            o[4 * K + kq] = (b_bo < INFINITY) ? b_bo : 0.f;
                        }
                    }
                }
                // rescuer lanes: contact flags and who-caught tests for the local rewards (G9)
                bool w_ho = false, w_enc = false, w_cr = false, t_ho = false, t_cr = false;
                if (fresh(lane) < Nr) {
                    if constexpr (BITROWS) {
                        const uint64_t row_ho = (col_ho >> (lane * Nh)) & ((1ull << Nh) - 1ull);
                        const uint64_t row_cr = (col_cr >> (lane * Nc)) & ((1ull << Nc) - 1ull);
                        t_ho = row_ho != 0ull;
                        t_cr = row_cr != 0ull;
                        w_ho = (row_ho & (caught_mask >> Nr)) != 0ull;          // touches a caught hostage
                        w_enc = (row_ho & (enc_mask >> Nr)) != 0ull;            // touches an encountered hostage
                        w_cr = (row_cr & (caught_mask >> (Nr + Nh))) != 0ull;   // touches a caught criminal
                    } else {
                    for (int j = 0; j < Nh; ++j) {
                        const bool c = COLH[lane * Nh + j];
                        t_ho |= c;
                        w_ho |= c && FLG[j];
                        w_enc |= c && FLG[Nh + j];
                    }
                    for (int j = 0; j < Nc; ++j) {
                        const bool c = COLC[lane * Nc + j];
                        t_cr |= c;
                        w_cr |= c && FLG[2 * Nh + j];
                    }
                    }
                }
                wave_sync();
                // phase D: process collisions (:365-383)
                saved |= (caught_mask & ho_lanes) >> Nr;
                if (fresh(lane) >= Nr + Nh && fresh(lane) < NP && my_caught) {
                    const int m = lane - Nr - Nh;
                    float x, y, u0, u1;
                    if (MODE == 1 && IOA.inj_resp != nullptr && !do_init) {
                        const float *r = IOA.inj_resp + (env * Nc + m) * 4;
                        x = r[0]; y = r[1]; u0 = r[2]; u1 = r[3];
                    } else {
                        const u32x4 r = philox4x32_10(gid, tick, (uint32_t)m, HW_TAG_RESPAWN, DA.k0, DA.k1);
                        x = u24(r.x); y = u24(r.y); u0 = u24(r.z); u1 = u24(r.w);
                    }
                    X[2 * lane] = x; X[2 * lane + 1] = y;
                    V[2 * lane] = (u0 - 0.5f) * DA.bad_speed; V[2 * lane + 1] = (u1 - 0.5f) * DA.bad_speed;
                }
                tick += 1;
                if (bo_caught) flags |= 2u;
                if (ke_caught) flags |= 1u;
                const float gate1 = (flags & 1u) ? 1.f : 0.f, bombed1 = (flags & 2u) ? 1.f : 0.f;  // states after processing (G6)
                // phase E: rewards (:385-396)
                if (fresh(lane) < Nr) {
                    if (DA.reward_global) {
                        reward += ((((float)n_ho_enc * DA.encounter_reward) * gate1 + (float)n_ho_caught * DA.save_reward) +
                                   (float)n_cr_caught * DA.hit_reward) + bombed1 * DA.bomb_reward;
                    } else {
                        if (w_ho) reward += DA.save_reward;
                        if (w_enc) reward += DA.encounter_reward * gate1;
                        if (w_cr) reward += DA.hit_reward;
                        if (col_bo) reward += bombed1 * DA.bomb_reward;
                    }
                }
                wave_sync();
                // phase F: criminals move; velocity flips only if BOTH coordinates left [0,1], no clipping (G7, :402-408)
                if (fresh(lane) >= Nr + Nh && fresh(lane) < NP) {
                    float x = X[2 * lane], y = X[2 * lane + 1], vx = V[2 * lane], vy = V[2 * lane + 1];
                    x = x + vx; y = y + vy;
                    const bool outx = !(x >= 0.f && x <= 1.f), outy = !(y >= 0.f && y <= 1.f);
                    if (outx && outy) { vx = -1.0f * vx; vy = -1.0f * vy; }
                    X[2 * lane] = x; X[2 * lane + 1] = y; V[2 * lane] = vx; V[2 * lane + 1] = vy;
                }
                if (fresh(lane) < Nr) {  // tail of the observation row (:410-425)
                    float *o = O + lane * D + 5 * K;
                    o[0] = t_ho ? 1.f : 0.f; o[1] = t_cr ? 1.f : 0.f; o[2] = col_ke ? 1.f : 0.f; o[3] = col_bo ? 1.f : 0.f;
                    o[4] = gate1;
                    if (DA.addid) o[5] = (float)(lane + 1);
                }
                tstep += 1;  // :427
                const uint64_t all_h = (Nh >= 64) ? ~0ull : ((1ull << Nh) - 1ull);
                const int limit = DA.max_steps > 0 ? DA.max_steps : 1000;  // timestep_limit :118-120
                const bool is_done = (flags & 2u) || ((saved & all_h) == all_h) || tstep >= limit;  // :179-182
                if (is_done && fresh(lane) < Nr) reward += (float)(Nh - __popcll(saved & all_h)) * DA.not_saved_reward;  // :429-430
                wave_sync();

                if (pass == 0) asm volatile("" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt_act));  // pipeline hinge
                // ---------------------------------------------------- outputs
#if MADRL_HW_ABLATE & 8
                if (DA.n_envs < 0)
#endif
                if (MODE == 1 && !do_init) {
                    if (fresh(lane) < Nr) uniform_ptr(IOA.rew + env * Nr)[ulane] = reward;
                    if (fresh(lane) == 0) {
                        IOA.done[env] = (uint8_t)is_done;
                        IOA.info[2 * env] = n_ho_caught;
                        IOA.info[2 * env + 1] = n_cr_caught;
                    }
                    if (is_done && DA.auto_reset) {  // wave-uniform: run the reset pass next
                        npass = 2;
                        do_init = true;
                    }
                }
                if (pass == npass - 1) {
                    const auto orow = uniform_ptr(IOA.obs + env * (int64_t)(Nr * D));
#if MADRL_HW_ABLATE & 1
                    if (DA.n_envs < 0)
#endif
#if MADRL_HW_ABLATE & 2
                    for (uint32_t e = ulane; e < (uint32_t)(Nr * D); e += 64u) __builtin_nontemporal_store(O[e], &orow[e]);
#else
                    for (uint32_t e = ulane; e < (uint32_t)(Nr * D); e += 64u) orow[e] = O[e];
#endif
                }
                wave_sync();
            }
            // ---------------------------------------------------------- LDS -> record
            if (fresh(lane) == 0) {
                SU[OFF_SAVED] = (uint32_t)saved; SU[OFF_SAVED + 1] = (uint32_t)(saved >> 32);
                SU[OFF_FLAGS] = flags;
                SU[OFF_T] = (uint32_t)tstep;
                SU[OFF_TICK] = tick;
            }
            wave_sync();
            {
                const auto dst = uniform_ptr(reinterpret_cast<uint32_t *>(DA.state) + env * (int64_t)rec_dw);
#if MADRL_HW_ABLATE & 4
                if (DA.n_envs < 0)
#endif
                for (uint32_t k = ulane; k < (uint32_t)rec_dw; k += 64u) dst[k] = SU[k];
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

}  // namespace

// =================================================================== host side / C ABI
struct madrl_hostage {
    madrl_hostage_config cfg;
    HwDev dev;
    int device;
    int64_t max_blocks;
    size_t lds_bytes;
    void *tables;
};

namespace {

int hw_validate(const madrl_hostage_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_hostage_config))
        return fail(MADRL_EINVAL, "madrl_hostage_config.struct_size=%d, library expects %d", c->struct_size, (int)sizeof(madrl_hostage_config));
    if (c->n_good < 1 || c->n_hostages < 1 || c->n_bad < 1) return fail(MADRL_EINVAL, "n_good, n_hostages, n_bad must be >= 1");
    // one wavefront per env; the packed record (4 * NP + 9 dwords) is prefetched as 4 dwords per lane
    if (c->n_good + c->n_hostages + c->n_bad > 61) return fail(MADRL_EINVAL, "at most 61 particles per env (one wavefront per env)");
    if (2 * c->n_good > 64) return fail(MADRL_EINVAL, "n_good must be <= 32");
    if (c->n_sensors < 1 || c->n_sensors > 256) return fail(MADRL_EINVAL, "n_sensors must be in 1..256");
    if (c->n_coop_save < 1) return fail(MADRL_EINVAL, "n_coop_save must be >= 1");
    return MADRL_OK;
}

int hw_obs_dim_of(const madrl_hostage_config *c) { return c->n_sensors * 5 + 5 + (c->addid ? 1 : 0); }  // CircAgent.__init__ :19-23

// largest float32 x with sqrtf(x) <= thr (thr >= 0 finite); see HwDev::sq_*
float hw_sq_threshold(float thr) {
    float x = thr * thr;
    while (x > 0.0f && sqrtf(x) > thr) x = nextafterf(x, 0.0f);
    for (;;) {
        const float up = nextafterf(x, INFINITY);
        if (!(sqrtf(up) <= thr)) break;
        x = up;
    }
    return x;
}

void hw_layout(const madrl_hostage_config *c, HwDev *d) {
    memset(d, 0, sizeof(*d));
    d->Nr = c->n_good; d->Nh = c->n_hostages; d->Nc = c->n_bad; d->NP = d->Nr + d->Nh + d->Nc;
    d->K = c->n_sensors; d->D = hw_obs_dim_of(c);
    d->n_coop_save = c->n_coop_save; d->addid = c->addid; d->reward_global = c->reward_global; d->key_fixed = c->key_fixed;
    d->max_steps = c->max_steps; d->auto_reset = c->auto_reset;
    d->rec_dw = (int)align_up((size_t)4 * d->NP + 9, 4);
    d->k0 = (uint32_t)c->seed; d->k1 = (uint32_t)(c->seed >> 32); d->gid_base = (uint32_t)c->env_id_base;
    d->radius = (float)c->radius; d->r_ho = (float)(c->radius * 2); d->gate_lo = (float)(0.5 + c->radius);  // evaluated in float64 like the reference
    d->bad_speed = (float)c->bad_speed; d->sensor_range = (float)c->sensor_range; d->action_scale = (float)c->action_scale;
    d->save_reward = (float)c->save_reward; d->hit_reward = (float)c->hit_reward; d->encounter_reward = (float)c->encounter_reward;
    d->not_saved_reward = (float)c->not_saved_reward; d->bomb_reward = (float)c->bomb_reward; d->bomb_radius = (float)c->bomb_radius;
    d->key_radius = (float)c->key_radius; d->control_penalty = (float)c->control_penalty;
    d->key_x = (float)c->key_loc[0]; d->key_y = (float)c->key_loc[1];
    // the float32 sums the kernel used to form before comparing
    d->sq_hit_ho = hw_sq_threshold(d->radius + d->r_ho); d->sq_hit_cr = hw_sq_threshold(d->radius + d->radius);
    d->sq_bomb = hw_sq_threshold(d->radius + d->bomb_radius); d->sq_key = hw_sq_threshold(d->radius + d->key_radius);
}

size_t hw_lds_bytes(const HwDev &d) {
    size_t f = align_up((size_t)d.rec_dw, 4) + align_up((size_t)(d.Nr + 1) * d.D, 4) + align_up((size_t)2 * d.K, 4);   // (Nr + 1: the spare row of the sensing phase)
    size_t b = f * 4 + 8 * (size_t)d.Nr + (size_t)d.Nr * (d.Nh + d.Nc) + 2 * (size_t)d.Nh + d.Nc;
    return align_up(b, 16);
}

int hw_launch(const madrl_hostage *h, const HwIO &io, int mode, void *stream) {
    int64_t blocks = h->max_blocks > 0 ? h->max_blocks : 256 * 64;
    if (blocks > h->dev.n_envs) blocks = h->dev.n_envs;
    hipStream_t s = (hipStream_t)stream;
    const HwDev &d = h->dev;
    const bool ex = d.Nr == 3 && d.Nh == 10 && d.Nc == 5 && d.K == 30 && d.D == 156;  // the module's own configuration (hostage.py:483), 30 sensors, agent id
    if (mode == 0) {
        if (ex) hipLaunchKernelGGL((hostage_kernel<0, 3, 10, 5, 30, 156>), dim3((unsigned)blocks), dim3(64), 0, s, h->dev, io);
        else hipLaunchKernelGGL((hostage_kernel<0, 0, 0, 0, 0>), dim3((unsigned)blocks), dim3(64), h->lds_bytes, s, h->dev, io);
    } else {
        if (ex) hipLaunchKernelGGL((hostage_kernel<1, 3, 10, 5, 30, 156>), dim3((unsigned)blocks), dim3(64), 0, s, h->dev, io);
        else hipLaunchKernelGGL((hostage_kernel<1, 0, 0, 0, 0>), dim3((unsigned)blocks), dim3(64), h->lds_bytes, s, h->dev, io);
    }
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

__global__ void hw_state_copy_kernel(const HwDev d, float *pos, float *vel, float *key, float *bomb, uint64_t *saved, uint8_t *flags, int32_t *t,
                                     uint32_t *tick, const int to_state) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    float *rec = d.state + env * (int64_t)d.rec_dw;
    uint32_t *ru = reinterpret_cast<uint32_t *>(rec);
    const int NP = d.NP;
    for (int k = 0; k < 2 * NP; ++k) {
        if (pos) { if (to_state) rec[k] = pos[env * 2 * NP + k]; else pos[env * 2 * NP + k] = rec[k]; }
        if (vel) { if (to_state) rec[2 * NP + k] = vel[env * 2 * NP + k]; else vel[env * 2 * NP + k] = rec[2 * NP + k]; }
    }
    for (int k = 0; k < 2; ++k) {
        if (key) { if (to_state) rec[4 * NP + k] = key[env * 2 + k]; else key[env * 2 + k] = rec[4 * NP + k]; }
        if (bomb) { if (to_state) rec[4 * NP + 2 + k] = bomb[env * 2 + k]; else bomb[env * 2 + k] = rec[4 * NP + 2 + k]; }
    }
    if (saved) {
        if (to_state) { ru[4 * NP + 4] = (uint32_t)saved[env]; ru[4 * NP + 5] = (uint32_t)(saved[env] >> 32); }
        else saved[env] = (uint64_t)ru[4 * NP + 4] | ((uint64_t)ru[4 * NP + 5] << 32);
    }
    if (flags) { if (to_state) ru[4 * NP + 6] = flags[env]; else flags[env] = (uint8_t)ru[4 * NP + 6]; }
    if (t) { if (to_state) ru[4 * NP + 7] = (uint32_t)t[env]; else t[env] = (int32_t)ru[4 * NP + 7]; }
    if (tick) { if (to_state) ru[4 * NP + 8] = tick[env]; else tick[env] = ru[4 * NP + 8]; }
}

}  // namespace

extern "C" {

int madrl_hostage_obs_dim(const madrl_hostage_config *cfg, int32_t *out_dim) {
    int rc = hw_validate(cfg);
    if (rc) return rc;
    if (!out_dim) return fail(MADRL_EINVAL, "out_dim is NULL");
    *out_dim = hw_obs_dim_of(cfg);
    return MADRL_OK;
}

int madrl_hostage_state_bytes(const madrl_hostage_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    int rc = hw_validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_bytes) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_bytes non-NULL");
    HwDev d;
    hw_layout(cfg, &d);
    *out_bytes = (uint64_t)d.rec_dw * 4u * (uint64_t)n_envs;
    return MADRL_OK;
}

int madrl_hostage_create(const madrl_hostage_config *cfg, const double *sensors_host, int64_t n_envs, int32_t device, void *state_dev,
                         madrl_hostage **out) {
    int rc = hw_validate(cfg);
    if (rc) return rc;
    if (!sensors_host || !state_dev || !out || n_envs < 1) return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    if (n_envs >= 0x7FF00000ll)  // the kernel indexes envs with 32-bit integers (index + workgroup count must stay below 2^31)
        return fail(MADRL_EINVAL, "n_envs=%lld is too large for one handle (limit 2146435071); shard the batch", (long long)n_envs);
    if (n_envs + cfg->env_id_base > 0xFFFFFFFFll) return fail(MADRL_EINVAL, "global env index must fit 32 bits");
    MADRL_HIP_TRY(hipSetDevice(device));
    madrl_hostage *h = new (std::nothrow) madrl_hostage();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    hw_layout(cfg, &h->dev);
    h->dev.n_envs = n_envs;
    h->dev.state = (float *)state_dev;
    h->lds_bytes = hw_lds_bytes(h->dev);
    h->max_blocks = 0;
    if (h->lds_bytes > 64 * 1024) {
        const size_t need = h->lds_bytes;
        delete h;
        return fail(MADRL_EINVAL, "configuration needs %zu B of LDS (> 64 KiB)", need);
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

void madrl_hostage_destroy(madrl_hostage *h) {
    if (!h) return;
    if (h->tables) (void)hipFree(h->tables);
    delete h;
}

int madrl_hostage_set_launch(madrl_hostage *h, int64_t max_blocks) {
    if (!h || max_blocks < 0) return fail(MADRL_EINVAL, "set_launch: bad argument");
    h->max_blocks = max_blocks;
    return MADRL_OK;
}

int madrl_hostage_reset(madrl_hostage *h, const uint8_t *mask_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    HwIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.obs = obs_dev;
    return hw_launch(h, io, 0, stream);
}

int madrl_hostage_step(madrl_hostage *h, const float *actions_dev, const float *inj_respawn_dev, float *obs_dev, float *rew_dev,
                       uint8_t *done_dev, int32_t *info_dev, void *stream) {
    if (!h || !actions_dev || !obs_dev || !rew_dev || !done_dev || !info_dev) return fail(MADRL_EINVAL, "step: NULL argument");
    HwIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions_dev;
    io.inj_resp = inj_respawn_dev;
    io.obs = obs_dev;
    io.rew = rew_dev;
    io.done = done_dev;
    io.info = info_dev;
    return hw_launch(h, io, 1, stream);
}

int madrl_hostage_get_state(madrl_hostage *h, float *pos, float *vel, float *key, float *bomb, uint64_t *saved, uint8_t *flags, int32_t *t,
                            uint32_t *tick, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(hw_state_copy_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, pos, vel, key, bomb, saved, flags, t, tick, 0);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_hostage_set_state(madrl_hostage *h, const float *pos, const float *vel, const float *key, const float *bomb, const uint64_t *saved,
                            const uint8_t *flags, const int32_t *t, const uint32_t *tick, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(hw_state_copy_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, (float *)pos, (float *)vel, (float *)key,
                       (float *)bomb, (uint64_t *)saved, (uint8_t *)flags, (int32_t *)t, (uint32_t *)tick, 1);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // extern "C"
