// pursuit_wave.hpp -- compile-time-specialised PursuitEvade kernel: ONE WAVEFRONT = ONE ENV.
//
// Used when the configuration matches one of the instantiations listed in
// pursuit_specializations.def (n_pursuers + n_evaders <= 64 lanes, odd obs_range, observation
// row length divisible by 4); everything else runs on the generic kernel in pursuit.hip.
// Both kernels share the packed state record, so they are interchangeable step by step
// (tests/test_pursuit_gpu.py checks them against each other and against the oracle).
//
// Design (DESIGN.md "pursuit_wave_kernel"):
//   * 64-thread workgroups (__launch_bounds__(64)): lane a < P is pursuer a, lane P+i is evader
//     slot i; every barrier is wave-local.  Workgroups are persistent and stride over envs,
//     so the per-map tables and each lane's observation slot constants are built once.
//   * LDS holds three PADDED dword layers addressed as one array L[3*GSZ]:
//       layer 0  map as float bits (0 free, 1/norm building, 1/norm outside the map), static;
//       layer 1  pursuer counts, layer 2 evader counts: integers while the dynamics run
//                (ds_add atomics), overwritten IN PLACE by their float32 observation values
//                (count / layer_norm, table lookup) right before the observation pass;
//                cells outside the map hold the sentinel 0xFFFFFFFF = "do not store" (Q2).
//     Catch credit (purs_sur) needs no extra layer: a caught evader adds 0x10000 to its own
//     evader-count cell and pursuers look at their four neighbours.
//     After the observation pass each agent lane zeroes its own cell again, so no per-env
//     re-initialisation of the layers is needed.
//   * Observation row of an env = P*D floats = P*D/4 float4 slots, slot q -> lane q % 64:
//     every store instruction writes 64 consecutive float4 (1 KiB, fully coalesced).  Per
//     slot a lane keeps 4 LDS offsets relative to the owning pursuer's window origin; the
//     pursuer's origin comes from its lane by ds_bpermute.  A float4 that contains a stale
//     (out-of-map) cell falls back to per-dword conditional stores.
#pragma once

#include <type_traits>

#include "common.hpp"

namespace madrl {
namespace pw {

// Register allocation for this many resident wavefronts per SIMD.  Measured at BASELINE C2 (scripts/sweep_wave.py): 4 waves (105
// VGPRs, 4 096 persistent workgroups) 87.8 us per launch, 5 waves (96 VGPRs, no scratch; 5 120 workgroups = exactly the resident
// capacity) 77.0 us, 6 waves (80 VGPRs, 32 B of scratch; 6 144 workgroups) 80.2 us.
#ifndef MADRL_PW_WAVES
#define MADRL_PW_WAVES 5
#endif
// Shapes with more than 5 float4 slots per lane keep 4 waves (at 96 VGPRs they would spill, and spill stores reach HBM).
#define MADRL_PW_OCC __attribute__((amdgpu_waves_per_eu(S::OCC, S::OCC)))

constexpr uint32_t SENT = 0xFFFFFFFFu;   // layer 1/2 outside the map: element is not stored
constexpr uint32_t CAUGHT = 0x10000u;    // added to an evader-count cell by a caught evader

struct WaveDev {
    int32_t n_catch, surround, reward_global, sample_maps, n_maps, max_steps, auto_reset;
    int32_t fmap_stride;  // dwords per map entry in fmaps
    int32_t max_opponents;  // > 0: random_opponents (pursuit_evade.py:177-181)
    int32_t reverse;      // 1: walk the envs from the last to the first (see launch() in pursuit.hip)
    uint32_t k0, k1, gid_base;
    double catchr, term_pursuit, urgency, cw;
    int64_t n_envs;
    const uint32_t *fmaps;   // per map: padded float layer [GSZ] then need_to_surround [XS*YS] as u32
    const float *vtab;       // fl32(k / layer_norm), k = 0..255
    const uint32_t *codes;   // D entries: bit31 = relative to window origin, low bits = dword offset
    // Host-built launch constants, so that the per-workgroup preamble is loads, not index arithmetic (every resident wavefront
    // runs it at the same moment, 2.8 us of a 78 us launch when it was computed in the kernel):
    const uint32_t *cnt_tmpl;  // [GSZ] an empty count layer: 0 inside the map, SENT outside
    const uint32_t *slot_tab;  // [NS][6][NT] per thread and float4 slot: 4 cell offsets, element-3-is-relative flag, owning pursuer
    const double *cw_env;      // per-env constraint_window / catchr (curriculum) or nullptr: the scalars above.  Read with scalar loads
    const double *catchr_env;  // (constant address space): written by the host or an earlier launch, never by these kernels
    uint8_t *state;
    uint32_t *flags;         // [n_envs] flag words of the step launches (done_flag_word, common.hpp)
    uint32_t *zmask;         // [n_envs][64]: per lane, which of its observation cells hold a NON-ZERO stale value (see "stale-zero mask")
};

struct WaveIO {
    const uint8_t *mask;
    const int32_t *inj_pos;
    const int32_t *inj_map;
    const int32_t *actions;
    const int32_t *inj_eact;
    float *obs;
    float *rew;
    uint8_t *done;
    int32_t *removed;
    int32_t flex;            // host side only: 1 = launch the FLEX instantiation (injected evader actions and / or per-env catchr)
    int32_t control_evaders; // host side only: 1 = launch the evader-control instantiations (madrl_pursuit_config::control_evaders)
};

// a wave-uniform float64 read through the scalar cache (s_load_dwordx2, counted by lgkmcnt -- not by the vmcnt the
// store pipeline of these kernels is scheduled around)
__device__ __forceinline__ double sload_f64(const double *p, int64_t index) {
    return ((const __attribute__((address_space(4))) double *)(uint64_t)p)[index];
}

// Launch parameters that only the rare paths read (episode reset, a change of map) are not kept in SGPRs across the env loop: the
// kernel reads them from its kernel-argument segment (scalar loads) where they are needed.  The loop otherwise runs out of SGPRs
// and the compiler parks the excess in VGPR lanes -- one v_readlane_b32 (a VALU issue slot) per use.
struct KArgs {
    WaveDev d;
    WaveIO io;
};
typedef const __attribute__((address_space(4))) KArgs *KArgsPtr;
__device__ __forceinline__ KArgsPtr cold_args() {
    KArgsPtr p = (KArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // a fresh pointer at every call: the loads cannot be merged with the ones at kernel entry or hoisted
    return p;
}

template <int XS_, int YS_, int P_, int E_, int R_, int FLATTEN_>
struct Shape {
    static constexpr int XS = XS_, YS = YS_, P = P_, E = E_, A = P_ + E_, R = R_, FLATTEN = FLATTEN_;
    static constexpr int OFF = (R - 1) / 2;
    static constexpr int PAD = OFF > 1 ? OFF : 1;
    static constexpr int GW = YS + 2 * PAD;
    static constexpr int GH = XS + 2 * PAD;
    static constexpr int GSZ = (GH * GW + 3) / 4 * 4;           // dwords per layer
    static constexpr int D = FLATTEN ? 3 * R * R + 1 : 4 * R * R;  // include_id is implied
    static constexpr int DV = D / 4;                             // float4 per pursuer row
    static constexpr int NQ = P * DV;                            // float4 slots per env
    static constexpr int NS = (NQ + 63) / 64;                    // slots per lane
    static constexpr int MWORDS = 1;                             // stale-zero mask dwords per lane
    static constexpr int OCC = NS <= 5 ? MADRL_PW_WAVES : (MADRL_PW_WAVES < 4 ? MADRL_PW_WAVES : 4);  // resident wavefronts per SIMD aimed at
    static constexpr int X_FILL = 3 * GSZ;                       // extras after the layers
    static constexpr int X_SKIP = 3 * GSZ + 1;
    static constexpr int X_ID = 3 * GSZ + 2;                     // P id values
    static constexpr int X_VTAB = (X_ID + P + 3) / 4 * 4;        // 72 count values
    static constexpr int NVT = 72;
    static constexpr int X_NEED = X_VTAB + NVT;                  // XS*YS bytes, as dwords
    static constexpr int X_OBSV = X_NEED + (XS * YS + 3) / 4;    // evader control: window origin of the observer of row p (P dwords)
    static constexpr int LDS_DWORDS = X_OBSV + (P + 3) / 4 * 4;
    // packed state record, identical to the generic kernel's layout() in pursuit.hip
    static constexpr int NGW = (E + 31) / 32 > 0 ? (E + 31) / 32 : 1;
    static constexpr int NTW = (A + 31) / 32;
    static constexpr int OFF_GONE = (16 + 2 * A + 3) / 4 * 4;
    static constexpr int OFF_TERM = OFF_GONE + 4 * NGW;
    static constexpr int REC_BYTES = (OFF_TERM + 4 * NTW + 15) / 16 * 16;
    static constexpr int REC_DW = REC_BYTES / 4;
    static_assert(REC_DW <= 64, "the whole record must fit one dword per lane");
    static_assert(A <= 64, "one wavefront per env: n_pursuers + n_evaders must fit 64 lanes");
    static_assert(R % 2 == 1, "odd obs_range only (even ranges run on the generic kernel)");
    static_assert(D % 4 == 0, "observation row must be a whole number of float4");
    static_assert(LDS_DWORDS * 4 <= 64 * 1024, "LDS budget");
    static_assert(NS <= 8, "stale-zero mask: one bit per slot in each byte of the lane's mask dword");
};

__device__ __forceinline__ double pairwise8(const double *r) {
    return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

// numpy's float64 add.reduce order over P values held in registers (see pursuit.hip)
template <int P>
__device__ __forceinline__ double np_sum_regs(const double (&a)[P]) {
    if constexpr (P < 8) {
        double res = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) res += a[i];
        return res;
    } else {
        double r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = a[i];
        constexpr int LIM = P - (P % 8);
#pragma unroll
        for (int i = 8; i < LIM; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        }
        double res = pairwise8(r);
#pragma unroll
        for (int i = LIM; i < P; ++i) res += a[i];
        return res;
    }
}

// Wave-local synchronisation.  A workgroup is one wavefront, its DS (LDS) instructions are
// executed in issue order, so cross-lane LDS hand-offs only need the COMPILER to keep the
// order; unlike __syncthreads() this emits no s_waitcnt vmcnt(0), i.e. the wave never waits
// for its observation stores to reach HBM.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Register-pressure control.  The compiler hoists every loop-invariant lane compare (lane < P, lane == k, ...)
// out of the env loop as an SGPR-pair mask and, with ~100 uniform values already live there, spills them to
// VGPR lanes: each use then costs two v_readlane_b32 (VALU issue slots, the resource this kernel is bound by).
// fresh(lane) hides the invariance, so a predicate is one v_cmp at its use site and dies there.
__device__ __forceinline__ int fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint32_t fresh_s(uint32_t v) {
    asm volatile("" : "+s"(v));
    return v;
}

// A wave-uniform pointer pinned to an SGPR pair at this point of the program.  Per-lane accesses written as
// uniform_ptr(base + env * stride)[lane] then select the "SGPR base + 32-bit VGPR offset" addressing form; without the pin the
// compiler reassociates to (base + lane * 4) + env * stride, keeps one 64-bit VGPR pair per array live across the env loop and,
// at 96 VGPRs, spills them -- and a scratch reload in the loop waits (in-order vmcnt) for the previous env's observation stores.
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *uniform_ptr(T *p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (__attribute__((address_space(1))) T *)(((uint64_t)hi << 32) | lo);  // global address space: global_*, not flat_*, instructions
}

// Masked global stores that are ALWAYS issued (exec narrowed inside the asm, no compiler-made skip branch around them): the
// number of vector-memory instructions per env iteration is then a compile-time constant, which is what lets the record
// prefetch wait with an exact s_waitcnt vmcnt(N) instead of vmcnt(0) (see "software pipeline").  base: wave-uniform pointer,
// voff: per-lane byte offset, OFF: immediate byte offset (< 4096).  The trailing s_nop covers the "store of more than 8 bytes,
// then a VALU write of its data registers" hazard that the compiler cannot see through inline asm.
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int OFF>
__device__ __forceinline__ void store4_nt_masked(const void *base, uint32_t voff, v4f_t val, uint64_t mask) {
    uint64_t sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dwordx4 %2, %3, %4 offset:%5 nt\n\ts_mov_b64 exec, %0\n\ts_nop 0"
                 : "=&s"(sv) : "s"(mask), "v"(voff), "v"(val), "s"(base), "n"(OFF) : "scc");  // s_and_b64 writes SCC
}
// the four elements of one float4 slot, each under its own lane mask, as plain (L2-merged) dword stores
template <int OFF>
__device__ __forceinline__ void store1x4_masked(const void *base, uint32_t voff, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3,
                                                uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
    uint64_t sv;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, %0, %1\n\tglobal_store_dword %5, %6, %10 offset:%11\n\t"
                 "s_and_b64 exec, %0, %2\n\tglobal_store_dword %5, %7, %10 offset:%11+4\n\t"
                 "s_and_b64 exec, %0, %3\n\tglobal_store_dword %5, %8, %10 offset:%11+8\n\t"
                 "s_and_b64 exec, %0, %4\n\tglobal_store_dword %5, %9, %10 offset:%11+12\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(voff), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(base), "n"(OFF) : "scc");
}

// Record prefetch with an exact wait (step kernel).  vmcnt counts loads and stores alike and retires them in issue order, and the
// compiler, which cannot know how many of the masked stores above were issued, would wait for a prefetched record with
// vmcnt(0): every wavefront then stalls until its previous env's observation stores have been acknowledged, half an iteration
// after issuing them.  Here the three loads of the next env are issued (inline asm, so the compiler inserts no wait of its own)
// at the top of an iteration and waited for at its END with s_waitcnt vmcnt(<stores per iteration>): only the stores of the
// PREVIOUS iteration -- a whole iteration old -- must have retired, this iteration's may all be in flight.
// Every lane loads (clamped offsets, no exec games); s_nop 4 covers a base pointer that a VALU instruction may have just written.
__device__ __forceinline__ void prefetch3(uint32_t &r0, uint32_t &r1, uint32_t &r2, const void *b0, uint32_t o0, const void *b1, uint32_t o1,
                                          const void *b2, uint32_t o2) {
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %3, %4\n\tglobal_load_dword %1, %5, %6\n\tglobal_load_dword %2, %7, %8"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(o0), "s"(b0), "v"(o1), "s"(b1), "v"(o2), "s"(b2));
}
template <int N>
__device__ __forceinline__ void prefetch_wait(uint32_t &r0, uint32_t &r1, uint32_t &r2) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r0), "+v"(r1), "+v"(r2) : "n"(N) : "memory");
}

// w[lane LN] = v for a wave-uniform v: one v_writelane_b32, no lane mask, no compare
template <int LN>
__device__ __forceinline__ void put_lane(uint32_t &w, uint32_t v) {
    const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
    asm("v_writelane_b32 %0, %1, %2" : "+v"(w) : "s"(sv), "n"(LN));
}
template <int LN, int END>
__device__ __forceinline__ void put_zero_from(uint32_t &w) {
    if constexpr (LN < END) {
        put_lane<LN>(w, 0u);
        put_zero_from<LN + 1, END>(w);
    }
}

// Profiling aid (scripts/variants.sh builds one library per value, never the shipped one):
//   1 no observation stores   2 store stale cells too (all float4 full, nt)   4 no Philox
//   8 no observation pass at all   16 no record / reward stores
//   64 no record prefetch (timing only, wrong results)
#ifndef MADRL_ABLATE
#define MADRL_ABLATE 0
#endif
// Timing aids (same rules): 4 return at once (launch cost)   8 return after the per-workgroup preamble
#ifndef MADRL_PW_EXP
#define MADRL_PW_EXP 0
#endif

// Stale-zero mask (quirk Q2).  A cell of channel 1 / 2 outside the map is not written: it keeps the value of the last time it
// was inside.  A float4 slot that mixes written and unwritten cells becomes a partial store, which the memory side turns into
// a read-modify-write of the sector (about 20 us of the 90 us launch at BASELINE C2).  Most stale values are 0.0 -- counts of
// mostly empty cells -- and storing 0.0 over a stale 0.0 changes nothing.  So the kernel keeps, per lane, one bit per observation
// cell it owns (5 slots x 4 cells; dword [env][lane] of `zmask`, prefetched and written back with the state record):
// "the value this cell holds in the caller's observation buffer is not known to be zero".  A cell inside the map sets its bit to
// (value != 0), a cell outside keeps it.  A slot is stored as ONE non-temporal float4, zeros in its outside cells, unless an
// outside cell's bit is set; only those slots (about 3 % instead of 19 %) fall back to masked dword stores.  All bits set =
// "nothing known" is always correct: the library starts there and returns there whenever the caller hands it another buffer.
//
// MODE 0: reset(mask)      MODE 1: step (+ fused auto-reset)
// INJECT (step only) = the FLEXIBLE instantiation: evader actions come from io.inj_eact when it is given (parity harness)
// instead of Philox, and the catch reward is d.catchr_env[env] when per-env curriculum arrays are bound.
// It is a template parameter because a conditional global load in the hot loop makes the
// compiler's s_waitcnt pass put a vmcnt(0) on the common path (see "pipeline hinge" below).
// CTRL = evader control (train_pursuit=False, pursuit_evade.py:105-112, :204-241, :418-428): the P actions drive the first P REMAINING
// evaders of the layer, the pursuers move by their controller (io.inj_eact [n_envs][P] or Philox, TAG_PURSUER_ACT), observation row k
// shows the window of the k-th remaining evader among slots 0..P-1 and the rows past the last one stay untouched; rewards stay the
// pursuers'.  Only instantiated for the FLEX step kernel and the reset kernel of shapes with E >= P.
template <class S, int MODE, bool INJECT, bool CTRL = false>
__global__ __launch_bounds__(64) MADRL_PW_OCC void pursuit_wave_kernel(const WaveDev d, const WaveIO io) {
    static_assert(!CTRL || (S::E >= S::P && (MODE == 0 || INJECT)), "evader control: n_evaders >= n_pursuers, flexible instantiation");
    constexpr int P = S::P, E = S::E, A = S::A, GW = S::GW, PAD = S::PAD, GSZ = S::GSZ, NS = S::NS;
    __shared__ __attribute__((aligned(16))) uint32_t L[S::LDS_DWORDS];
    const int lane = threadIdx.x;
    const uint32_t ulane = threadIdx.x;  // as an unsigned 32-bit index: zero-extends into the VGPR-offset addressing form
    const bool is_p = lane < P;
    const int eslot = lane - P;

#if MADRL_PW_EXP & 4
    if (d.n_envs > 0) return;
#endif
    // ---------------------------------------------------------------- once per workgroup
    for (int k = lane; k < GSZ; k += 64) {  // count layers: 0 inside the map, SENT outside
        const uint32_t v = d.cnt_tmpl[k];
        L[GSZ + k] = v;
        L[2 * GSZ + k] = v;
    }
    if (lane == 0) {
        L[S::X_FILL] = d.fmaps[0];  // a corner of the padded map layer is always outside the map
        L[S::X_SKIP] = SENT;
    }
    if (lane < P) L[S::X_ID + lane] = __float_as_uint((float)((double)lane / (double)P));  // :440-445
    for (int k = lane; k < S::NVT; k += 64) L[S::X_VTAB + k] = __float_as_uint(d.vtab[k]);
    // observation slot constants (registers, live across the env loop)
    int s_cst[NS][4];
    int s_rel3[NS];   // 1: element 3 is window-relative, 0: absolute (id / skip cell)
    int s_src[NS];    // ds_bpermute byte address of the owning pursuer's lane
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const uint32_t *t = d.slot_tab + s * 6 * 64 + lane;  // slots past the end of the row read harmless cells and are never stored
#pragma unroll
        for (int k = 0; k < 4; ++k) s_cst[s][k] = (int)t[64 * k] * 4;  // byte offsets into L
        s_rel3[s] = (int)t[64 * 4];
        s_src[s] = (int)t[64 * 5] * 4;
    }
    // Map 0 is staged with the tables above (its loads are in flight together with theirs): with a single map -- every BASELINE
    // config -- the first env then needs no second round trip to HBM for a map chosen by its record, at a moment when every
    // wavefront of the launch waits for the same thing.  With a map pool the first env may reload (load_map below).
    for (int k = lane; k < GSZ; k += 64) L[k] = d.fmaps[k];
    for (int k = lane; k < (S::XS * S::YS + 3) / 4; k += 64) L[S::X_NEED + k] = d.fmaps[GSZ + k];
    int cached_map = 0;
    const uint8_t *need_tab = reinterpret_cast<const uint8_t *>(&L[S::X_NEED]);
    uint32_t *const layer = &L[is_p ? GSZ : 2 * GSZ];  // this lane's count layer
    // which lanes feed my dword of the packed state record (agents 2j and 2j+1 for dword 4+j)
    const int rec_src0 = (lane >= 4 ? 2 * (lane - 4) : 0) * 4, rec_src1 = rec_src0 + 4;

    // ---------------------------------------------------------------- software pipeline
    // The whole packed state record (S::REC_DW dwords) is fetched by ONE coalesced load, lane k
    // holding dword k, together with the pursuer action of the same env; both are issued one
    // env ahead so their HBM latency hides behind the current env's work.
    auto isP = [&]() { return fresh(lane) < P; };
    auto isE = [&]() { return (unsigned)(fresh(lane) - P) < (unsigned)E; };
    auto isAgent = [&]() { return fresh(lane) < A; };
    auto fetch_rec = [&](int64_t env) -> uint32_t {
        return (fresh(lane) < S::REC_DW) ? uniform_ptr(reinterpret_cast<const uint32_t *>(d.state + env * (int64_t)S::REC_BYTES))[ulane] : 0u;
    };
    auto fetch_act = [&](int64_t env) -> int {
        if constexpr (MODE == 1) return isP() ? uniform_ptr(io.actions + env * P)[ulane] : 4;
        else return 4;
    };
    auto fetch_zm = [&](int64_t env) -> uint32_t { return uniform_ptr(d.zmask + env * 64)[ulane]; };
    uint32_t cur_rec = 0, cur_zm = 0xFFFFFFFFu;
    int cur_act = 4;
    // env indices are 32-bit (the C ABI caps n_envs below 2^31); only byte offsets are 64-bit
    const int n_envs = (int)d.n_envs, stride = (int)gridDim.x;
    auto phys = [&](int e) -> int64_t { return (int64_t)(d.reverse ? n_envs - 1 - e : e); };
    if ((int)blockIdx.x < n_envs) {
        cur_rec = fetch_rec(phys(blockIdx.x));
        cur_act = fetch_act(phys(blockIdx.x));
        cur_zm = fetch_zm(phys(blockIdx.x));
    }
    asm volatile("" : "+v"(cur_rec), "+v"(cur_act), "+v"(cur_zm));  // loads complete before the loop (see hinge below)
    wave_sync();
#if MADRL_PW_EXP & 8
    if (d.n_envs > 0) { if (s_cst[0][0] + s_cst[NS - 1][3] + s_rel3[0] + s_src[NS - 1] + (int)cur_rec == 0x12345) io.rew[0] = 1.f; return; }
#endif

    // PIPE: the exact-wait prefetch above (production step kernel); the reset kernel (envs may be skipped) and the parity
    // harness variant keep compiler-managed loads and the mid-iteration hinge.
    constexpr bool PIPE = (MODE == 1) && !INJECT && !(MADRL_ABLATE & (1 | 8 | 16 | 64));
    constexpr int VM_PER_ENV = 5 * NS + 6;  // stores every step iteration issues: NS x (4 dword + 1 float4), reward, done, removed, flag word, record, mask
    static_assert(!PIPE || VM_PER_ENV < 64, "vmcnt range");
    const uint32_t rec_off = (ulane < (uint32_t)S::REC_DW ? ulane : (uint32_t)S::REC_DW - 1u) * 4u;
    const uint32_t act_off = (ulane < (uint32_t)P ? ulane : (uint32_t)P - 1u) * 4u;
    for (int e = blockIdx.x; e < n_envs; e += stride) {
        const int64_t env = phys(e);
        const bool has_next = e + stride < n_envs;  // n_envs + number of workgroups < 2^31
        const int64_t nenv = phys(has_next ? e + stride : e);
        uint32_t nxt_rec = 0, nxt_zm = 0xFFFFFFFFu;
        int nxt_act = 4;
        uint32_t pf_rec, pf_act, pf_zm;
        if constexpr (PIPE) {
            prefetch3(pf_rec, pf_act, pf_zm, d.state + nenv * (int64_t)S::REC_BYTES, rec_off, io.actions + nenv * P, act_off,
                      d.zmask + nenv * 64, ulane * 4u);
        } else {
#if MADRL_ABLATE & 64
            nxt_rec = cur_rec ^ (uint32_t)(fresh(lane) == 0);  // no loads in the loop: is the in-order vmcnt drain what serialises a wave?
            nxt_act = cur_act;
#else
            if (has_next) {
                nxt_rec = fetch_rec(nenv);
                nxt_act = fetch_act(nenv);
                nxt_zm = fetch_zm(nenv);
            }
#endif
        }
        bool skip = false;
        if constexpr (MODE == 0) skip = (io.mask != nullptr && io.mask[env] == 0);
        if (!skip) {
            // -------------------------------------------------------- unpack the record
            uint32_t tick = __builtin_amdgcn_readlane(cur_rec, 0);
            int32_t tstep = (int32_t)__builtin_amdgcn_readlane(cur_rec, 1);
            int32_t map_id = (int32_t)__builtin_amdgcn_readlane(cur_rec, 2);
            const uint32_t xyw = (uint32_t)__shfl((int)cur_rec, 4 + (lane >> 1));
            const uint32_t xy = (lane & 1) ? (xyw >> 16) : (xyw & 0xFFFFu);
            int x = (int)(xy & 0xFF), y = (int)(xy >> 8);
            uint64_t gone = (uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_GONE / 4);  // (uint32_t): readlane returns int, bit 31 must not sign-extend
            if constexpr (S::NGW > 1) gone |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_GONE / 4 + 1) << 32;
            uint64_t term = (uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_TERM / 4);
            if constexpr (S::NTW > 1) term |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_TERM / 4 + 1) << 32;
            const uint32_t gid = d.gid_base + (uint32_t)env;
            const uint32_t k0 = fresh_s(d.k0), k1 = fresh_s(d.k1);  // round keys recomputed on the SALU, not kept live
            bool do_reset = (MODE == 0);
            uint32_t done_bits = 0;
            float rew_out = 0.0f;
            int n_removed = 0;
            bool alive = isP() || (isE() && !((gone >> eslot) & 1ull));
            int cell = (x + PAD) * GW + y + PAD;

            auto load_map = [&](int mid) {
                if (cached_map == mid) return;
                const KArgsPtr ka = cold_args();
                const uint32_t *src = ka->d.fmaps + (int64_t)mid * ka->d.fmap_stride;
                for (int k = lane; k < GSZ; k += 64) L[k] = src[k];
                for (int k = lane; k < (S::XS * S::YS + 3) / 4; k += 64) L[S::X_NEED + k] = src[GSZ + k];
                cached_map = mid;
                wave_sync();
            };
            load_map(map_id);

            if constexpr (MODE == 1) {
                const bool e_alive = alive && !isP();
                // ---------------------------------------------------- pre-move reward (:359-381)
                if (e_alive) atomicAdd(&layer[cell], 1u);
                wave_sync();
                int kpre = 0;
                if (isP()) {  // np.clip keeps a border pursuer on its own cell (:374-380)
                    const int dxm = (x > 0) ? GW : 0, dxp = (x < S::XS - 1) ? GW : 0;
                    const int dym = (y > 0) ? 1 : 0, dyp = (y < S::YS - 1) ? 1 : 0;
                    const uint32_t *ec = &L[2 * GSZ];
                    kpre = (int)(ec[cell - dxm] + ec[cell + dxp] + ec[cell + dyp] + ec[cell - dym]);
                }
                wave_sync();
                if (e_alive) atomicSub(&layer[cell], 1u);
                // ---------------------------------------------------- moves (:229-241), branch-free
                // evader draw: index in the evader layer = alive evaders in lower slots
                const int kidx = __popcll((~gone) & ((1ull << (eslot & 63)) - 1ull));
                int act = cur_act;
                bool injected = false;
                if constexpr (CTRL) {
                    // action k belongs to the k-th remaining evader (:229-230 with agent_layer = evader_layer); evaders past the first P of
                    // the layer stay; every pursuer moves by one pursuer_controller.act() (:238-241)
                    const int from_k = __builtin_amdgcn_ds_bpermute((kidx < P ? kidx : 0) * 4, cur_act);   // lane k holds action k
                    injected = io.inj_eact != nullptr;
                    int pact;
                    if (injected) pact = isP() ? io.inj_eact[env * P + lane] : 4;
                    else pact = (int)__umulhi(philox4x32_10(gid, tick, (uint32_t)lane, TAG_PURSUER_ACT, k0, k1).x, 5u);
                    act = isP() ? pact : (kidx < P ? from_k : 4);
                    injected = true;   // (nothing below draws evader moves)
                } else if constexpr (INJECT) {
                    injected = io.inj_eact != nullptr;
                    if (injected && e_alive) act = io.inj_eact[env * E + kidx];
                }
                if (!injected) {
#if MADRL_ABLATE & 4
                    u32x4 r; r.x = (gid * 2654435761u + tick * 40503u + (uint32_t)kidx * 2246822519u) ^ k0 ^ k1;
#else
                    const u32x4 r = philox4x32_10(gid, tick, (uint32_t)kidx, TAG_EVADER_ACT, k0, k1);
#endif
                    if (!isP()) act = (int)__umulhi(r.x, 5u);  // RandomPolicy.act, Controllers.py:15-16
                }
                // DiscreteAgent.step, DiscreteAgent.py:69-97
                const int dcell = (act == 0 ? -GW : 0) + (act == 1 ? GW : 0) + (act == 2 ? 1 : 0) + (act == 3 ? -1 : 0);
                const bool tflag = (term >> lane) & 1ull;
                const bool in_building = L[cell] != 0u;          // layer 0 is +0.0f only on free in-map cells
                const bool target_free = L[cell + dcell] == 0u;  // building or outside the map otherwise
                const bool newterm = alive && !tflag && in_building;
                if (alive && !tflag && !in_building && target_free) {
                    cell += dcell;
                    x += (act == 1) - (act == 0);
                    y += (act == 2) - (act == 3);
                }
                term |= __ballot(newterm);
                if (alive) atomicAdd(&layer[cell], 1u);  // :244-246
                wave_sync();
                // ---------------------------------------------------- catch resolution (:463-521)
                bool caught = false;
                if (e_alive) {
                    const uint32_t *pc = &L[GSZ];
                    if (d.surround) {
                        const uint32_t n0 = pc[cell - GW], n1 = pc[cell + GW], n2 = pc[cell + 1], n3 = pc[cell - 1];
                        const int cnt = (int)(n0 - 1u < SENT - 1u) + (int)(n1 - 1u < SENT - 1u) +
                                        (int)(n2 - 1u < SENT - 1u) + (int)(n3 - 1u < SENT - 1u);
                        caught = cnt == (int)need_tab[x * S::YS + y];  // need_to_surround :523-540
                    } else {
                        caught = (int)pc[cell] >= d.n_catch;  // :498
                    }
                    if (caught) atomicAdd(&layer[cell], CAUGHT);
                }
                const uint64_t caught_mask = __ballot(caught) >> P;
                gone |= caught_mask;
                wave_sync();
                // ---------------------------------------------------- rewards (:254-262)
                double r = 0.0;
                if (isP()) {
                    const uint32_t *ec = &L[2 * GSZ];
                    bool sur;
                    if (d.surround) {  // a caught evader on one of my four neighbour cells (:489-495)
                        const uint32_t n0 = ec[cell - GW], n1 = ec[cell + GW], n2 = ec[cell + 1], n3 = ec[cell - 1];
                        sur = ((n0 != SENT) & (n0 >= CAUGHT)) | ((n1 != SENT) & (n1 >= CAUGHT)) |
                              ((n2 != SENT) & (n2 >= CAUGHT)) | ((n3 != SENT) & (n3 >= CAUGHT));
                    } else {
                        sur = ec[cell] >= CAUGHT;  // :503-506
                    }
                    double catchr = d.catchr;
                    if constexpr (INJECT) { if (d.catchr_env != nullptr) catchr = sload_f64(d.catchr_env, env); }
                    r = catchr * (double)kpre;
                    r += d.term_pursuit * (sur ? 1.0 : 0.0);
                    r += d.urgency;
                }
                if (d.reward_global) {  // [rewards.mean()] * n_pursuers, numpy summation order
                    double all[P];
#pragma unroll
                    for (int k = 0; k < P; ++k) all[k] = __shfl(r, k);
                    r = np_sum_regs<P>(all) / (double)P;
                }
                tick += 1;
                tstep += 1;
                constexpr uint64_t all_e = (1ull << E) - 1ull;
                if ((gone & all_e) == all_e) done_bits |= 1u;  // :383-389
                if (d.max_steps > 0 && tstep >= d.max_steps) done_bits |= 2u;
                do_reset = d.auto_reset && done_bits != 0;
                rew_out = (float)r;
                n_removed = __popcll(caught_mask);
            }

            // Pipeline hinge: everything above only LOADS from HBM (issued one env ago), everything
            // below only STORES.  Touching the prefetched registers here makes the compiler wait
            // for the next env's record now -- when only loads and the previous env's long-issued
            // stores are outstanding -- instead of at the loop back-edge, where an in-order
            // vmcnt(0) would also wait for this env's observation stores to reach HBM.
            if constexpr (!PIPE) asm volatile("" : "+v"(nxt_rec), "+v"(nxt_act), "+v"(nxt_zm));
            uint32_t zm = cur_zm;  // stale-zero mask of this env: byte k, bit (4 - s) = cell k of slot s

            // One observation pass normally.  On auto-reset two: the reference sequence is step()
            // then reset(), both write the persistent observation buffer and the cells the second
            // write skips keep the first one's values.  `alive` is the PRE-catch set in the first
            // pass: a just-caught evader is still drawn in channel 2 (Q6) and its cell (count +
            // CAUGHT mark) is zeroed with the others.
            const int npass = (MODE == 1 && do_reset) ? 2 : 1;
            for (int pass = 0; pass < npass; ++pass) {
                if (do_reset && pass == npass - 1) {
                    // -------------------------------------------------- reset (:173-207)
                    gone = 0ull;
                    term = 0ull;
                    const KArgsPtr ka = cold_args();
                    const double cw = ka->d.cw_env != nullptr ? sload_f64(ka->d.cw_env, env) : ka->d.cw;
                    const int max_opponents = ka->d.max_opponents;
                    bool inj_map = false, inj_pos = false;
                    if constexpr (MODE == 0) {
                        inj_map = io.inj_map != nullptr;
                        inj_pos = io.inj_pos != nullptr;
                    }
                    if (inj_map) {
                        map_id = __builtin_amdgcn_readfirstlane(io.inj_map[env]);
                    } else if (ka->d.sample_maps) {  // :182-183
                        const u32x4 rm = philox4x32_10(gid, tick, 0u, TAG_RESET_ENV, k0, k1);
                        map_id = (int)__umulhi(rm.x, (uint32_t)ka->d.n_maps);
                    }
                    load_map(map_id);
                    const u32x4 rw = philox4x32_10(gid, tick, 1u, TAG_RESET_ENV, k0, k1);
                    const double sx = u53(rw.x, rw.y) * (1.0 - cw);  // :185-191, float64
                    const double sy = u53(rw.z, rw.w) * (1.0 - cw);
                    const int xlb = (int)(S::XS * sx), xub = (int)(S::XS * (sx + cw));
                    const int ylb = (int)(S::YS * sy), yub = (int)(S::YS * (sy + cw));
                    // random_opponents (train_pursuit, :177-181): n_create <= E evaders this episode, the slots above are not
                    // created and count as gone; an injected position with x < 0 marks a slot that is not created
                    int n_create = E;
                    if (max_opponents > 0 && !inj_pos) {
                        const u32x4 r3 = philox4x32_10(gid, tick, 2u, TAG_RESET_ENV, k0, k1);
                        n_create = min(1 + (int)__umulhi(r3.x, (uint32_t)(max_opponents - 1)), E);
                    }
                    bool exists = false;
                    if (isAgent()) {  // create_agents / feasible_position, agent_utils.py:12-47
                        exists = isP() || eslot < n_create;
                        if (inj_pos) {
                            x = io.inj_pos[(env * A + lane) * 2];
                            y = io.inj_pos[(env * A + lane) * 2 + 1];
                            if (!isP() && x < 0) exists = false;
                        } else {
                            for (uint32_t att = 0; att < 1024u; ++att) {
                                const u32x4 rp = philox4x32_10(gid, tick, (uint32_t)lane, TAG_RESET_POS | (att << 8), k0, k1);
                                x = xlb + (int)__umulhi(rp.x, (uint32_t)(xub - xlb));
                                y = ylb + (int)__umulhi(rp.y, (uint32_t)(yub - ylb));
                                // building cells hold fl32(1/norm) != 0; window cells are inside the map
                                if (L[(x + PAD) * GW + y + PAD] == 0u) break;
                            }
                        }
                        if (exists) {
                            cell = (x + PAD) * GW + y + PAD;
                            atomicAdd(&layer[cell], 1u);  // :201-203
                        } else {
                            x = 0;
                            y = 0;
                        }
                    }
                    gone = __ballot(isE() && !exists) >> P;
                    alive = exists;
                    tick += 1;
                    tstep = 0;
                    wave_sync();
                }
                // ------------------------------------------------------ observations (:418-461)
                // integer counts -> float32 observation values, in place (a wave executes the
                // ds_read of every lane before the ds_write of any lane)
                uint32_t cnt = 0;
                if (alive) cnt = layer[cell] & 0xFFFFu;
                wave_sync();
                if (alive) layer[cell] = L[S::X_VTAB + cnt];
                wave_sync();
#if MADRL_ABLATE & 8
                if (d.n_envs < 0)
#endif
                {
                    // byte offset of this pursuer's window origin in L; the slot constants are byte offsets too, so a cell address is one add
                    int origin = isP() ? ((x - S::OFF + PAD) * GW + (y - S::OFF + PAD)) * 4 : 0;
                    int n_rows = P;
                    if constexpr (CTRL) {
                        // observers (collect_obs walks range(n_pursuers) over evaders_gone, :418-428): the remaining evaders of slots
                        // 0..P-1 in slot order; row k = the k-th of them.  Each tells lane k its window origin through LDS.
                        const uint64_t obsv = ~gone & ((1ull << P) - 1ull);
                        n_rows = __popcll(obsv);
                        const bool is_obs = isE() && eslot < P && ((obsv >> (eslot & 63)) & 1ull);
                        if (is_obs) L[S::X_OBSV + __popcll(obsv & ((1ull << (eslot & 63)) - 1ull))] = (uint32_t)(((x - S::OFF + PAD) * GW + (y - S::OFF + PAD)) * 4);
                        wave_sync();
                        origin = (isP() && lane < n_rows) ? (int)L[S::X_OBSV + lane] : 0;
                    }
                    // SGPR base + one loop-invariant 32-bit VGPR offset (+ immediate) for every store of the row
                    const char *orow_u = (const char *)uniform_ptr(io.obs + env * (int64_t)(P * S::D));
                    // if that base came through v_readfirstlane (a VALU write of an SGPR), a vector-memory instruction may read it as its
                    // scalar address only 5 wait states later; the compiler cannot see the stores inside the asm blocks below
                    asm volatile("s_nop 4" : "+s"(orow_u));
                    const uint32_t voff = ulane * 16u;
                    const char *Lb = reinterpret_cast<const char *>(L);
                    auto cell_at = [&](int off) -> uint32_t { return *reinterpret_cast<const uint32_t *>(Lb + off); };
                    uint32_t acc = 0u;  // the new mask, slot by slot
                    static_for<0, NS>([&](auto sc) {
                        constexpr int s = decltype(sc)::value;
                        const int base = __builtin_amdgcn_ds_bpermute(s_src[s], origin);
                        const uint32_t v0 = cell_at(base + s_cst[s][0]);
                        const uint32_t v1 = cell_at(base + s_cst[s][1]);
                        const uint32_t v2 = cell_at(base + s_cst[s][2]);
                        const uint32_t v3 = cell_at((int)__umul24((uint32_t)base, (uint32_t)s_rel3[s]) + s_cst[s][3]);
                        // flags of the four cells, one per byte (bit 0).  The top byte of a value is 0x00 for +0.0f, 0x3D..0x41 for the
                        // positive observation values and 0xFF for SENT (outside the map).
                        const uint32_t top = __builtin_amdgcn_perm(v1, v0, 0x0C0C0703u) | __builtin_amdgcn_perm(v3, v2, 0x07030C0Cu);
                        const uint32_t out4 = (top >> 7) & 0x01010101u;                    // outside the map: not written
                        const uint32_t nz4 = ((top >> 5) | (top >> 6)) & 0x01010101u;      // value != 0 (meaningful where inside)
                        const uint32_t old4 = (zm >> (NS - 1 - s)) & 0x01010101u;          // holds a value not known to be zero
                        const uint32_t dirty = out4 & old4;                                 // outside AND possibly non-zero: must stay untouched
                        bool valid = (64 * (s + 1) <= S::NQ) ? true : (fresh(lane) + 64 * s < S::NQ);
                        if constexpr (CTRL) {
                            const bool row_live = (s_src[s] >> 2) < n_rows;   // rows of absent observers keep their contents -- and their flags
                            acc = (acc << 1) | (row_live ? ((out4 & old4) | (~out4 & nz4)) : old4);
                            valid = valid && row_live;
                        } else {
                            acc = (acc << 1) | ((out4 & old4) | (~out4 & nz4));
                        }
                        bool clean = dirty == 0u;
#if MADRL_ABLATE & 1
                        valid = d.n_envs < 0;
#endif
#if MADRL_ABLATE & 2
                        clean = true;
#endif
#if MADRL_ABLATE & 32
                        // what a scheme that REMEMBERS small stale values (two bits per cell: 0, 0.1, 0.2, other) and rebuilds them would add to
                        // this pass: about twenty vector instructions per slot (second byte of the four values, two flag planes, byte-to-float
                        // conversions, one fused multiply-add per cell).  With `& 2` (every float4 stored whole) this prices the scheme:
                        // its stores at their best, its instructions at their count (Pursuit fast path, "stale values remembered").
                        {
                            uint32_t burn = acc;
                            asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                                         "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                                         "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n"
                                         "v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_xor_b32 %0, %0, %1"
                                         : "+v"(burn) : "v"(v0));
                            acc = burn;   // (an even number of XORs with the same value: unchanged)
                        }
#endif
                        // An outside cell with a non-zero stale value: leave it alone (Q2), store the inside cells one by one.  Plain
                        // (L2-cached) stores: partial lines must merge in L2 -- nontemporal partial writes cost a read-modify-write
                        // at the memory side (3x slower).  They are issued BEFORE the non-temporal store of the slot's other lanes: the
                        // line is then in L2 when the streaming store arrives and leaves as one write (the other order: 175 us, not 77).
                        const void *sb = orow_u + 4096 * (s / 4);
                        const uint64_t md = __builtin_amdgcn_ballot_w64(valid && !clean);
                        store1x4_masked<1024 * (s % 4)>(sb, voff, v0, v1, v2, v3, md & __builtin_amdgcn_ballot_w64(v0 != SENT),
                                                        md & __builtin_amdgcn_ballot_w64(v1 != SENT), md & __builtin_amdgcn_ballot_w64(v2 != SENT),
                                                        md & __builtin_amdgcn_ballot_w64(v3 != SENT));
                        // One non-temporal float4 unless a cell must stay untouched; nothing if all four are outside and known zero.
                        // Outside cells (SENT = -1 as an integer) are written as the +0.0f they already hold.
                        uint64_t m_nt = __builtin_amdgcn_ballot_w64(valid && clean && out4 != 0x01010101u);
                        // Whole 64-byte chunks: a slot that lies ENTIRELY outside the map and whose cells are known to hold zero normally
                        // stores nothing -- but when another slot of its aligned group of four lanes is written, it is written too (as the
                        // zeros it already holds), so that the group's 64 bytes leave the CU as one full chunk instead of a partial one the
                        // memory side has to merge.  Round 4, mask equilibrium, 65 536 envs: 80.0 -> 71.6 us per launch (groups of two:
                        // 74.0, of eight: 75.8; scripts/zmask_drift.py).  (The two-wavefront kernel, whose 32 x 32 map leaves far fewer cells outside, loses
                        // with the same rule -- 86.5 against 80.4 us per 32 768-env launch -- and does not apply it.)
                        {
                            constexpr uint64_t lo4 = 0x1111111111111111ull;
                            const uint64_t grp = (m_nt | (m_nt >> 1) | (m_nt >> 2) | (m_nt >> 3)) & lo4;   // bit 4g: group g stores something
                            m_nt |= (grp * 0xFull) & __builtin_amdgcn_ballot_w64(valid && clean && out4 == 0x01010101u);
                        }
                        const v4f_t val = {__uint_as_float((uint32_t)max((int)v0, 0)), __uint_as_float((uint32_t)max((int)v1, 0)),
                                           __uint_as_float((uint32_t)max((int)v2, 0)), __uint_as_float((uint32_t)max((int)v3, 0))};
                        store4_nt_masked<1024 * (s % 4)>(sb, voff, val, m_nt);
                    });
                    zm = acc;
                }
                wave_sync();
                if (alive) layer[cell] = 0u;  // restore the count layers for the next pass / env
                wave_sync();
            }
#if MADRL_ABLATE & 16
            if (d.n_envs < 0)
#endif
            if constexpr (MODE == 1) {
                if (isP()) uniform_ptr(io.rew + env * P)[ulane] = rew_out;
                if (fresh(lane) == 0) {
                    io.done[env] = (uint8_t)done_bits;
                    io.removed[env] = n_removed;
                    cold_args()->d.flags[env] = done_flag_word(done_bits);   // (the pointer is not kept in SGPRs across the env loop)
                }
            }
            // ---------------------------------------------------------- registers -> state record
            // one coalesced dword store: lane k writes dword k of the record
            {
                const int myxy = x | (y << 8);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(rec_src0, myxy);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(rec_src1, myxy);
                uint32_t w = (lo & 0xFFFFu) | (hi << 16);
                // the uniform header / bit-set dwords go in with v_writelane (one VALU op each, no lane mask)
                put_lane<0>(w, tick);
                put_lane<1>(w, (uint32_t)tstep);
                put_lane<2>(w, (uint32_t)map_id);
                put_lane<3>(w, 0u);
                put_lane<S::OFF_GONE / 4>(w, (uint32_t)gone);
                if constexpr (S::NGW > 1) put_lane<S::OFF_GONE / 4 + 1>(w, (uint32_t)(gone >> 32));
                put_lane<S::OFF_TERM / 4>(w, (uint32_t)term);
                if constexpr (S::NTW > 1) put_lane<S::OFF_TERM / 4 + 1>(w, (uint32_t)(term >> 32));
                put_zero_from<S::OFF_TERM / 4 + S::NTW, S::REC_DW>(w);  // padding dwords
#if MADRL_ABLATE & 16
                if (d.n_envs < 0)
#endif
                if (fresh(lane) < S::REC_DW)
                    uniform_ptr(reinterpret_cast<uint32_t *>(d.state + env * (int64_t)S::REC_BYTES))[ulane] = w;
#if MADRL_ABLATE & 16
                if (d.n_envs < 0)
#endif
                uniform_ptr(d.zmask + env * 64)[ulane] = zm;
            }
        }
        if constexpr (PIPE) {
            prefetch_wait<VM_PER_ENV>(pf_rec, pf_act, pf_zm);
            nxt_rec = (fresh(lane) < S::REC_DW) ? pf_rec : 0u;
            nxt_act = (int)pf_act;  // lanes that are not pursuers never read it
            nxt_zm = pf_zm;
        }
        cur_rec = nxt_rec;
        cur_act = nxt_act;
        cur_zm = nxt_zm;
    }
}

}  // namespace pw
}  // namespace madrl
