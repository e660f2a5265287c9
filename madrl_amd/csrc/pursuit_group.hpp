// pursuit_group.hpp -- compile-time-specialised PursuitEvade kernel for shapes with MORE THAN 64 AGENTS:
// one workgroup of NW wavefronts = one env (BASELINE configs[4]: 32x32, 16 pursuers / 60 evaders).
//
// Same scheme as pursuit_wave.hpp (read that header first): one lane per agent, padded dword layers in LDS,
// counts converted in place to their float32 observation values, float4 observation slots with non-temporal
// stores, one coalesced dword load / store of the packed state record, prefetch one env ahead with the pipeline
// hinge.  What changes when the agents of an env no longer fit one wavefront:
//   * thread t = agent t (pursuers first): the agents occupy the first NAW = ceil(A / 64) wavefronts; the
//     observation row is dealt over ALL NW wavefronts (slot q -> thread q % (64 NW)), so a wavefront without
//     agents still carries its share of the row;
//   * phases are separated by `s_waitcnt lgkmcnt(0); s_barrier` (LDS traffic only -- unlike __syncthreads()
//     no vmcnt(0): a wavefront never waits for its observation stores to reach HBM);
//   * wave-wide ballots (caught evaders, evaders not created by a reset) are combined through two LDS words
//     per wavefront; the alive mask of the <= 64 evaders stays ONE 64-bit scalar in every wavefront, the
//     terminal flags are one 64-bit scalar per wavefront for its own lanes;
//   * window origins come from LDS (written by the pursuer lanes) instead of ds_bpermute, since the pursuers
//     live in wavefront 0 only;
//   * every wavefront loads the whole record (dword k in lane k) and stores the dwords it owns.
// Requirements: n_pursuers <= 64, n_evaders <= 64, record <= 64 dwords, odd obs_range, row length % 4 == 0.
//
// LONG ROWS (round 6; the authors' own training shapes, runners/old/rllab/pursuit.sh:1 -- 30 pursuers, obs_range 11: 2 730 float4 per
// env, 22 slots per thread): with more than 8 slots per thread the per-slot constants no longer live in registers (6 VGPRs per slot).
// Such a shape is TABLED: slot q = tid + NT s still belongs to thread tid (every store instruction writes NT consecutive float4, whole
// 64-byte chunks), but its constants come from a table in LDS indexed by the float4's position f = q mod DV inside its row -- four
// 15-bit dword offsets relative to the window origin, packed in two dwords, identical for every pursuer -- and (pursuer, f) advance
// from slot to slot by constants.  The stale-zero mask grows to MWORDS = ceil(NS / 8) dwords per thread (one bit per slot in each byte).
#pragma once

#include "pursuit_wave.hpp"

namespace madrl {
namespace pw {

// resident wavefronts per SIMD the group kernel's registers are allocated for
#ifndef MADRL_PG_WAVES
#define MADRL_PG_WAVES 4
#endif
#ifndef MADRL_PG_UNROLL
#define MADRL_PG_UNROLL 2   // LONG ROWS: slots of one mask word in flight together (their LDS reads issued side by side)
#endif

template <int XS_, int YS_, int P_, int E_, int R_, int FLATTEN_, int NW_>
struct GShape {
    static constexpr int XS = XS_, YS = YS_, P = P_, E = E_, A = P_ + E_, R = R_, FLATTEN = FLATTEN_, NW = NW_;
    static constexpr int NT = 64 * NW;
    static constexpr int NAW = (A + 63) / 64;                    // wavefronts that hold agents
    static constexpr int OFF = (R - 1) / 2;
    static constexpr int PAD = OFF > 1 ? OFF : 1;
    static constexpr int GW = YS + 2 * PAD;
    static constexpr int GH = XS + 2 * PAD;
    static constexpr int GSZ = (GH * GW + 3) / 4 * 4;
    static constexpr int D = FLATTEN ? 3 * R * R + 1 : 4 * R * R;
    static constexpr int DV = D / 4;
    static constexpr int NQ = P * DV;
    static constexpr int NS = (NQ + NT - 1) / NT;                // float4 slots per thread
    static constexpr int MWORDS = (NS + 7) / 8;                  // stale-zero mask dwords per thread: one bit per slot in each byte, 8 slots per dword
    static constexpr bool TABLED = NS > 8;                       // slot constants from an LDS table indexed by the position in the row (see above)
    static constexpr int X_FILL = 3 * GSZ;
    static constexpr int X_SKIP = 3 * GSZ + 1;
    static constexpr int X_ID = 3 * GSZ + 2;
    static constexpr int X_VTAB = (X_ID + P + 3) / 4 * 4;
    static constexpr int NVT = 72;
    static constexpr int X_NEED = X_VTAB + NVT;
    static constexpr int X_ORG = X_NEED + (XS * YS + 3) / 4;     // P window origins
    static constexpr int X_XCH = (X_ORG + P + 3) / 4 * 4;        // 2 dwords per wavefront: ballot exchange
    static constexpr int X_TAB = (X_XCH + 2 * NW + 3) / 4 * 4;   // TABLED: DV entries of two dwords (8-byte aligned)
    static constexpr int LDS_DWORDS = X_TAB + (TABLED ? 2 * DV : 0);
    static constexpr int NGW = (E + 31) / 32 > 0 ? (E + 31) / 32 : 1;
    static constexpr int NTW = (A + 31) / 32;
    static constexpr int OFF_GONE = (16 + 2 * A + 3) / 4 * 4;
    static constexpr int OFF_TERM = OFF_GONE + 4 * NGW;
    static constexpr int REC_BYTES = (OFF_TERM + 4 * NTW + 15) / 16 * 16;
    static constexpr int REC_DW = REC_BYTES / 4;
    static_assert(NAW <= NW, "not enough wavefronts for the agents");
    static_assert(P <= 64 && E <= 64, "pursuers must fit wavefront 0, the evader alive mask one 64-bit scalar");
    static_assert(REC_DW <= 64, "the whole record must fit one dword per lane");
    static_assert(R % 2 == 1, "odd obs_range only");
    static_assert(D % 4 == 0, "observation row must be a whole number of float4");
    static_assert(LDS_DWORDS * 4 <= 64 * 1024, "LDS budget");
    // resident wavefronts per SIMD: what the registers are allocated for, capped by what the LDS of a CU (160 KB) admits
    static constexpr int OCC_LDS = (160 * 1024 / (LDS_DWORDS * 4)) * NW / 4;
    static constexpr int OCC = OCC_LDS < MADRL_PG_WAVES ? (OCC_LDS < 1 ? 1 : OCC_LDS) : MADRL_PG_WAVES;
    static_assert(MWORDS <= 4, "at most 32 float4 slots per thread");
    static_assert(!TABLED || 3 * GSZ + 2 + P < 32768, "TABLED: dword offsets of the layers must fit 15 bits");
};

// LDS-only workgroup barrier: the DS queue of this wavefront is drained, global stores stay in flight.
__device__ __forceinline__ void group_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class S, int MODE, bool INJECT>
__global__ __launch_bounds__(S::NT) __attribute__((amdgpu_waves_per_eu(S::OCC, S::OCC))) void pursuit_group_kernel(const WaveDev d, const WaveIO io) {
    constexpr int P = S::P, E = S::E, A = S::A, GW = S::GW, PAD = S::PAD, GSZ = S::GSZ, NS = S::NS, NT = S::NT;
    __shared__ __attribute__((aligned(16))) uint32_t L[S::LDS_DWORDS];
    const int tid = threadIdx.x;            // = agent index for tid < A
    const int lane = tid & 63;
    const uint32_t utid = threadIdx.x, ulane = utid & 63u;  // unsigned 32-bit indices: SGPR base + VGPR offset addressing
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_p = tid < P;
    const int eslot = tid - P;

    // ---------------------------------------------------------------- once per workgroup
    for (int k = tid; k < GSZ; k += NT) {  // count layers: 0 inside the map, SENT outside
        const uint32_t v = d.cnt_tmpl[k];
        L[GSZ + k] = v;
        L[2 * GSZ + k] = v;
    }
    if (tid == 0) {
        L[S::X_FILL] = d.fmaps[0];
        L[S::X_SKIP] = SENT;
    }
    if (tid < P) L[S::X_ID + tid] = __float_as_uint((float)((double)tid / (double)P));
    for (int k = tid; k < S::NVT; k += NT) L[S::X_VTAB + k] = __float_as_uint(d.vtab[k]);
    constexpr int NSR = S::TABLED ? 1 : NS;   // slots whose constants live in registers
    int s_cst[NSR][4];
    int s_rel3[NSR];
    int s_org[NSR];    // LDS index of the owning pursuer's window origin
    if constexpr (S::TABLED) {
        for (int k = tid; k < 2 * S::DV; k += NT) L[S::X_TAB + k] = d.slot_tab[k];   // host-built (pursuit.hip): [DV][2] packed offsets
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint32_t *t = d.slot_tab + s * 6 * NT + tid;  // host-built (pursuit.hip), see WaveDev
#pragma unroll
            for (int k = 0; k < 4; ++k) s_cst[s][k] = (int)t[NT * k];
            s_rel3[s] = (int)t[NT * 4];
            s_org[s] = S::X_ORG + (int)t[NT * 5];
        }
    }
    const int q0_p = tid / S::DV, q0_f = tid % S::DV;   // TABLED: (pursuer, float4 position in its row) of this thread's slot 0
    // map 0 is staged with the tables above (pursuit_wave.hpp); the group_sync before the env loop publishes it
    for (int k = tid; k < GSZ; k += NT) L[k] = d.fmaps[k];
    for (int k = tid; k < (S::XS * S::YS + 3) / 4; k += NT) L[S::X_NEED + k] = d.fmaps[GSZ + k];
    int cached_map = 0;
    const uint8_t *need_tab = reinterpret_cast<const uint8_t *>(&L[S::X_NEED]);
    uint32_t *const layer = &L[is_p ? GSZ : 2 * GSZ];
    // record dword k (held by lane k of every wavefront) is STORED by one wavefront: the agent-pair dwords by the
    // wavefront that holds the two agents, the terminal words by the wavefront whose lanes they describe, the
    // rest (header, alive mask, padding) by wavefront 0
    constexpr int XY_END = 4 + (A + 1) / 2;
    const bool own_xy = lane >= 4 && lane < XY_END && (lane - 4) / 32 == wv;
    const bool own_term = lane >= S::OFF_TERM / 4 && lane < S::OFF_TERM / 4 + S::NTW && (lane - S::OFF_TERM / 4) / 2 == wv;
    const bool own_rest = wv == 0 && lane < S::REC_DW && !(lane >= 4 && lane < XY_END) &&
                          !(lane >= S::OFF_TERM / 4 && lane < S::OFF_TERM / 4 + S::NTW);
    const bool own_dw = own_xy || own_term || own_rest;
    const int rec_src0 = (own_xy ? 2 * (lane - 4) - 64 * wv : 0) * 4, rec_src1 = rec_src0 + 4;

    auto isP = [&]() { return fresh(tid) < P; };
    auto isE = [&]() { return (unsigned)(fresh(tid) - P) < (unsigned)E; };
    auto isAgent = [&]() { return fresh(tid) < A; };
    auto fetch_rec = [&](int64_t env) -> uint32_t {
        return (fresh(lane) < S::REC_DW) ? uniform_ptr(reinterpret_cast<const uint32_t *>(d.state + env * (int64_t)S::REC_BYTES))[ulane] : 0u;
    };
    auto fetch_act = [&](int64_t env) -> int {
        if constexpr (MODE == 1) return isP() ? uniform_ptr(io.actions + env * P)[utid] : 4;
        else return 4;
    };
    // evader-slot mask of a wave-wide predicate, combined over the wavefronts that hold agents
    auto evader_mask = [&](bool pred) -> uint64_t {
        const uint64_t b = __ballot(pred);
        if constexpr (S::NAW == 1) {
            return b >> P;
        } else {
            const uint64_t mine = (wv == 0) ? (b >> P) : ((wv == 1) ? (b << (64 - P)) : 0ull);
            if (lane == 0) {
                L[S::X_XCH + 2 * wv] = (uint32_t)mine;
                L[S::X_XCH + 2 * wv + 1] = (uint32_t)(mine >> 32);
            }
            group_sync();
            const uint32_t a0 = L[S::X_XCH], a1 = L[S::X_XCH + 1], b0 = L[S::X_XCH + 2], b1 = L[S::X_XCH + 3];
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a0 | b0));
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(a1 | b1));
            return ((uint64_t)hi << 32) | lo;
        }
    };
    constexpr int MW = S::MWORDS;
    auto fetch_zm = [&](int64_t env, uint32_t (&zmw)[MW]) {   // stale-zero masks, pursuit_wave.hpp: [env][MWORDS][NT]
#pragma unroll
        for (int w = 0; w < MW; ++w) zmw[w] = uniform_ptr(d.zmask + (env * MW + w) * NT)[utid];
    };
    uint32_t cur_rec = 0, cur_zm[MW];
#pragma unroll
    for (int w = 0; w < MW; ++w) cur_zm[w] = 0xFFFFFFFFu;
    int cur_act = 4;
    // env indices are 32-bit (the fast path is not taken for n_envs >= 2^31 - 2^20), byte offsets 64-bit
    const int n_envs = (int)d.n_envs, stride = (int)gridDim.x;
    auto phys = [&](int e) -> int64_t { return (int64_t)(d.reverse ? n_envs - 1 - e : e); };
    if ((int)blockIdx.x < n_envs) {
        cur_rec = fetch_rec(phys(blockIdx.x));
        cur_act = fetch_act(phys(blockIdx.x));
        fetch_zm(phys(blockIdx.x), cur_zm);
    }
    asm volatile("" : "+v"(cur_rec), "+v"(cur_act));
#pragma unroll
    for (int w = 0; w < MW; ++w) asm volatile("" : "+v"(cur_zm[w]));
    group_sync();

    for (int e = blockIdx.x; e < n_envs; e += stride) {
        const int64_t env = phys(e);
        const bool has_next = e + stride < n_envs;
        const int64_t nenv = phys(has_next ? e + stride : e);
        uint32_t nxt_rec = 0, nxt_zm[MW];
#pragma unroll
        for (int w = 0; w < MW; ++w) nxt_zm[w] = 0xFFFFFFFFu;
        int nxt_act = 4;
        if (has_next) {
            nxt_rec = fetch_rec(nenv);
            nxt_act = fetch_act(nenv);
            fetch_zm(nenv, nxt_zm);
        }
        bool skip = false;
        if constexpr (MODE == 0) skip = (io.mask != nullptr && io.mask[env] == 0);
        if (!skip) {
            // -------------------------------------------------------- unpack the record
            uint32_t tick = __builtin_amdgcn_readlane(cur_rec, 0);
            int32_t tstep = (int32_t)__builtin_amdgcn_readlane(cur_rec, 1);
            int32_t map_id = (int32_t)__builtin_amdgcn_readlane(cur_rec, 2);
            const uint32_t xyw = (uint32_t)__shfl((int)cur_rec, (4 + (tid >> 1)) & 63);
            const uint32_t xy = (tid & 1) ? (xyw >> 16) : (xyw & 0xFFFFu);
            int x = (int)(xy & 0xFF), y = (int)(xy >> 8);
            if (!isAgent()) x = y = 0;  // threads past the last agent keep a harmless in-map cell
            uint64_t gone = (uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_GONE / 4);  // (uint32_t): readlane returns int, bit 31 must not sign-extend
            if constexpr (S::NGW > 1) gone |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane(cur_rec, S::OFF_GONE / 4 + 1) << 32;
            // terminal flags of THIS wavefront's lanes (record words 2 wv, 2 wv + 1)
            uint64_t term = 0ull;
            {
                const uint32_t tw = (uint32_t)__shfl((int)cur_rec, (S::OFF_TERM / 4 + 2 * wv + (lane & 1)) & 63);
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane(tw, 0), t1 = (uint32_t)__builtin_amdgcn_readlane(tw, 1);
                if (2 * wv < S::NTW) term = t0;
                if (2 * wv + 1 < S::NTW) term |= (uint64_t)t1 << 32;
            }
            const uint32_t gid = d.gid_base + (uint32_t)env;
            const uint32_t k0 = fresh_s(d.k0), k1 = fresh_s(d.k1);
            bool do_reset = (MODE == 0);
            uint32_t done_bits = 0;
            float rew_out = 0.0f;
            int n_removed = 0;
            bool alive = isP() || (isE() && !((gone >> (eslot & 63)) & 1ull));
            int cell = (x + PAD) * GW + y + PAD;

            auto load_map = [&](int mid) {
                if (cached_map == mid) return;
                const KArgsPtr ka = cold_args();  // rare-path launch parameters come from the kernarg segment (pursuit_wave.hpp)
                const uint32_t *src = ka->d.fmaps + (int64_t)mid * ka->d.fmap_stride;
                for (int k = tid; k < GSZ; k += NT) L[k] = src[k];
                for (int k = tid; k < (S::XS * S::YS + 3) / 4; k += NT) L[S::X_NEED + k] = src[GSZ + k];
                cached_map = mid;
                group_sync();
            };
            load_map(map_id);

            if constexpr (MODE == 1) {
                const bool e_alive = alive && !isP();
                // ---------------------------------------------------- pre-move reward (:359-381)
                if (e_alive) atomicAdd(&layer[cell], 1u);
                group_sync();
                int kpre = 0;
                if (isP()) {
                    const int dxm = (x > 0) ? GW : 0, dxp = (x < S::XS - 1) ? GW : 0;
                    const int dym = (y > 0) ? 1 : 0, dyp = (y < S::YS - 1) ? 1 : 0;
                    const uint32_t *ec = &L[2 * GSZ];
                    kpre = (int)(ec[cell - dxm] + ec[cell + dxp] + ec[cell + dyp] + ec[cell - dym]);
                }
                group_sync();
                if (e_alive) atomicSub(&layer[cell], 1u);
                // ---------------------------------------------------- moves (:229-241)
                const int kidx = __popcll((~gone) & ((1ull << (eslot & 63)) - 1ull));
                int act = cur_act;
                bool injected = false;
                if constexpr (INJECT) {   // the FLEX instantiation, see pursuit_wave.hpp
                    injected = io.inj_eact != nullptr;
                    if (injected && e_alive) act = io.inj_eact[env * E + kidx];
                }
                if (!injected) {
                    const u32x4 r = philox4x32_10(gid, tick, (uint32_t)kidx, TAG_EVADER_ACT, k0, k1);
                    if (!isP()) act = (int)__umulhi(r.x, 5u);
                }
                const int dcell = (act == 0 ? -GW : 0) + (act == 1 ? GW : 0) + (act == 2 ? 1 : 0) + (act == 3 ? -1 : 0);
                const bool tflag = (term >> lane) & 1ull;
                const bool in_building = L[cell] != 0u;
                const bool target_free = L[cell + dcell] == 0u;
                const bool newterm = alive && !tflag && in_building;
                if (alive && !tflag && !in_building && target_free) {
                    cell += dcell;
                    x += (act == 1) - (act == 0);
                    y += (act == 2) - (act == 3);
                }
                term |= __ballot(newterm);
                if (alive) atomicAdd(&layer[cell], 1u);
                group_sync();
                // ---------------------------------------------------- catch resolution (:463-521)
                bool caught = false;
                if (e_alive) {
                    const uint32_t *pc = &L[GSZ];
                    if (d.surround) {
                        const uint32_t n0 = pc[cell - GW], n1 = pc[cell + GW], n2 = pc[cell + 1], n3 = pc[cell - 1];
                        const int cnt = (int)(n0 - 1u < SENT - 1u) + (int)(n1 - 1u < SENT - 1u) +
                                        (int)(n2 - 1u < SENT - 1u) + (int)(n3 - 1u < SENT - 1u);
                        caught = cnt == (int)need_tab[x * S::YS + y];
                    } else {
                        caught = (int)pc[cell] >= d.n_catch;
                    }
                    if (caught) atomicAdd(&layer[cell], CAUGHT);
                }
                const uint64_t caught_mask = evader_mask(caught);  // contains the barrier for the CAUGHT marks when NAW > 1
                gone |= caught_mask;
                if constexpr (S::NAW == 1) group_sync();
                // ---------------------------------------------------- rewards (:254-262)
                double r = 0.0;
                if (isP()) {
                    const uint32_t *ec = &L[2 * GSZ];
                    bool sur;
                    if (d.surround) {
                        const uint32_t n0 = ec[cell - GW], n1 = ec[cell + GW], n2 = ec[cell + 1], n3 = ec[cell - 1];
                        sur = ((n0 != SENT) & (n0 >= CAUGHT)) | ((n1 != SENT) & (n1 >= CAUGHT)) |
                              ((n2 != SENT) & (n2 >= CAUGHT)) | ((n3 != SENT) & (n3 >= CAUGHT));
                    } else {
                        sur = ec[cell] >= CAUGHT;
                    }
                    double catchr = d.catchr;
                    if constexpr (INJECT) { if (d.catchr_env != nullptr) catchr = sload_f64(d.catchr_env, env); }
                    r = catchr * (double)kpre;
                    r += d.term_pursuit * (sur ? 1.0 : 0.0);
                    r += d.urgency;
                }
                if (d.reward_global && wv == 0) {
                    double all[P];
#pragma unroll
                    for (int k = 0; k < P; ++k) all[k] = __shfl(r, k);
                    r = np_sum_regs<P>(all) / (double)P;
                }
                tick += 1;
                tstep += 1;
                constexpr uint64_t all_e = E >= 64 ? ~0ull : ((1ull << (E & 63)) - 1ull);
                if ((gone & all_e) == all_e) done_bits |= 1u;
                if (d.max_steps > 0 && tstep >= d.max_steps) done_bits |= 2u;
                do_reset = d.auto_reset && done_bits != 0;
                rew_out = (float)r;
                n_removed = __popcll(caught_mask);
            }

            asm volatile("" : "+v"(nxt_rec), "+v"(nxt_act));  // pipeline hinge (pursuit_wave.hpp)
            uint32_t zm[MW];
#pragma unroll
            for (int w = 0; w < MW; ++w) {
                asm volatile("" : "+v"(nxt_zm[w]));
                zm[w] = cur_zm[w];
            }

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
                    } else if (ka->d.sample_maps) {
                        const u32x4 rm = philox4x32_10(gid, tick, 0u, TAG_RESET_ENV, k0, k1);
                        map_id = (int)__umulhi(rm.x, (uint32_t)ka->d.n_maps);
                    }
                    load_map(map_id);
                    const u32x4 rw = philox4x32_10(gid, tick, 1u, TAG_RESET_ENV, k0, k1);
                    const double sx = u53(rw.x, rw.y) * (1.0 - cw);
                    const double sy = u53(rw.z, rw.w) * (1.0 - cw);
                    const int xlb = (int)(S::XS * sx), xub = (int)(S::XS * (sx + cw));
                    const int ylb = (int)(S::YS * sy), yub = (int)(S::YS * (sy + cw));
                    int n_create = E;
                    if (max_opponents > 0 && !inj_pos) {
                        const u32x4 r3 = philox4x32_10(gid, tick, 2u, TAG_RESET_ENV, k0, k1);
                        n_create = min(1 + (int)__umulhi(r3.x, (uint32_t)(max_opponents - 1)), E);
                    }
                    bool exists = false;
                    if (isAgent()) {
                        exists = isP() || eslot < n_create;
                        if (inj_pos) {
                            x = io.inj_pos[(env * A + tid) * 2];
                            y = io.inj_pos[(env * A + tid) * 2 + 1];
                            if (!isP() && x < 0) exists = false;
                        } else {
                            for (uint32_t att = 0; att < 1024u; ++att) {
                                const u32x4 rp = philox4x32_10(gid, tick, (uint32_t)tid, TAG_RESET_POS | (att << 8), k0, k1);
                                x = xlb + (int)__umulhi(rp.x, (uint32_t)(xub - xlb));
                                y = ylb + (int)__umulhi(rp.y, (uint32_t)(yub - ylb));
                                if (L[(x + PAD) * GW + y + PAD] == 0u) break;
                            }
                        }
                        if (exists) {
                            cell = (x + PAD) * GW + y + PAD;
                            atomicAdd(&layer[cell], 1u);
                        } else {
                            x = 0;
                            y = 0;
                        }
                    }
                    gone = evader_mask(isE() && !exists);
                    alive = exists;
                    tick += 1;
                    tstep = 0;
                    if constexpr (S::NAW == 1) group_sync();
                }
                // ------------------------------------------------------ observations (:418-461)
                uint32_t cnt = 0;
                if (alive) cnt = layer[cell] & 0xFFFFu;
                group_sync();
                if (alive) layer[cell] = L[S::X_VTAB + cnt];
                // window origin of pursuer tid: a dword index into L (a byte offset for TABLED shapes)
                if (isP()) L[S::X_ORG + tid] = (uint32_t)((x - S::OFF + PAD) * GW + (y - S::OFF + PAD)) * (S::TABLED ? 4u : 1u);
                group_sync();
                {
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    v4f *orow = reinterpret_cast<v4f *>(io.obs + env * (int64_t)(P * S::D));
                    uint32_t acc[MW];
#pragma unroll
                    for (int w = 0; w < MW; ++w) acc[w] = 0u;
                    // what happens to the four cells of one slot (stale-zero mask: see pursuit_wave.hpp).  Slot s = bit (slots of its word - 1 - s % 8)
                    // of every byte of mask word s / 8; `wc` is that word (a compile-time index: the words live in registers), `sh` the bit
                    auto finish_slot = [&](auto wc, int sh, int q, bool valid, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
                        constexpr int w = decltype(wc)::value;
                        const uint32_t top = __builtin_amdgcn_perm(v1, v0, 0x0C0C0703u) | __builtin_amdgcn_perm(v3, v2, 0x07030C0Cu);
                        const uint32_t out4 = (top >> 7) & 0x01010101u;
                        const uint32_t nz4 = ((top >> 5) | (top >> 6)) & 0x01010101u;
                        const uint32_t old4 = (zm[w] >> sh) & 0x01010101u;
                        const uint32_t dirty = out4 & old4;
                        acc[w] = (acc[w] << 1) | (dirty | (~out4 & nz4));
                        if (valid) {
                            if (dirty == 0u) {
                                if (out4 != 0x01010101u) {
                                    const v4f val = {__uint_as_float((uint32_t)max((int)v0, 0)), __uint_as_float((uint32_t)max((int)v1, 0)),
                                                     __uint_as_float((uint32_t)max((int)v2, 0)), __uint_as_float((uint32_t)max((int)v3, 0))};
                                    __builtin_nontemporal_store(val, &orow[q]);
                                }
                            } else {  // an outside cell with a non-zero stale value (Q2): plain stores that merge in L2
                                float *o = reinterpret_cast<float *>(orow + q);
                                if (v0 != SENT) o[0] = __uint_as_float(v0);
                                if (v1 != SENT) o[1] = __uint_as_float(v1);
                                if (v2 != SENT) o[2] = __uint_as_float(v2);
                                if (v3 != SENT) o[3] = __uint_as_float(v3);
                            }
                        }
                    };
                    if constexpr (S::TABLED) {
                        // a ROLLED loop per mask word (fully unrolled, the 22 slots of the authors' shape cost 241 VGPRs: two wavefronts per SIMD)
                        int pq = q0_p, fq = q0_f, q = tid;   // (pursuer, position in its row) and index of the slot, advanced by constants
                        const char *Lb = reinterpret_cast<const char *>(L);
                        auto cell_b = [&](int byte_off) -> uint32_t { return *reinterpret_cast<const uint32_t *>(Lb + byte_off); };
                        static_for<0, MW>([&](auto wc) {
                            constexpr int w = decltype(wc)::value, nsw = (NS - 8 * w) < 8 ? (NS - 8 * w) : 8;
#pragma unroll MADRL_PG_UNROLL
                            for (int i = 0; i < nsw; ++i) {
                                const bool valid = q < S::NQ;
                                const int pp = valid ? pq : 0;                                   // threads past the end of the rows read harmless cells
                                const uint2 t = *reinterpret_cast<const uint2 *>(&L[S::X_TAB + 2 * fq]);
                                const int base = (int)L[S::X_ORG + pp];                          // byte offset of the window origin
                                const uint32_t v0 = cell_b(base + (int)((t.x & 0xFFFFu) << 2));
                                const uint32_t v1 = cell_b(base + (int)((t.x >> 16) << 2));
                                const uint32_t v2 = cell_b(base + (int)((t.y & 0x7FFFu) << 2));
                                // element 3: relative like the others, or absolute (bit 31: the skip cell / the id cells; bit 15: + pursuer)
                                const int b3 = ((int)t.y < 0) ? 0 : base;
                                const int id3 = (t.y & 0x8000u) ? pp : 0;
                                const uint32_t v3 = cell_b(b3 + (int)((((t.y >> 16) & 0x7FFFu) + (uint32_t)id3) << 2));
                                finish_slot(wc, nsw - 1 - i, q, valid, v0, v1, v2, v3);
                                q += NT;
                                fq += NT % S::DV;
                                pq += NT / S::DV;
                                if (fq >= S::DV) { fq -= S::DV; pq += 1; }
                            }
                        });
                    } else {
                        static_for<0, NS>([&](auto sc) {
                            constexpr int s = decltype(sc)::value;
                            const int q = tid + NT * s;
                            const bool valid = (NT * (s + 1) <= S::NQ) ? true : (fresh(tid) + NT * s < S::NQ);
                            const int base = (int)L[s_org[s]];
                            const uint32_t v0 = L[base + s_cst[s][0]];
                            const uint32_t v1 = L[base + s_cst[s][1]];
                            const uint32_t v2 = L[base + s_cst[s][2]];
                            const uint32_t v3 = L[(int)__umul24((uint32_t)base, (uint32_t)s_rel3[s]) + s_cst[s][3]];
                            finish_slot(std::integral_constant<int, 0>{}, NS - 1 - s, q, valid, v0, v1, v2, v3);
                        });
                    }
#pragma unroll
                    for (int w = 0; w < MW; ++w) zm[w] = acc[w];
                }
                group_sync();
                if (alive) layer[cell] = 0u;
                group_sync();
            }
            if constexpr (MODE == 1) {
                if (isP()) uniform_ptr(io.rew + env * P)[utid] = rew_out;
                if (fresh(tid) == 0) {
                    io.done[env] = (uint8_t)done_bits;
                    io.removed[env] = n_removed;
                    cold_args()->d.flags[env] = done_flag_word(done_bits);
                }
            }
            // ---------------------------------------------------------- registers -> state record
            {
                const int myxy = x | (y << 8);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(rec_src0, myxy);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(rec_src1, myxy);
                uint32_t w = (lo & 0xFFFFu) | (hi << 16);
                if (lane == 0) w = tick;
                if (lane == 1) w = (uint32_t)tstep;
                if (lane == 2) w = (uint32_t)map_id;
                if (lane == 3) w = 0u;
                if (lane == S::OFF_GONE / 4) w = (uint32_t)gone;
                if (S::NGW > 1 && lane == S::OFF_GONE / 4 + 1) w = (uint32_t)(gone >> 32);
                if (own_term) w = ((lane - S::OFF_TERM / 4) & 1) ? (uint32_t)(term >> 32) : (uint32_t)term;
                if (lane >= S::OFF_TERM / 4 + S::NTW) w = 0u;  // padding dwords
                if (own_dw) uniform_ptr(reinterpret_cast<uint32_t *>(d.state + env * (int64_t)S::REC_BYTES))[ulane] = w;
#pragma unroll
                for (int w = 0; w < MW; ++w) uniform_ptr(d.zmask + (env * MW + w) * NT)[utid] = zm[w];
            }
        }
        cur_rec = nxt_rec;
        cur_act = nxt_act;
#pragma unroll
        for (int w = 0; w < MW; ++w) cur_zm[w] = nxt_zm[w];
    }
}

}  // namespace pw
}  // namespace madrl
