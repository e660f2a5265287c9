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

#include "common.hpp"

namespace madrl {
namespace pw {

constexpr uint32_t SENT = 0xFFFFFFFFu;   // layer 1/2 outside the map: element is not stored
constexpr uint32_t CAUGHT = 0x10000u;    // added to an evader-count cell by a caught evader

struct WaveDev {
    int32_t n_catch, surround, reward_global, sample_maps, n_maps, max_steps, auto_reset;
    int32_t rec_bytes, off_gone, off_term, ngw, ntw;
    int32_t fmap_stride;  // dwords per map entry in fmaps
    uint32_t k0, k1, gid_base;
    double catchr, term_pursuit, urgency, cw;
    int64_t n_envs;
    const uint32_t *fmaps;   // per map: padded float layer [GSZ] then need_to_surround [XS*YS] as u32
    const float *vtab;       // fl32(k / layer_norm), k = 0..255
    const uint32_t *codes;   // D entries: bit31 = relative to window origin, low bits = dword offset
    uint8_t *state;
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
};

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
    static constexpr int X_FILL = 3 * GSZ;                       // extras after the layers
    static constexpr int X_SKIP = 3 * GSZ + 1;
    static constexpr int X_ID = 3 * GSZ + 2;                     // P id values
    static constexpr int X_VTAB = (X_ID + P + 3) / 4 * 4;        // 72 count values
    static constexpr int NVT = 72;
    static constexpr int X_NEED = X_VTAB + NVT;                  // XS*YS bytes, as dwords
    static constexpr int LDS_DWORDS = X_NEED + (XS * YS + 3) / 4;
    static_assert(A <= 64, "one wavefront per env: n_pursuers + n_evaders must fit 64 lanes");
    static_assert(R % 2 == 1, "odd obs_range only (even ranges run on the generic kernel)");
    static_assert(D % 4 == 0, "observation row must be a whole number of float4");
    static_assert(LDS_DWORDS * 4 <= 64 * 1024, "LDS budget");
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

template <class S>
__global__ __launch_bounds__(64) void pursuit_wave_kernel(const WaveDev d, const WaveIO io, const int mode) {
    constexpr int P = S::P, E = S::E, A = S::A, GW = S::GW, PAD = S::PAD, GSZ = S::GSZ, NS = S::NS;
    __shared__ __attribute__((aligned(16))) uint32_t L[S::LDS_DWORDS];
    const int lane = threadIdx.x;
    const bool is_p = lane < P;
    const bool is_agent = lane < A;
    const int eslot = lane - P;

    // ---------------------------------------------------------------- once per workgroup
    for (int k = lane; k < 2 * GSZ; k += 64) {  // count layers: 0 inside the map, SENT outside
        const int c = k % GSZ;
        const int gx = c / GW - PAD, gy = c % GW - PAD;
        L[GSZ + k] = (gx >= 0 && gx < S::XS && gy >= 0 && gy < S::YS) ? 0u : SENT;
    }
    if (lane == 0) {
        L[S::X_FILL] = d.fmaps[0];  // a corner of the padded map layer is always outside the map
        L[S::X_SKIP] = SENT;
    }
    if (lane < P) L[S::X_ID + lane] = __float_as_uint((float)((double)lane / (double)P));  // :440-445
    for (int k = lane; k < S::NVT; k += 64) L[S::X_VTAB + k] = __float_as_uint(d.vtab[k]);
    // observation slot constants
    int s_cst[NS][4];
    int s_rel3[NS];   // 1: element 3 is window-relative, 0: absolute (id / skip / fill cell)
    int s_src[NS];    // ds_bpermute byte address of the owning pursuer's lane
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int q = lane + 64 * s;
        const int pidx = q / S::DV, f = q % S::DV;
        s_src[s] = (q < S::NQ ? pidx : 0) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // slots past the end of the row read harmless cells and are never stored
            const uint32_t c = (q < S::NQ) ? d.codes[4 * f + k] : (k == 3 ? (uint32_t)S::X_SKIP : 0u);
            int cst = (int)(c & 0x7FFFFFFFu);
            if (cst >= S::X_ID && cst < S::X_ID + P) cst = S::X_ID + pidx;
            s_cst[s][k] = cst;
            if (k == 3) s_rel3[s] = (int)(c >> 31);
        }
    }
    int cached_map = -1;
    __syncthreads();

    auto load_map = [&](int map_id) {
        if (cached_map == map_id) return;
        const uint32_t *src = d.fmaps + (int64_t)map_id * d.fmap_stride;
        for (int k = lane; k < GSZ; k += 64) L[k] = src[k];
        for (int k = lane; k < (S::XS * S::YS + 3) / 4; k += 64) L[S::X_NEED + k] = src[GSZ + k];
        cached_map = map_id;
        __syncthreads();
    };
    const uint8_t *need_tab = reinterpret_cast<const uint8_t *>(&L[S::X_NEED]);

    for (int64_t env = blockIdx.x; env < d.n_envs; env += gridDim.x) {
        if (mode == 0 && io.mask != nullptr && io.mask[env] == 0) continue;
        uint8_t *rec = d.state + env * (int64_t)d.rec_bytes;
        // ------------------------------------------------------------ state record -> registers
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(rec);
        uint32_t tick = __builtin_amdgcn_readfirstlane(hdr[0]);
        int32_t tstep = (int32_t)__builtin_amdgcn_readfirstlane(hdr[1]);
        int32_t map_id = (int32_t)__builtin_amdgcn_readfirstlane(hdr[2]);
        int x = 0, y = 0;
        if (is_agent) {
            const uint32_t xy = reinterpret_cast<const uint16_t *>(rec + 16)[lane];
            x = (int)(xy & 0xFF);
            y = (int)(xy >> 8);
        }
        uint64_t gone = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(rec + d.off_gone)[0]);
        if (d.ngw > 1)
            gone |= (uint64_t)__builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(rec + d.off_gone)[1]) << 32;
        uint64_t term = __builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(rec + d.off_term)[0]);
        if (d.ntw > 1)
            term |= (uint64_t)__builtin_amdgcn_readfirstlane(reinterpret_cast<const uint32_t *>(rec + d.off_term)[1]) << 32;
        const uint32_t gid = d.gid_base + (uint32_t)env;
        bool do_reset = (mode == 0);
        uint32_t done_bits = 0;
        bool alive = is_p || (is_agent && !((gone >> eslot) & 1ull));
        uint32_t *layer = &L[is_p ? GSZ : 2 * GSZ];  // this lane's count layer
        int cell = (x + PAD) * GW + y + PAD;

        // ------------------------------------------------------------ observations (:418-461)
        auto write_obs = [&]() {
            // integer counts -> float32 observation values, in place (all reads precede all writes:
            // one wave executes the ds_read for every lane before the ds_write)
            uint32_t cnt = 0;
            if (alive) cnt = layer[cell] & 0xFFFFu;
            __syncthreads();
            if (alive) layer[cell] = L[S::X_VTAB + cnt];
            __syncthreads();
            const int origin = is_p ? (x - S::OFF + PAD) * GW + (y - S::OFF + PAD) : 0;
            float4 *orow = reinterpret_cast<float4 *>(io.obs + env * (int64_t)(P * S::D));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int q = lane + 64 * s;
                const int base = __builtin_amdgcn_ds_bpermute(s_src[s], origin);
                const uint32_t v0 = L[base + s_cst[s][0]];
                const uint32_t v1 = L[base + s_cst[s][1]];
                const uint32_t v2 = L[base + s_cst[s][2]];
                const uint32_t v3 = L[(int)__umul24((uint32_t)base, (uint32_t)s_rel3[s]) + s_cst[s][3]];
                if (q < S::NQ) {
                    if ((v0 != SENT) & (v1 != SENT) & (v2 != SENT) & (v3 != SENT)) {
                        orow[q] = make_float4(__uint_as_float(v0), __uint_as_float(v1), __uint_as_float(v2),
                                              __uint_as_float(v3));
                    } else {  // some cell is outside the map in a count layer: leave it stale (Q2)
                        float *o = reinterpret_cast<float *>(orow + q);
                        if (v0 != SENT) o[0] = __uint_as_float(v0);
                        if (v1 != SENT) o[1] = __uint_as_float(v1);
                        if (v2 != SENT) o[2] = __uint_as_float(v2);
                        if (v3 != SENT) o[3] = __uint_as_float(v3);
                    }
                }
            }
            __syncthreads();
            if (alive) layer[cell] = 0u;  // restore the layers for the next env
            // an evader caught this step is no longer `alive` but still owns a mark + count
        };

        load_map(map_id);

        if (mode == 1) {
            // -------------------------------------------------------- pre-move reward (:359-381)
            if (is_agent && !is_p && alive) atomicAdd(&layer[cell], 1u);
            __syncthreads();
            int kpre = 0;
            if (is_p) {
                const int xm = max(x - 1, 0), xp = min(x + 1, S::XS - 1);
                const int ym = max(y - 1, 0), yp = min(y + 1, S::YS - 1);
                const uint32_t *ec = &L[2 * GSZ];
                kpre = (int)(ec[(xm + PAD) * GW + y + PAD] + ec[(xp + PAD) * GW + y + PAD] +
                             ec[(x + PAD) * GW + yp + PAD] + ec[(x + PAD) * GW + ym + PAD]);
            }
            __syncthreads();
            if (is_agent && !is_p && alive) atomicSub(&layer[cell], 1u);
            // -------------------------------------------------------- moves (:229-241)
            bool newterm = false;
            if (alive) {
                int act;
                if (is_p) {
                    act = io.actions[env * P + lane];
                } else {
                    // index in the evader layer = alive evaders in lower slots
                    const uint64_t below = (~gone) & ((1ull << eslot) - 1ull);
                    const int k = __popcll(below);
                    if (io.inj_eact != nullptr) {
                        act = io.inj_eact[env * E + k];
                    } else {
                        const u32x4 r = philox4x32_10(gid, tick, (uint32_t)k, TAG_EVADER_ACT, d.k0, d.k1);
                        act = (int)__umulhi(r.x, 5u);
                    }
                }
                const bool tflag = (term >> lane) & 1ull;
                if (!tflag) {  // DiscreteAgent.step, DiscreteAgent.py:69-97
                    if (L[cell] != 0u) {
                        newterm = true;  // standing in a building
                    } else {
                        int nx = x, ny = y;
                        if (act == 0) nx = x - 1;
                        else if (act == 1) nx = x + 1;
                        else if (act == 2) ny = y + 1;
                        else if (act == 3) ny = y - 1;
                        const int ncell = (nx + PAD) * GW + ny + PAD;
                        if (L[ncell] == 0u) {  // layer 0 is +0.0f only on free in-map cells
                            x = nx;
                            y = ny;
                            cell = ncell;
                        }
                    }
                }
                atomicAdd(&layer[cell], 1u);  // :244-246
            }
            term |= __ballot(newterm);
            __syncthreads();
            // -------------------------------------------------------- catch resolution (:463-521)
            bool caught = false;
            if (is_agent && !is_p && alive) {
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
            const uint64_t caught_mask = __ballot(caught) >> P;
            gone |= caught_mask;
            const int n_removed = __popcll(caught_mask);
            __syncthreads();
            // -------------------------------------------------------- rewards (:254-262)
            double r = 0.0;
            if (is_p) {
                const uint32_t *ec = &L[2 * GSZ];
                bool sur;
                if (d.surround) {
                    const uint32_t n0 = ec[cell - GW], n1 = ec[cell + GW], n2 = ec[cell + 1], n3 = ec[cell - 1];
                    sur = ((n0 != SENT) & (n0 >= CAUGHT)) | ((n1 != SENT) & (n1 >= CAUGHT)) |
                          ((n2 != SENT) & (n2 >= CAUGHT)) | ((n3 != SENT) & (n3 >= CAUGHT));
                } else {
                    sur = ec[cell] >= CAUGHT;
                }
                r = d.catchr * (double)kpre;
                r += d.term_pursuit * (sur ? 1.0 : 0.0);
                r += d.urgency;
            }
            if (d.reward_global) {
                double all[P];
#pragma unroll
                for (int k = 0; k < P; ++k) all[k] = __shfl(r, k);
                r = np_sum_regs<P>(all) / (double)P;
            }
            if (is_p) io.rew[env * P + lane] = (float)r;
            tick += 1;
            tstep += 1;
            const uint64_t all_e = (E >= 64) ? ~0ull : ((1ull << E) - 1ull);
            if ((gone & all_e) == all_e) done_bits |= 1u;
            if (d.max_steps > 0 && tstep >= d.max_steps) done_bits |= 2u;
            if (lane == 0) {
                io.done[env] = (uint8_t)done_bits;
                io.removed[env] = n_removed;
            }
            do_reset = d.auto_reset && done_bits != 0;
            if (do_reset) {
                // the final step's observation lands in the persistent buffer first; write_obs also
                // zeroes the cell of every agent counted in this step (`alive` still includes the
                // evaders caught just now, which is what clears their count and CAUGHT mark)
                write_obs();
                __syncthreads();
            }
        }

        if (do_reset) {
            // ---------------------------------------------------------- reset (:173-207)
            gone = 0ull;
            term = 0ull;
            if (io.inj_map != nullptr && mode == 0) {
                map_id = __builtin_amdgcn_readfirstlane(io.inj_map[env]);
            } else if (d.sample_maps) {
                const u32x4 rm = philox4x32_10(gid, tick, 0u, TAG_RESET_ENV, d.k0, d.k1);
                map_id = (int)__umulhi(rm.x, (uint32_t)d.n_maps);
            }
            load_map(map_id);
            const u32x4 rw = philox4x32_10(gid, tick, 1u, TAG_RESET_ENV, d.k0, d.k1);
            const double sx = u53(rw.x, rw.y) * (1.0 - d.cw);
            const double sy = u53(rw.z, rw.w) * (1.0 - d.cw);
            const int xlb = (int)(S::XS * sx), xub = (int)(S::XS * (sx + d.cw));
            const int ylb = (int)(S::YS * sy), yub = (int)(S::YS * (sy + d.cw));
            if (is_agent) {
                if (io.inj_pos != nullptr && mode == 0) {
                    x = io.inj_pos[(env * A + lane) * 2];
                    y = io.inj_pos[(env * A + lane) * 2 + 1];
                } else {
                    for (uint32_t att = 0; att < 1024u; ++att) {
                        const u32x4 rp = philox4x32_10(gid, tick, (uint32_t)lane, TAG_RESET_POS | (att << 8), d.k0, d.k1);
                        x = xlb + (int)__umulhi(rp.x, (uint32_t)(xub - xlb));
                        y = ylb + (int)__umulhi(rp.y, (uint32_t)(yub - ylb));
                        // building cells hold fl32(1/norm) != 0; in-window cells are never outside the map
                        if (L[(x + PAD) * GW + y + PAD] == 0u) break;
                    }
                }
                cell = (x + PAD) * GW + y + PAD;
                atomicAdd(&layer[cell], 1u);
            }
            alive = is_agent;
            tick += 1;
            tstep = 0;
            __syncthreads();
        }

        // `alive` is the pre-catch set in step mode: a just-caught evader is still drawn in channel 2
        // of this observation (Q6) and its cell (count + CAUGHT mark) is zeroed with the others
        write_obs();
        // ------------------------------------------------------------ registers -> state record
        if (is_agent) reinterpret_cast<uint16_t *>(rec + 16)[lane] = (uint16_t)(x | (y << 8));
        if (lane == 0) {
            uint32_t *h = reinterpret_cast<uint32_t *>(rec);
            h[0] = tick;
            h[1] = (uint32_t)tstep;
            h[2] = (uint32_t)map_id;
            reinterpret_cast<uint32_t *>(rec + d.off_gone)[0] = (uint32_t)gone;
            if (d.ngw > 1) reinterpret_cast<uint32_t *>(rec + d.off_gone)[1] = (uint32_t)(gone >> 32);
            reinterpret_cast<uint32_t *>(rec + d.off_term)[0] = (uint32_t)term;
            if (d.ntw > 1) reinterpret_cast<uint32_t *>(rec + d.off_term)[1] = (uint32_t)(term >> 32);
        }
    }
}

}  // namespace pw
}  // namespace madrl
