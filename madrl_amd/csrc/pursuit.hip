// pursuit.hip -- batched PursuitEvade for MI355X (gfx950 / CDNA4).
//
// One workgroup owns one env instance at a time (workgroups stride over envs).  All of an
// env's mutable state -- agent positions, alive/terminal bitmasks, the three observation
// layers (map / pursuer counts / evader counts) as PADDED byte grids -- is staged in LDS;
// HBM sees exactly one packed state record in, one record out, the action row in, and the
// observation / reward / done rows out.  There is no dense contraction anywhere: this kernel
// is HBM-write bound (the observation row is 94 % of the bytes), so the design
// goals are (1) fully coalesced observation stores, (2) few instructions per stored dword,
// (3) enough resident waves (64-thread workgroups, <4 KB LDS) to cover store latency.
//
// Reference semantics (file:line under /root/reference/madrl_environments/pursuit):
//   step order ............... pursuit_evade.py:209-262      (A.1 in SURVEY.md)
//   pre-move proximity reward  pursuit_evade.py:359-381
//   agent motion ............. utils/DiscreteAgent.py:69-97
//   catch resolution ......... pursuit_evade.py:463-521, need_to_surround :523-540
//   observations ............. pursuit_evade.py:418-461 (stale out-of-map cells kept: the
//                              obs buffer is IN/OUT and those cells are simply not stored)
//   reset .................... pursuit_evade.py:173-207, utils/agent_utils.py:12-47
#include "common.hpp"
#include "pursuit_wave.hpp"
#include "pursuit_group.hpp"

#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

using namespace madrl;

// slot-code kinds (one code per element of an agent's observation row, built on the host)
enum : uint32_t { K_GRID = 0, K_ID = 1, K_FILL = 2, K_SKIP = 3 };

constexpr uint32_t PAD_MAP = 0xFEu;  // map layer outside the map  -> vtab[0xFE] = 1/layer_norm
constexpr uint32_t PAD_CNT = 0xFFu;  // count layers outside the map -> "do not store" (Q2)
constexpr int MAX_CELL_COUNT = 253;  // byte grids: the agents of one layer on ONE cell must stay < PAD_MAP (checked where they are counted)
constexpr int MAX_COUNT = 1023;      // agents per layer.  The authors' largest launch line runs 100 pursuers / 300 evaders
                                     // (runners/old/rllab/pursuit_cnn.sh:1); more than 253 of one kind on one cell raises the overflow bit

struct PursuitDev {
    int32_t xs, ys, P, E, A, R, D;
    int32_t pad, GW, GSZ;  // padded grid: width (y extent), bytes per layer (multiple of 16)
    int32_t n_catch, surround, reward_global, sample_maps, n_maps, max_steps, auto_reset;
    int32_t max_opponents;  // > 0: random_opponents (pursuit_evade.py:177-181)
    int32_t train_pursuit;  // 0: the ACTIONS drive the evaders, the pursuers move by their controller (pursuit_evade.py:215-224)
    int32_t rec_bytes, off_gone, off_term, ngw, ntw;  // state record layout (byte offsets)
    int32_t map_stride;                               // bytes per map entry in `maps`
    uint32_t k0, k1, gid_base;
    float fill32;
    double catchr, term_pursuit, urgency, cw;
    int64_t n_envs;
    const uint8_t *maps;     // per map: padded wall layer [GSZ] then need_to_surround [xs*ys]
    const uint8_t *cnt_tmpl; // padded count-layer template [GSZ]: 0 inside, 0xFF outside
    const float *vtab;       // 256 floats: fl32(k / layer_norm), [0xFE] = fl32(1.0 / layer_norm)
    const uint32_t *codes;   // D slot codes
    const double *cw_env;     // per-env constraint_window / catchr (curriculum, pursuit_evade.py:264-272) or nullptr: the scalars above
    const double *catchr_env;
    uint8_t *state;
    uint32_t *flags;         // [n_envs] flag words (done_flag_word, common.hpp): in the caller's state buffer, behind the stale-zero masks
};

struct PursuitIO {
    const uint8_t *mask;       // reset mode
    const int32_t *inj_pos;    // reset mode
    const int32_t *inj_map;    // reset mode
    const int32_t *actions;    // step mode
    const int32_t *inj_eact;   // step mode
    float *obs;
    float *rew;
    uint8_t *done;
    int32_t *removed;
};

// state record: [u32 tick][u32 t][u32 map_id][u32 spare][u8 xy[2A]][u32 gone[ngw]][u32 term[ntw]]
constexpr int HDR_BYTES = 16;

// `ovf`: set when the cell already held MAX_CELL_COUNT agents -- its count leaves the byte's usable range (254 / 255 are the padding
// sentinels, one more would carry into the neighbouring cell).  The env's results are void from there on: sticky word 3 of its
// record, reported as bit 7 of the done byte (BatchedPursuitEvade raises for it); a new episode clears it.
__device__ __forceinline__ void lds_byte_add(uint8_t *grid, int idx, uint32_t *ovf) {
    const unsigned sh = 8u * (unsigned)(idx & 3);
    const unsigned old = atomicAdd(reinterpret_cast<unsigned *>(grid + (idx & ~3)), 1u << sh);
    if (((old >> sh) & 0xFFu) >= (unsigned)MAX_CELL_COUNT) *ovf = 1u;
}
__device__ __forceinline__ void lds_byte_sub(uint8_t *grid, int idx) {
    atomicSub(reinterpret_cast<unsigned *>(grid + (idx & ~3)), 1u << (8 * (idx & 3)));
}

// numpy float64 add.reduce order (pairwise, 8-way unrolled base case), used by
// `rewards.mean()` at pursuit_evade.py:261.  Base case: 8 <= n <= 128 (or n < 8).
__device__ __forceinline__ double np_pairwise_base(const double *a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}
// numpy's recursive split above 128 elements (pairwise_sum in numpy/core/src/umath/loops_utils.h).  The halves are uneven (n2 = n / 2 rounded
// down to a multiple of 8, the rest goes right): 1 023 -> 504 + 519 -> ... 263 -> 128 + 135, and 135 splits once more -- four levels for
// n <= 1 024 (three left the 49 counts 969 .. 1 023 with a flat sum over more than 128 elements at the bottom: another order of additions)
template <int DEPTH>
__device__ __forceinline__ double np_pairwise_sum_t(const double *a, int n) {
    if (n <= 128) return np_pairwise_base(a, n);
    if constexpr (DEPTH == 0) return np_pairwise_base(a, n);   // (not reached: MAX_COUNT <= 1024, tests/test_advice_regressions.py)
    else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_t<DEPTH - 1>(a, n2) + np_pairwise_sum_t<DEPTH - 1>(a + n2, n - n2);
    }
}
__device__ __forceinline__ double np_pairwise_sum(const double *a, int n) { return np_pairwise_sum_t<4>(a, n); }
static_assert(MAX_COUNT <= 1024, "np_pairwise_sum: four levels of numpy's split");

// mode 0: reset(mask)   mode 1: step (+ fused auto-reset)
template <int NT>
__global__ void pursuit_kernel(const PursuitDev d, const PursuitIO io, const int mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const int nthr = blockDim.x;

    // ---- LDS carve (all offsets multiples of 16)
    float *s_vtab = reinterpret_cast<float *>(smem);                   // 256 floats
    uint8_t *g_map = smem + 1024;                                      // 3 layers, contiguous:
    uint8_t *g_pc = g_map + d.GSZ;                                     //   map | pursuers | evaders
    uint8_t *g_ec = g_pc + d.GSZ;
    uint8_t *g_cr = g_ec + d.GSZ;                                      // credit layer (purs_sur)
    const int A16 = (d.A + 15) & ~15;
    uint8_t *s_ax = g_cr + d.GSZ;
    uint8_t *s_ay = s_ax + A16;
    uint32_t *s_gone = reinterpret_cast<uint32_t *>(s_ay + A16);       // ngw words
    uint32_t *s_term = s_gone + ((d.ngw + 3) & ~3);                    // ntw words
    uint32_t *s_misc = s_term + ((d.ntw + 3) & ~3);                    // [0..3] header, [4] removed
    int32_t *s_kpre = reinterpret_cast<int32_t *>(s_misc + 8);         // P ints (pre-move counts)
    double *s_rew = reinterpret_cast<double *>(s_kpre + ((d.P + 3) & ~3));  // P doubles
    uint32_t *s_code = reinterpret_cast<uint32_t *>(s_rew + ((d.P + 1) & ~1));  // D slot codes (float4 observation path)
    // observers: whose windows the P observation rows show.  train_pursuit: pursuer p.  Evader control (:204-207, :251 with
    // agent_layer = evader_layer): row k = the k-th remaining evader among slots 0..P-1 (collect_obs :418-428 walks
    // range(n_agents()) = range(n_pursuers) over evaders_gone and indexes the compacted layer); s_misc[5] = number of rows
    uint8_t *s_ox = reinterpret_cast<uint8_t *>(s_code + ((d.D + 3) & ~3));
    uint8_t *s_oy = s_ox + ((d.P + 15) & ~15);

    // ---- once per workgroup: value table and this thread's observation slot codes
    for (int k = tid; k < 256; k += nthr) s_vtab[k] = d.vtab[k];
    const bool vec4 = (d.D & 3) == 0;  // rows are whole float4s (always for odd obs_range)
    if (vec4) for (int k = tid; k < d.D; k += nthr) s_code[k] = d.codes[k];
    uint32_t code[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int r = tid + t * nthr;
        code[t] = (r < d.D) ? d.codes[r] : (K_SKIP << 24);
    }
    const int GW = d.GW, pad = d.pad, GSZ = d.GSZ;
    const int obs_off = (d.R - 1) / 2;  // pursuit_evade.py:65
    const int gsz_words = GSZ >> 2;
    int cached_map = -1;

    for (int64_t env = blockIdx.x; env < d.n_envs; env += gridDim.x) {
        if (mode == 0 && io.mask != nullptr && io.mask[env] == 0) continue;  // block-uniform
        uint8_t *rec = d.state + env * (int64_t)d.rec_bytes;
        __syncthreads();  // previous env's LDS reads are finished
        // ------------------------------------------------------------ load state record
        if (tid < 4) s_misc[tid] = reinterpret_cast<const uint32_t *>(rec)[tid];
        if (tid == 4) s_misc[4] = 0;
        for (int a = tid; a < d.A; a += nthr) {
            const uint32_t xy = reinterpret_cast<const uint16_t *>(rec + HDR_BYTES)[a];
            s_ax[a] = (uint8_t)(xy & 0xFF);
            s_ay[a] = (uint8_t)(xy >> 8);
        }
        for (int w = tid; w < d.ngw; w += nthr)
            s_gone[w] = reinterpret_cast<const uint32_t *>(rec + d.off_gone)[w];
        for (int w = tid; w < d.ntw; w += nthr)
            s_term[w] = reinterpret_cast<const uint32_t *>(rec + d.off_term)[w];
        __syncthreads();
        uint32_t tick = s_misc[0];
        int32_t tstep = (int32_t)s_misc[1];
        int32_t map_id = (int32_t)s_misc[2];
        const uint32_t gid = d.gid_base + (uint32_t)env;
        bool do_reset = (mode == 0);
        uint32_t done_bits = 0;

        // -------------------------------------------------------------- observations (:418-461)
        // Element r of pursuer p's row: code[] says which padded-grid byte (relative to the
        // window origin) feeds it.  Stores are lane-contiguous dwords; cells of the count
        // layers outside the map read 0xFF and are NOT stored (reference leaves them stale).
        auto write_obs = [&]() {
#if defined(MADRL_ABLATE) && (MADRL_ABLATE & 8)
            if (d.n_envs >= 0) return;
#endif
            float *orow = io.obs + env * (int64_t)d.P * d.D;
            int n_rows = d.P;
            if (!d.train_pursuit) {  // observers = the remaining evaders of slots 0..P-1, in slot order
                __syncthreads();
                if (tid == 0) {
                    int k = 0;
                    for (int i = 0; i < d.P && i < d.E; ++i)
                        if (!((s_gone[i >> 5] >> (i & 31)) & 1u)) { s_ox[k] = s_ax[d.P + i]; s_oy[k] = s_ay[d.P + i]; ++k; }
                    s_misc[5] = (uint32_t)k;
                }
                __syncthreads();
                n_rows = (int)s_misc[5];
            }
            const uint8_t *obx = d.train_pursuit ? s_ax : s_ox, *oby = d.train_pursuit ? s_ay : s_oy;
            if (vec4) {
                // float4 path (same scheme as the wave kernel): the P*D/4 float4 slots of the env are spread over the threads;
                // a slot without stale cells is ONE non-temporal 16-byte store, a slot with stale cells falls back to masked
                // dword stores (plain, merged in L2).  One float4 instruction touches each 64-byte chunk once, where the
                // per-pursuer dword rows re-touch the row tails (scripts/ubench/vmem_issue.hip).
                typedef float v4f __attribute__((ext_vector_type(4)));
                const int DV = d.D >> 2, NQ = d.P * DV;
                int p = tid / DV, f = tid - p * DV;
                const int dp = nthr / DV, df = nthr - dp * DV;
                for (int q = tid; q < NQ; q += nthr) {
                    if (p >= n_rows) break;  // rows of absent observers keep their old contents
                    const int base = (obx[p] - obs_off + pad) * GW + (oby[p] - obs_off + pad);
                    float val[4];
                    bool keep[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t c = s_code[4 * f + k];
                        const uint32_t kind = c >> 24;
                        keep[k] = true;
                        val[k] = 0.0f;
                        if (kind == K_GRID) {
                            const uint32_t v = g_map[base + (int)(c & 0xFFFFFFu)];
                            keep[k] = v != PAD_CNT;
                            val[k] = s_vtab[v];
                        } else if (kind == K_ID) {
                            val[k] = (float)((double)p / (double)d.P);  // :440-445
                        } else if (kind == K_FILL) {
                            val[k] = d.fill32;  // even obs_range: never-copied channel-0 cells
                        } else {
                            keep[k] = false;
                        }
                    }
                    float *o = orow + 4 * (int64_t)q;
                    if (keep[0] & keep[1] & keep[2] & keep[3]) {
                        const v4f v = {val[0], val[1], val[2], val[3]};
                        __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(o));
                    } else {
                        if (keep[0]) o[0] = val[0];
                        if (keep[1]) o[1] = val[1];
                        if (keep[2]) o[2] = val[2];
                        if (keep[3]) o[3] = val[3];
                    }
                    f += df; p += dp;
                    if (f >= DV) { f -= DV; ++p; }
                }
                return;
            }
            // dword path (rows that are not whole float4s: even obs_range with flatten)
            for (int p = 0; p < n_rows; ++p) {
                const int base = (obx[p] - obs_off + pad) * GW + (oby[p] - obs_off + pad);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int r = tid + t * nthr;
                    const uint32_t kind = code[t] >> 24;
                    if (kind == K_GRID) {
                        const uint32_t v = g_map[base + (int)(code[t] & 0xFFFFFFu)];
                        if (v != PAD_CNT) orow[p * d.D + r] = s_vtab[v];
                    } else if (kind == K_ID) {
                        orow[p * d.D + r] = (float)((double)p / (double)d.P);  // :440-445
                    } else if (kind == K_FILL) {
                        orow[p * d.D + r] = d.fill32;  // even obs_range: never-copied channel-0 cells
                    }
                }
            }
        };


        if (mode == 1) {
            // -------------------------------------------------------- grids for this env
            {
                const uint32_t *mt = reinterpret_cast<const uint32_t *>(d.maps + (int64_t)map_id * d.map_stride);
                const uint32_t *ct = reinterpret_cast<const uint32_t *>(d.cnt_tmpl);
                for (int k = tid; k < gsz_words; k += nthr) {
                    if (cached_map != map_id) reinterpret_cast<uint32_t *>(g_map)[k] = mt[k];
                    const uint32_t c = ct[k];
                    reinterpret_cast<uint32_t *>(g_pc)[k] = c;
                    reinterpret_cast<uint32_t *>(g_ec)[k] = c;
                    reinterpret_cast<uint32_t *>(g_cr)[k] = 0u;
                }
                cached_map = map_id;
            }
            __syncthreads();
            // -------------------------------------------------------- pre-move evader counts (:364-365)
            for (int i = tid; i < d.E; i += nthr) {
                if (!((s_gone[i >> 5] >> (i & 31)) & 1u))
                    lds_byte_add(g_ec, (s_ax[d.P + i] + pad) * GW + s_ay[d.P + i] + pad, &s_misc[3]);
            }
            __syncthreads();
            // proximity reward on the PRE-move state, np.clip keeps border pursuers on their
            // own cell (pursuit_evade.py:374-380)
            for (int p = tid; p < d.P; p += nthr) {
                const int x = s_ax[p], y = s_ay[p];
                const int xm = max(x - 1, 0), xp = min(x + 1, d.xs - 1);
                const int ym = max(y - 1, 0), yp = min(y + 1, d.ys - 1);
                s_kpre[p] = (int)g_ec[(xm + pad) * GW + y + pad] + (int)g_ec[(xp + pad) * GW + y + pad] +
                            (int)g_ec[(x + pad) * GW + yp + pad] + (int)g_ec[(x + pad) * GW + ym + pad];
            }
            __syncthreads();
            // -------------------------------------------------------- moves (:229-241)
            for (int a = tid; a < d.A; a += nthr) {
                const bool is_p = a < d.P;
                const int i = a - d.P;
                if (!is_p && ((s_gone[i >> 5] >> (i & 31)) & 1u)) continue;
                int x = s_ax[a], y = s_ay[a];
                int act;
                int k = 0;  // evaders: index in the evader LAYER = alive evaders in slots below i
                if (!is_p) {
                    lds_byte_sub(g_ec, (x + pad) * GW + y + pad);  // undo the pre-move count
                    for (int w = 0; w < (i >> 5); ++w) k += 32 - __popc(s_gone[w]);
                    k += (i & 31) - __popc(s_gone[i >> 5] & ((1u << (i & 31)) - 1u));
                }
                if (d.train_pursuit) {
                    if (is_p) {
                        act = io.actions[env * d.P + a];
                    } else if (io.inj_eact != nullptr) {
                        act = io.inj_eact[env * d.E + k];
                    } else {
                        const u32x4 r = philox4x32_10(gid, tick, (uint32_t)k, TAG_EVADER_ACT, d.k0, d.k1);
                        act = (int)__umulhi(r.x, 5u);  // RandomPolicy.act, Controllers.py:15-16
                    }
                } else {
                    // evader control (:215-224): action k moves the k-th agent of the evader layer (`for i, a in enumerate(actions):
                    // agent_layer.move_agent(i, a)`, :229-230; the caller passes one action per env.agents entry = n_pursuers of
                    // them, so evaders past the first n_pursuers of the layer never move); every pursuer moves by one
                    // pursuer_controller.act() draw (:238-241), injected as entry a of inj_eact [n_envs][P]
                    if (!is_p) {
                        act = k < d.P ? io.actions[env * d.P + k] : 4;
                    } else if (io.inj_eact != nullptr) {
                        act = io.inj_eact[env * d.P + a];
                    } else {
                        const u32x4 r = philox4x32_10(gid, tick, (uint32_t)a, TAG_PURSUER_ACT, d.k0, d.k1);
                        act = (int)__umulhi(r.x, 5u);
                    }
                }
                // DiscreteAgent.step, DiscreteAgent.py:69-97
                const bool term = (s_term[a >> 5] >> (a & 31)) & 1u;
                if (!term) {
                    if (g_map[(x + pad) * GW + y + pad] == 1) {
                        atomicOr(&s_term[a >> 5], 1u << (a & 31));  // standing in a building
                    } else {
                        int nx = x, ny = y;
                        if (act == 0) nx = x - 1;
                        else if (act == 1) nx = x + 1;
                        else if (act == 2) ny = y + 1;
                        else if (act == 3) ny = y - 1;
                        // padded map layer: 0 = free, 1 = building, 0xFE = outside the map
                        if (g_map[(nx + pad) * GW + ny + pad] == 0) {
                            x = nx;
                            y = ny;
                        }
                    }
                }
                s_ax[a] = (uint8_t)x;
                s_ay[a] = (uint8_t)y;
                lds_byte_add(is_p ? g_pc : g_ec, (x + pad) * GW + y + pad, &s_misc[3]);  // :244-246
            }
            __syncthreads();
            // -------------------------------------------------------- catch resolution (:463-521)
            const uint8_t *need_tab = d.maps + (int64_t)map_id * d.map_stride + GSZ;
            for (int i = tid; i < d.E; i += nthr) {
                if ((s_gone[i >> 5] >> (i & 31)) & 1u) continue;
                const int x = s_ax[d.P + i], y = s_ay[d.P + i];
                const int c0 = (x + pad) * GW + y + pad;
                bool caught;
                if (d.surround) {
                    // neighbour order of surround_mask (:150); pad cells hold 0xFF => never a hit
                    const bool h0 = (uint8_t)(g_pc[c0 - GW] - 1) < 0xFEu;
                    const bool h1 = (uint8_t)(g_pc[c0 + GW] - 1) < 0xFEu;
                    const bool h2 = (uint8_t)(g_pc[c0 + 1] - 1) < 0xFEu;
                    const bool h3 = (uint8_t)(g_pc[c0 - 1] - 1) < 0xFEu;
                    const int cnt = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                    caught = (cnt == (int)need_tab[x * d.ys + y]);  // need_to_surround :523-540
                    if (caught) {  // pursuers standing on a matched neighbour get credit (:489-495)
                        if (h0) g_cr[c0 - GW] = 1;
                        if (h1) g_cr[c0 + GW] = 1;
                        if (h2) g_cr[c0 + 1] = 1;
                        if (h3) g_cr[c0 - 1] = 1;
                    }
                } else {
                    caught = (int)g_pc[c0] >= d.n_catch;  // :498
                    if (caught) g_cr[c0] = 1;             // :503-506
                }
                if (caught) {
                    atomicOr(&s_gone[i >> 5], 1u << (i & 31));
                    atomicAdd(&s_misc[4], 1u);
                }
            }
            __syncthreads();
            // -------------------------------------------------------- rewards (:254-262)
            int n_alive = d.E;
            for (int w = 0; w < d.ngw; ++w) n_alive -= __popc(s_gone[w]);
            for (int p = tid; p < d.P; p += nthr) {
                const int sur = g_cr[(s_ax[p] + pad) * GW + s_ay[p] + pad];
                const double catchr = d.catchr_env ? d.catchr_env[env] : d.catchr;
                double r = catchr * (double)s_kpre[p];
                r += d.term_pursuit * (sur ? 1.0 : 0.0);
                r += d.urgency;
                if (d.reward_global) s_rew[p] = r;
                else io.rew[env * d.P + p] = (float)r;
            }
            if (d.reward_global) {
                __syncthreads();
                if (tid < d.P) {
                    const double m = np_pairwise_sum(s_rew, d.P) / (double)d.P;
                    for (int p = tid; p < d.P; p += nthr) io.rew[env * d.P + p] = (float)m;
                }
            }
            tick += 1;
            tstep += 1;
            if (n_alive == 0) done_bits |= 1u;                               // :383-389
            if (d.max_steps > 0 && tstep >= d.max_steps) done_bits |= 2u;
            const uint32_t overflow = s_misc[3] ? 0x80u : 0u;                // a cell's count left the byte range: results void
            if (tid == 0) {
                io.done[env] = (uint8_t)(done_bits | overflow);
                io.removed[env] = (int32_t)s_misc[4];
                d.flags[env] = done_flag_word(done_bits | overflow);
            }
            do_reset = d.auto_reset && done_bits != 0;
        }

        if (mode == 1 && do_reset) {
            // auto-reset: the reference sequence is step() then reset(); both write the persistent
            // observation buffer, and cells the second write skips keep the first one's values
            write_obs();
            __syncthreads();
        }
        if (do_reset) {
            // ---------------------------------------------------------- reset (:173-207)
            __syncthreads();
            if (tid == 0) s_misc[3] = 0u;                            // a new episode: the overflow mark goes
            for (int w = tid; w < d.ngw; w += nthr) s_gone[w] = 0u;  // :175-176
            for (int w = tid; w < d.ntw; w += nthr) s_term[w] = 0u;  // fresh agents
            if (io.inj_map != nullptr && mode == 0) {
                map_id = io.inj_map[env];
            } else if (d.sample_maps) {  // :182-183
                const u32x4 r = philox4x32_10(gid, tick, 0u, TAG_RESET_ENV, d.k0, d.k1);
                map_id = (int)__umulhi(r.x, (uint32_t)d.n_maps);
            }
            {
                const uint32_t *mt = reinterpret_cast<const uint32_t *>(d.maps + (int64_t)map_id * d.map_stride);
                const uint32_t *ct = reinterpret_cast<const uint32_t *>(d.cnt_tmpl);
                for (int k = tid; k < gsz_words; k += nthr) {
                    if (cached_map != map_id) reinterpret_cast<uint32_t *>(g_map)[k] = mt[k];
                    const uint32_t c = ct[k];
                    reinterpret_cast<uint32_t *>(g_pc)[k] = c;
                    reinterpret_cast<uint32_t *>(g_ec)[k] = c;
                }
                cached_map = map_id;
            }
            // constraint window (:185-191), float64 like the reference
            const u32x4 rw = philox4x32_10(gid, tick, 1u, TAG_RESET_ENV, d.k0, d.k1);
            const double cw = d.cw_env ? d.cw_env[env] : d.cw;
            const double sx = u53(rw.x, rw.y) * (1.0 - cw);
            const double sy = u53(rw.z, rw.w) * (1.0 - cw);
            const int xlb = (int)(d.xs * sx), xub = (int)(d.xs * (sx + cw));
            const int ylb = (int)(d.ys * sy), yub = (int)(d.ys * (sy + cw));
            // random_opponents (train_pursuit, :177-181): this episode has n_create <= E evaders; the slots above are not
            // created and count as gone.  An injected position with x < 0 marks a slot that is not created.
            const bool inj = io.inj_pos != nullptr && mode == 0;
            int n_create = d.E;
            if (d.max_opponents > 0 && !inj) {
                const u32x4 r3 = philox4x32_10(gid, tick, 2u, TAG_RESET_ENV, d.k0, d.k1);
                n_create = min(1 + (int)__umulhi(r3.x, (uint32_t)(d.max_opponents - 1)), d.E);
            }
            __syncthreads();
            for (int a = tid; a < d.A; a += nthr) {  // create_agents, agent_utils.py:12-28
                int x = 0, y = 0;
                if (a >= d.P && (a - d.P >= n_create || (inj && io.inj_pos[(env * d.A + a) * 2] < 0))) {
                    atomicOr(&s_gone[(a - d.P) >> 5], 1u << ((a - d.P) & 31));
                    s_ax[a] = 0;
                    s_ay[a] = 0;
                    continue;
                }
                if (io.inj_pos != nullptr && mode == 0) {
                    x = io.inj_pos[(env * d.A + a) * 2];
                    y = io.inj_pos[(env * d.A + a) * 2 + 1];
                } else {
                    // feasible_position: rejection sampling (agent_utils.py:37-47); bounded
                    for (uint32_t att = 0; att < 1024u; ++att) {
                        const u32x4 r = philox4x32_10(gid, tick, (uint32_t)a, TAG_RESET_POS | (att << 8), d.k0, d.k1);
                        x = xlb + (int)__umulhi(r.x, (uint32_t)(xub - xlb));
                        y = ylb + (int)__umulhi(r.y, (uint32_t)(yub - ylb));
                        if (g_map[(x + pad) * GW + y + pad] != 1) break;
                    }
                }
                s_ax[a] = (uint8_t)x;
                s_ay[a] = (uint8_t)y;
                lds_byte_add(a < d.P ? g_pc : g_ec, (x + pad) * GW + y + pad, &s_misc[3]);  // :201-203
            }
            tick += 1;
            tstep = 0;
            __syncthreads();
        }

        write_obs();
        // -------------------------------------------------------------- store state record
        for (int a = tid; a < d.A; a += nthr)
            reinterpret_cast<uint16_t *>(rec + HDR_BYTES)[a] = (uint16_t)(s_ax[a] | (s_ay[a] << 8));
        for (int w = tid; w < d.ngw; w += nthr) reinterpret_cast<uint32_t *>(rec + d.off_gone)[w] = s_gone[w];
        for (int w = tid; w < d.ntw; w += nthr) reinterpret_cast<uint32_t *>(rec + d.off_term)[w] = s_term[w];
        if (tid == 0) {
            uint32_t *h = reinterpret_cast<uint32_t *>(rec);
            h[0] = tick;
            h[1] = (uint32_t)tstep;
            h[2] = (uint32_t)map_id;
            h[3] = s_misc[3];   // sticky count-overflow mark of the episode (0 in every run that stays inside the byte grids)
        }
    }
}

// ------------------------------------------------------------------ state (un)packing
__global__ void pursuit_get_state_kernel(const PursuitDev d, int32_t *pos_p, int32_t *pos_e, uint8_t *gone,
                                         uint8_t *term_p, uint8_t *term_e, int32_t *map_id, uint32_t *tick,
                                         int32_t *t) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    const uint8_t *rec = d.state + env * (int64_t)d.rec_bytes;
    const uint32_t *h = reinterpret_cast<const uint32_t *>(rec);
    const uint8_t *xy = rec + HDR_BYTES;
    const uint32_t *gw = reinterpret_cast<const uint32_t *>(rec + d.off_gone);
    const uint32_t *tw = reinterpret_cast<const uint32_t *>(rec + d.off_term);
    if (tick) tick[env] = h[0];
    if (t) t[env] = (int32_t)h[1];
    if (map_id) map_id[env] = (int32_t)h[2];
    for (int p = 0; p < d.P; ++p) {
        if (pos_p) {
            pos_p[(env * d.P + p) * 2] = xy[2 * p];
            pos_p[(env * d.P + p) * 2 + 1] = xy[2 * p + 1];
        }
        if (term_p) term_p[env * d.P + p] = (tw[p >> 5] >> (p & 31)) & 1u;
    }
    for (int i = 0; i < d.E; ++i) {
        const uint32_t g = (gw[i >> 5] >> (i & 31)) & 1u;
        const int a = d.P + i;
        if (gone) gone[env * d.E + i] = (uint8_t)g;
        if (pos_e) {
            pos_e[(env * d.E + i) * 2] = g ? -1 : (int32_t)xy[2 * a];
            pos_e[(env * d.E + i) * 2 + 1] = g ? -1 : (int32_t)xy[2 * a + 1];
        }
        if (term_e) term_e[env * d.E + i] = g ? 0 : (uint8_t)((tw[a >> 5] >> (a & 31)) & 1u);
    }
}

__global__ void pursuit_set_state_kernel(const PursuitDev d, const int32_t *pos_p, const int32_t *pos_e,
                                         const uint8_t *gone, const uint8_t *term_p, const uint8_t *term_e,
                                         const int32_t *map_id, const uint32_t *tick, const int32_t *t) {
    const int64_t env = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    uint8_t *rec = d.state + env * (int64_t)d.rec_bytes;
    uint32_t *h = reinterpret_cast<uint32_t *>(rec);
    uint8_t *xy = rec + HDR_BYTES;
    uint32_t *gw = reinterpret_cast<uint32_t *>(rec + d.off_gone);
    uint32_t *tw = reinterpret_cast<uint32_t *>(rec + d.off_term);
    if (tick) h[0] = tick[env];
    if (t) h[1] = (uint32_t)t[env];
    if (map_id) h[2] = (uint32_t)map_id[env];
    for (int p = 0; p < d.P; ++p) {
        if (pos_p) {
            xy[2 * p] = (uint8_t)pos_p[(env * d.P + p) * 2];
            xy[2 * p + 1] = (uint8_t)pos_p[(env * d.P + p) * 2 + 1];
        }
        if (term_p) {
            if (term_p[env * d.P + p]) tw[p >> 5] |= 1u << (p & 31);
            else tw[p >> 5] &= ~(1u << (p & 31));
        }
    }
    for (int i = 0; i < d.E; ++i) {
        const int a = d.P + i;
        bool g = (gw[i >> 5] >> (i & 31)) & 1u;
        if (gone) {
            g = gone[env * d.E + i] != 0;
            if (g) gw[i >> 5] |= 1u << (i & 31);
            else gw[i >> 5] &= ~(1u << (i & 31));
        }
        if (pos_e && !g) {
            xy[2 * a] = (uint8_t)pos_e[(env * d.E + i) * 2];
            xy[2 * a + 1] = (uint8_t)pos_e[(env * d.E + i) * 2 + 1];
        }
        if (term_e) {
            if (term_e[env * d.E + i] && !g) tw[a >> 5] |= 1u << (a & 31);
            else tw[a >> 5] &= ~(1u << (a & 31));
        }
    }
}

}  // namespace

// =================================================================== host side / C ABI
struct WaveEntry;  // one compiled specialisation of pursuit_wave_kernel

struct madrl_pursuit {
    madrl_pursuit_config cfg;
    PursuitDev dev;
    int device;
    int threads;
    int nt;
    int64_t max_blocks;
    size_t lds_bytes;
    void *tables;  // one device allocation holding maps | cnt_tmpl | vtab | codes
    // one-wavefront-per-env fast path (pursuit_wave.hpp), when a specialisation matches
    const WaveEntry *wave;
    madrl::pw::WaveDev wdev;
    void *zmask = nullptr;          // stale-zero masks of the fast path, [n_envs][64 * waves] dwords (pursuit_wave.hpp): the tail of
                                    // the caller's state buffer (madrl_pursuit_state_bytes), not a library allocation
    const void *zmask_obs = nullptr; // the observation buffer the masks describe; another buffer, a generic-kernel launch,
                                    // set_state or madrl_pursuit_invalidate_obs resets them to "nothing known"
    uint64_t step_count = 0;  // step launches so far (parity of the walk direction, see launch())
    int walk_mode = 0;        // 0 auto (alternate above ~375 MB per launch), 1 always alternate, 2 always forward; fixed at create
    void *wtables;
    int kernel_kind;  // MADRL_KERNEL_AUTO / _GENERIC / _WAVE (requested)
    hipEvent_t ev_fork = nullptr, ev_done = nullptr;   // madrl_pursuit_step_sharded: made by madrl_pursuit_create on the handle's device, destroyed with the handle
};

namespace {

// ------------------------------------------------------------------ wave-kernel specialisations
struct WaveGeom {
    int xs, ys, P, E, R, flatten;
    int GW, PAD, GSZ, D, X_ID, X_SKIP;
    int rec_bytes, off_gone, off_term;
    int waves;  // wavefronts per env: 1 = pursuit_wave_kernel, > 1 = pursuit_group_kernel
    int occ;    // resident wavefronts per SIMD the kernel's registers are allocated for
    int mwords; // stale-zero mask dwords per lane (1: one bit per float4 slot in each byte, up to 8 slots per lane; the row-loop kernel: P / 8)
};
}  // namespace

struct WaveEntry {
    WaveGeom g;
    void (*launch)(const madrl::pw::WaveDev &, const madrl::pw::WaveIO &, int mode, int64_t blocks, hipStream_t s);
};

namespace {

template <class S>
void wave_launch(const pw::WaveDev &d, const pw::WaveIO &io, int mode, int64_t blocks, hipStream_t s) {
    if constexpr (S::E >= S::P) {
        if (io.control_evaders) {   // train_pursuit=False: the evader-control instantiations (reset, flexible step)
            if (mode == 0) hipLaunchKernelGGL((pw::pursuit_wave_kernel<S, 0, false, true>), dim3((unsigned)blocks), dim3(64), 0, s, d, io);
            else hipLaunchKernelGGL((pw::pursuit_wave_kernel<S, 1, true, true>), dim3((unsigned)blocks), dim3(64), 0, s, d, io);
            return;
        }
    }
    if (mode == 0)
        hipLaunchKernelGGL((pw::pursuit_wave_kernel<S, 0, false>), dim3((unsigned)blocks), dim3(64), 0, s, d, io);
    else if (io.flex)
        hipLaunchKernelGGL((pw::pursuit_wave_kernel<S, 1, true>), dim3((unsigned)blocks), dim3(64), 0, s, d, io);
    else
        hipLaunchKernelGGL((pw::pursuit_wave_kernel<S, 1, false>), dim3((unsigned)blocks), dim3(64), 0, s, d, io);
}

template <class S>
void group_launch(const pw::WaveDev &d, const pw::WaveIO &io, int mode, int64_t blocks, hipStream_t s) {
    if (mode == 0)
        hipLaunchKernelGGL((pw::pursuit_group_kernel<S, 0, false>), dim3((unsigned)blocks), dim3(S::NT), 0, s, d, io);
    else if (io.flex)
        hipLaunchKernelGGL((pw::pursuit_group_kernel<S, 1, true>), dim3((unsigned)blocks), dim3(S::NT), 0, s, d, io);
    else
        hipLaunchKernelGGL((pw::pursuit_group_kernel<S, 1, false>), dim3((unsigned)blocks), dim3(S::NT), 0, s, d, io);
}

template <class S>
constexpr WaveGeom wave_geom(int waves = 1, int occ = 4) {
    return WaveGeom{S::XS, S::YS, S::P, S::E, S::R, S::FLATTEN, S::GW, S::PAD, S::GSZ, S::D, S::X_ID, S::X_SKIP,
                    S::REC_BYTES, S::OFF_GONE, S::OFF_TERM, waves, occ, S::MWORDS};
}

#define X(XS, YS, NP, NE, R, FL) {wave_geom<pw::Shape<XS, YS, NP, NE, R, FL>>(1, pw::Shape<XS, YS, NP, NE, R, FL>::OCC), wave_launch<pw::Shape<XS, YS, NP, NE, R, FL>>},
#define XG(XS, YS, NP, NE, R, FL, NW)
const WaveEntry WAVE_TABLE[] = {
#include "pursuit_specializations.def"
#if __has_include("pursuit_specializations.local.def")   // shapes added on this machine by `python -m madrl_amd.build --pursuit-shape ...` (git-ignored)
#include "pursuit_specializations.local.def"
#endif
#undef X
#undef XG
#define X(XS, YS, NP, NE, R, FL)
#define XG(XS, YS, NP, NE, R, FL, NW) {wave_geom<pw::GShape<XS, YS, NP, NE, R, FL, NW>>(NW, pw::GShape<XS, YS, NP, NE, R, FL, NW>::OCC), group_launch<pw::GShape<XS, YS, NP, NE, R, FL, NW>>},
#include "pursuit_specializations.def"
#if __has_include("pursuit_specializations.local.def")   // shapes added on this machine by `python -m madrl_amd.build --pursuit-shape ...` (git-ignored)
#include "pursuit_specializations.local.def"
#endif
};
#undef X
#undef XG

const WaveEntry *find_wave(const madrl_pursuit_config *c) {
    if (c->flatten && !c->include_id) return nullptr;
    for (const WaveEntry &e : WAVE_TABLE) {
        const WaveGeom &g = e.g;
        if (c->control_evaders && g.waves > 1) continue;  // evader control: the one-wavefront kernel or the generic one
        if (g.xs == c->xs && g.ys == c->ys && g.P == c->n_pursuers && g.E == c->n_evaders && g.R == c->obs_range &&
            g.flatten == (c->flatten ? 1 : 0))
            return &e;
    }
    return nullptr;
}


int validate(const madrl_pursuit_config *c) {
    if (!c) return fail(MADRL_EINVAL, "config is NULL");
    if (c->struct_size != (int32_t)sizeof(madrl_pursuit_config))
        return fail(MADRL_EINVAL, "madrl_pursuit_config.struct_size=%d, library expects %d", c->struct_size,
                    (int)sizeof(madrl_pursuit_config));
    if (c->xs < 1 || c->ys < 1 || c->xs > 255 || c->ys > 255)
        return fail(MADRL_EINVAL, "map size %dx%d unsupported (1..255)", c->xs, c->ys);
    if (c->n_pursuers < 1 || c->n_evaders < 0 || c->n_pursuers > MAX_COUNT || c->n_evaders > MAX_COUNT)
        return fail(MADRL_EINVAL, "n_pursuers=%d n_evaders=%d unsupported (pursuers 1..%d, evaders 0..%d: "
                    "byte count grids)", c->n_pursuers, c->n_evaders, MAX_COUNT, MAX_COUNT);
    if (c->obs_range < 1 || c->obs_range > 63) return fail(MADRL_EINVAL, "obs_range=%d unsupported", c->obs_range);
    if (c->n_maps < 1) return fail(MADRL_EINVAL, "n_maps must be >= 1");
    if (!(c->layer_norm > 0.0)) return fail(MADRL_EINVAL, "layer_norm must be > 0");
    if (!(c->constraint_window > 0.0 && c->constraint_window <= 1.0))
        return fail(MADRL_EINVAL, "constraint_window must be in (0,1]");
    if (c->control_evaders != 0 && c->control_evaders != 1) return fail(MADRL_EINVAL, "control_evaders must be 0 or 1");
    if (c->control_evaders && c->n_evaders < c->n_pursuers)
        return fail(MADRL_EINVAL, "control_evaders=1 (train_pursuit=False): collect_obs walks range(n_pursuers) over evaders_gone (pursuit_evade.py:418-428): n_evaders=%d must be >= n_pursuers=%d",
                    c->n_evaders, c->n_pursuers);
    if (c->control_evaders && c->max_opponents != 0)
        return fail(MADRL_EINVAL, "control_evaders=1 (train_pursuit=False) with random_opponents (a per-reset number of pursuers, :180-181) is not supported");
    if (c->max_opponents != 0 && c->max_opponents < 2) return fail(MADRL_EINVAL, "max_opponents=%d: random_opponents draws randint(1, max_opponents)", c->max_opponents);
    return MADRL_OK;
}

int obs_dim_of(const madrl_pursuit_config *c) {
    const int R = c->obs_range;
    return c->flatten ? 3 * R * R + (c->include_id ? 1 : 0) : 4 * R * R;
}

void layout(const madrl_pursuit_config *c, PursuitDev *d) {
    memset(d, 0, sizeof(*d));
    d->xs = c->xs; d->ys = c->ys; d->P = c->n_pursuers; d->E = c->n_evaders; d->A = d->P + d->E;
    d->R = c->obs_range; d->D = obs_dim_of(c);
    const int off = (c->obs_range - 1) / 2;
    d->pad = off > 1 ? off : 1;
    d->GW = c->ys + 2 * d->pad;
    d->GSZ = (int)align_up((size_t)(c->xs + 2 * d->pad) * d->GW, 16);
    d->n_catch = c->n_catch; d->surround = c->surround; d->reward_global = c->reward_global;
    d->sample_maps = c->sample_maps; d->n_maps = c->n_maps; d->max_steps = c->max_steps;
    d->auto_reset = c->auto_reset;
    d->max_opponents = c->max_opponents;
    d->train_pursuit = !c->control_evaders;
    d->ngw = (d->E + 31) / 32; if (d->ngw < 1) d->ngw = 1;
    d->ntw = (d->A + 31) / 32;
    d->off_gone = (int)align_up(HDR_BYTES + 2 * (size_t)d->A, 4);
    d->off_term = d->off_gone + 4 * d->ngw;
    d->rec_bytes = (int)align_up((size_t)d->off_term + 4 * d->ntw, 16);
    d->map_stride = (int)align_up((size_t)d->GSZ + (size_t)c->xs * c->ys, 16);
    d->k0 = (uint32_t)c->seed; d->k1 = (uint32_t)(c->seed >> 32);
    d->gid_base = (uint32_t)c->env_id_base;
    d->catchr = c->catchr; d->term_pursuit = c->term_pursuit; d->urgency = c->urgency_reward;
    d->cw = c->constraint_window;
    d->fill32 = (float)(1.0 / c->layer_norm);  // local_obs[..][0].fill(1.0 / layer_norm), :433
}

// need_to_surround(x, y), pursuit_evade.py:523-540, tabulated per map cell
int need_to_surround(const int8_t *map, int xs, int ys, int x, int y) {
    static const int mx[4] = {-1, 1, 0, 0}, my[4] = {0, 0, 1, -1};
    int tosur = 4;
    if (x == 0 || x == xs - 1) tosur -= 1;
    if (y == 0 || y == ys - 1) tosur -= 1;
    for (int m = 0; m < 4; ++m) {
        const int xn = x + mx[m], yn = y + my[m];
        if (!(0 < xn && xn < xs) || !(0 < yn && yn < ys)) continue;  // sic: row/col 0 skipped
        if (map[xn * ys + yn] == -1) tosur -= 1;
    }
    return tosur;
}

size_t lds_bytes_for(const PursuitDev &d) {
    size_t b = 1024 + 4 * (size_t)d.GSZ;
    b += 2 * align_up((size_t)d.A, 16);
    b += 4 * align_up((size_t)d.ngw, 4) + 4 * align_up((size_t)d.ntw, 4) + 32;
    b += 4 * align_up((size_t)d.P, 4);
    b = align_up(b, 8);
    b += 8 * align_up((size_t)d.P, 2);
    b += 4 * align_up((size_t)d.D, 4);  // slot codes
    b += 2 * align_up((size_t)d.P, 16);  // observer positions (evader control)
    return align_up(b, 16);
}

template <int NT>
void launch_nt(const madrl_pursuit *h, const PursuitIO &io, int mode, hipStream_t s) {
    int64_t blocks = h->dev.n_envs;
    if (h->max_blocks > 0 && blocks > h->max_blocks) blocks = h->max_blocks;
    hipLaunchKernelGGL(pursuit_kernel<NT>, dim3((unsigned)blocks), dim3((unsigned)h->threads), h->lds_bytes, s,
                       h->dev, io, mode);
}

bool use_wave(const madrl_pursuit *h) {
    return h->wave != nullptr && h->kernel_kind != MADRL_KERNEL_GENERIC;
}

int launch(madrl_pursuit *h, const PursuitIO &io, int mode, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (use_wave(h)) {
        pw::WaveIO w;
        w.mask = io.mask; w.inj_pos = io.inj_pos; w.inj_map = io.inj_map; w.actions = io.actions;
        w.inj_eact = io.inj_eact; w.obs = io.obs; w.rew = io.rew; w.done = io.done; w.removed = io.removed;
        w.flex = (io.inj_eact != nullptr || h->dev.catchr_env != nullptr) ? 1 : 0;
        w.control_evaders = h->dev.train_pursuit ? 0 : 1;
        int64_t blocks = h->max_blocks > 0 ? h->max_blocks : 256 * 4 * h->wave->g.occ / h->wave->g.waves;  // exactly the resident capacity
        if (blocks > h->dev.n_envs) blocks = h->dev.n_envs;
        // Large batches: successive step launches walk the env range in opposite directions, so the rows written last by
        // one step are the first ones touched by the next while they are still in the 256 MB memory-side cache.  Measured
        // (scripts/sweep_wave.py, C2 shape): forward-only holds 7.2e8 env-steps/s up to 73 728 envs and collapses beyond
        // (98 304: 4.7e8, 131 072: 4.5e8); alternating holds 6.4-6.7e8 from 81 920 to 131 072 but costs 6 % below.  Hence
        // the switch at ~375 MB of rows + records per launch.  Env results do not depend on the processing order.
        pw::WaveDev wd = h->wdev;
        wd.catchr = h->dev.catchr; wd.cw = h->dev.cw; wd.cw_env = h->dev.cw_env; wd.catchr_env = h->dev.catchr_env;  // curriculum
        if (h->zmask_obs != (const void *)io.obs) {  // unknown buffer contents: every cell "not known to be zero"
            MADRL_HIP_TRY(hipMemsetAsync(h->zmask, 0xFF, (size_t)h->dev.n_envs * 256 * h->wave->g.waves * h->wave->g.mwords, s));
            h->zmask_obs = io.obs;
        }
        if (mode == 1) {
            bool alternate = (double)h->dev.n_envs * (4.0 * h->dev.P * h->dev.D + 2.0 * h->dev.rec_bytes) > 375e6;
            if (h->walk_mode != 0) alternate = h->walk_mode == 1;
            if (alternate) wd.reverse = (int32_t)(h->step_count++ & 1);
        }
        h->wave->launch(wd, w, mode, blocks, s);
        MADRL_HIP_TRY(hipGetLastError());
        return MADRL_OK;
    }
    h->zmask_obs = nullptr;  // the generic kernel does not maintain the fast path's stale-zero masks
    switch (h->nt) {
        case 1: launch_nt<1>(h, io, mode, s); break;
        case 2: launch_nt<2>(h, io, mode, s); break;
        case 3: launch_nt<3>(h, io, mode, s); break;
        case 4: launch_nt<4>(h, io, mode, s); break;
        case 5: launch_nt<5>(h, io, mode, s); break;
        case 6: launch_nt<6>(h, io, mode, s); break;
        case 7: launch_nt<7>(h, io, mode, s); break;
        case 8: launch_nt<8>(h, io, mode, s); break;
        default: return fail(MADRL_EINVAL, "internal: nt=%d", h->nt);
    }
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

// caller-owned state buffer = [n_envs packed records][pad to 256 B][stale-zero masks of the fast path, 256 B per env, wavefront and
// mask word][flag plane, one dword per env]
uint64_t zmask_offset(int rec_bytes, int64_t n_envs) { return align_up((uint64_t)rec_bytes * (uint64_t)n_envs, 256); }
uint64_t zmask_bytes(const madrl_pursuit_config *cfg, int64_t n_envs) {
    const WaveEntry *w = find_wave(cfg);
    return w ? (uint64_t)n_envs * 256u * (uint64_t)w->g.waves * (uint64_t)w->g.mwords : 0u;
}
uint64_t flags_offset(const madrl_pursuit_config *cfg, int rec_bytes, int64_t n_envs) { return zmask_offset(rec_bytes, n_envs) + zmask_bytes(cfg, n_envs); }

int pick_threads(const PursuitDev &d, int requested) {
    int thr = requested;
    if (thr <= 0) {
        // one wavefront unless the observation row needs more than 8 slots per lane: agents loop over
        // lanes (measured at C5, 76 agents: 64 threads 171 us, 128 threads 253 us, 256 threads 388 us)
        thr = 64;
        while (thr < 1024 && (d.D + thr - 1) / thr > 8) thr += 64;
    }
    return thr;
}

}  // namespace

extern "C" {

int madrl_pursuit_obs_dim(const madrl_pursuit_config *cfg, int32_t *out_dim) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (!out_dim) return fail(MADRL_EINVAL, "out_dim is NULL");
    *out_dim = obs_dim_of(cfg);
    return MADRL_OK;
}

int madrl_pursuit_state_bytes(const madrl_pursuit_config *cfg, int64_t n_envs, uint64_t *out_bytes) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_bytes) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_bytes non-NULL");
    PursuitDev d;
    layout(cfg, &d);
    *out_bytes = flags_offset(cfg, d.rec_bytes, n_envs) + 4u * (uint64_t)n_envs;
    return MADRL_OK;
}

int madrl_pursuit_flags_offset(const madrl_pursuit_config *cfg, int64_t n_envs, uint64_t *out_offset) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (n_envs < 1 || !out_offset) return fail(MADRL_EINVAL, "n_envs must be >= 1 and out_offset non-NULL");
    PursuitDev d;
    layout(cfg, &d);
    *out_offset = flags_offset(cfg, d.rec_bytes, n_envs);
    return MADRL_OK;
}

int madrl_pursuit_record_bytes(const madrl_pursuit_config *cfg, int32_t *out_bytes) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (!out_bytes) return fail(MADRL_EINVAL, "out_bytes is NULL");
    PursuitDev d;
    layout(cfg, &d);
    *out_bytes = d.rec_bytes;
    return MADRL_OK;
}

int madrl_pursuit_create(const madrl_pursuit_config *cfg, const int8_t *map_pool_host, int64_t n_envs,
                         int32_t device, void *state_dev, madrl_pursuit **out) {
    int rc = validate(cfg);
    if (rc) return rc;
    if (!map_pool_host || !state_dev || !out || n_envs < 1)
        return fail(MADRL_EINVAL, "create: NULL argument or n_envs < 1");
    if (n_envs + cfg->env_id_base > 0xFFFFFFFFll)
        return fail(MADRL_EINVAL, "global env index must fit 32 bits");
    MADRL_HIP_TRY(hipSetDevice(device));
    madrl_pursuit *h = new (std::nothrow) madrl_pursuit();
    if (!h) return fail(MADRL_ENOMEM, "out of host memory");
    h->cfg = *cfg;
    h->device = device;
    layout(cfg, &h->dev);
    PursuitDev &d = h->dev;
    d.n_envs = n_envs;
    d.state = (uint8_t *)state_dev;
    d.flags = reinterpret_cast<uint32_t *>((uint8_t *)state_dev + flags_offset(cfg, d.rec_bytes, n_envs));
    const int xs = d.xs, ys = d.ys, pad = d.pad, GW = d.GW;
    const size_t cells = (size_t)xs * ys;

    // ---- host-side tables
    const size_t maps_bytes = (size_t)d.map_stride * d.n_maps;
    const size_t off_cnt = align_up(maps_bytes, 16);
    const size_t off_vtab = align_up(off_cnt + d.GSZ, 16);
    const size_t off_codes = off_vtab + 256 * sizeof(float);
    const size_t total = align_up(off_codes + sizeof(uint32_t) * d.D, 16);
    std::vector<uint8_t> host(total, 0);
    for (int m = 0; m < d.n_maps; ++m) {
        const int8_t *map = map_pool_host + (size_t)m * cells;
        uint8_t *wall = host.data() + (size_t)m * d.map_stride;
        uint8_t *need = wall + d.GSZ;
        memset(wall, PAD_MAP, d.GSZ);
        for (int x = 0; x < xs; ++x)
            for (int y = 0; y < ys; ++y) {
                const int8_t v = map[x * ys + y];
                if (v != 0 && v != -1) {
                    delete h;
                    return fail(MADRL_EINVAL, "map %d cell (%d,%d) = %d, expected 0 or -1", m, x, y, (int)v);
                }
                wall[(x + pad) * GW + y + pad] = (v == -1) ? 1 : 0;
                need[x * ys + y] = (uint8_t)need_to_surround(map, xs, ys, x, y);
            }
    }
    {
        uint8_t *ct = host.data() + off_cnt;
        memset(ct, PAD_CNT, d.GSZ);
        for (int x = 0; x < xs; ++x)
            for (int y = 0; y < ys; ++y) ct[(x + pad) * GW + y + pad] = 0;
    }
    {
        // np.abs(model_state) / layer_norm is float32 / weak python scalar -> float32 (:438-439)
        float *vt = reinterpret_cast<float *>(host.data() + off_vtab);
        const float norm32 = (float)cfg->layer_norm;
        for (int k = 0; k < 256; ++k) vt[k] = (float)k / norm32;
        vt[PAD_MAP] = d.fill32;
        vt[PAD_CNT] = 0.0f;
    }
    {
        // slot codes: which padded-grid byte (relative to the window origin) feeds element r
        uint32_t *codes = reinterpret_cast<uint32_t *>(host.data() + off_codes);
        const int R = d.R, off = (R - 1) / 2, W = 2 * off + 1;  // W: copied window width (:451-461)
        for (int r = 0; r < d.D; ++r) {
            int c, i, j;
            if (cfg->flatten) {
                if (r == 3 * R * R) { codes[r] = K_ID << 24; continue; }  // :444-445
                c = r / (R * R); i = (r % (R * R)) / R; j = r % R;
            } else {
                c = r % 4; i = (r / 4) / R; j = (r / 4) % R;  // rollaxis -> (R,R,4), :449
                if (c == 3) {  // :440-441: only the centre of channel 3 is ever written
                    codes[r] = ((i == R / 2 && j == R / 2) ? K_ID : K_SKIP) << 24;
                    continue;
                }
            }
            if (i < W && j < W) codes[r] = (K_GRID << 24) | (uint32_t)(c * d.GSZ + i * GW + j);
            else codes[r] = (c == 0 ? K_FILL : K_SKIP) << 24;  // even obs_range (Q11)
        }
    }
    hipError_t e = hipMalloc(&h->tables, total);
    if (e != hipSuccess) {
        delete h;
        return fail(MADRL_EHIP, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    }
    e = hipMemcpy(h->tables, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(h->tables);
        delete h;
        return fail(MADRL_EHIP, "hipMemcpy tables failed: %s", hipGetErrorString(e));
    }
    uint8_t *tb = (uint8_t *)h->tables;
    d.maps = tb;
    d.cnt_tmpl = tb + off_cnt;
    d.vtab = reinterpret_cast<const float *>(tb + off_vtab);
    d.codes = reinterpret_cast<const uint32_t *>(tb + off_codes);

    // ---- one-wavefront-per-env fast path tables (pursuit_wave.hpp)
    h->wave = find_wave(cfg);
    h->wtables = nullptr;
    h->kernel_kind = MADRL_KERNEL_AUTO;
    if (h->wave) {
        const WaveGeom &g = h->wave->g;
        const int need_words = ((int)cells + 3) / 4;
        const int fstride = g.GSZ + need_words;
        const size_t w_codes = (size_t)fstride * d.n_maps;
        // after the maps: slot codes [D] | empty count layer [GSZ] | per-thread slot table [NS][6][NT] -- or, for shapes with more than 8
        // slots per thread (pursuit_group.hpp, TABLED), one packed entry of two dwords per float4 of a pursuer's row [DV][2]
        const int NT = 64 * g.waves, DV = d.D / 4, NQ = d.P * DV, NS = (NQ + NT - 1) / NT;
        const bool tabled = NS > 8;
        const size_t w_tmpl = w_codes + (size_t)d.D, w_slots = w_tmpl + (size_t)g.GSZ;
        std::vector<uint32_t> wh(w_slots + (tabled ? (size_t)2 * DV : (size_t)NS * 6 * NT), 0u);
        const float wallv = (float)1 / (float)cfg->layer_norm;  // |-1| / layer_norm in float32
        uint32_t wall_bits, fill_bits;
        memcpy(&wall_bits, &wallv, 4);
        memcpy(&fill_bits, &d.fill32, 4);
        for (int m = 0; m < d.n_maps; ++m) {
            const int8_t *map = map_pool_host + (size_t)m * cells;
            uint32_t *fm = wh.data() + (size_t)m * fstride;
            uint8_t *need = reinterpret_cast<uint8_t *>(fm + g.GSZ);
            for (int k = 0; k < g.GSZ; ++k) fm[k] = fill_bits;
            for (int x = 0; x < xs; ++x)
                for (int y = 0; y < ys; ++y) {
                    fm[(x + g.PAD) * g.GW + y + g.PAD] = (map[x * ys + y] == -1) ? wall_bits : 0u;
                    need[x * ys + y] = (uint8_t)need_to_surround(map, xs, ys, x, y);
                }
        }
        uint32_t *wc = wh.data() + w_codes;
        const int R = d.R;
        // the fast path tells 0.0f, observation values and its "outside the map" marker apart by the top byte of the value:
        // every non-zero observation value must be a positive float >= 2^-63 (top byte 0x20..0x7F)
        uint32_t unit_bits;
        { const float unit = (float)1 / (float)cfg->layer_norm; memcpy(&unit_bits, &unit, 4); }
        bool eligible = (wall_bits != 0u) && (fill_bits != 0u) && (unit_bits >> 24) >= 0x20u && (unit_bits >> 24) < 0x80u &&
                        (fill_bits >> 24) >= 0x20u && (fill_bits >> 24) < 0x80u && g.rec_bytes == d.rec_bytes &&
                        g.off_gone == d.off_gone && g.off_term == d.off_term && g.D == d.D &&
                        n_envs < 0x7FF00000ll;  // the fast kernels index envs with 32-bit integers (index + workgroup count < 2^31)
        for (int r = 0; r < d.D; ++r) {
            int c, i, j;
            if (cfg->flatten) {
                if (r == 3 * R * R) { wc[r] = (uint32_t)g.X_ID; continue; }
                c = r / (R * R); i = (r % (R * R)) / R; j = r % R;
            } else {
                c = r % 4; i = (r / 4) / R; j = (r / 4) % R;
                if (c == 3) { wc[r] = (uint32_t)((i == R / 2 && j == R / 2) ? g.X_ID : g.X_SKIP); continue; }
            }
            wc[r] = 0x80000000u | (uint32_t)(c * g.GSZ + i * g.GW + j);
        }
        for (int r = 0; r < d.D; ++r)  // the kernel assumes only element 3 of a float4 can be absolute
            if ((r & 3) != 3 && !(wc[r] >> 31)) eligible = false;
        for (int k = 0; k < g.GSZ; ++k) {
            const int gx = k / g.GW - g.PAD, gy = k % g.GW - g.PAD;
            wh[w_tmpl + k] = (gx >= 0 && gx < xs && gy >= 0 && gy < ys) ? 0u : pw::SENT;
        }
        if (tabled) {
            // entry f: dword 0 = off0 | off1 << 16, dword 1 = off2 | id3 << 15 | off3 << 16 | abs3 << 31 -- dword offsets into the LDS array,
            // relative to the window origin unless abs3 (element 3 only: the skip cell, or with id3 the id cell of pursuer 0 + pursuer)
            if (3 * g.GSZ + 2 + d.P >= 32768 || g.mwords != (NS + 7) / 8) eligible = false;
            for (int f = 0; f < DV && eligible; ++f) {
                uint32_t off[4], abs3 = 0u, id3 = 0u;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = wc[4 * f + k];
                    off[k] = c & 0x7FFFFFFFu;
                    if (k == 3 && !(c >> 31)) { abs3 = 1u; id3 = ((int)off[k] == g.X_ID) ? 1u : 0u; }
                    if (off[k] >= 32768u) eligible = false;
                }
                wh[w_slots + 2 * (size_t)f] = off[0] | (off[1] << 16);
                wh[w_slots + 2 * (size_t)f + 1] = off[2] | (id3 << 15) | (off[3] << 16) | (abs3 << 31);
            }
        }
        for (int sl = 0; sl < (tabled ? 0 : NS); ++sl)
            for (int t = 0; t < NT; ++t) {
                const int q = t + NT * sl, pidx = q / DV, f = q % DV;
                uint32_t *row = wh.data() + w_slots + (size_t)sl * 6 * NT + t;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = (q < NQ) ? wc[4 * f + k] : (k == 3 ? (uint32_t)g.X_SKIP : 0u);
                    uint32_t cst = c & 0x7FFFFFFFu;
                    if ((int)cst >= g.X_ID && (int)cst < g.X_ID + d.P) cst = (uint32_t)(g.X_ID + pidx);  // this pursuer's id cell
                    row[(size_t)NT * k] = cst;
                    if (k == 3) row[(size_t)NT * 4] = c >> 31;
                }
                row[(size_t)NT * 5] = (uint32_t)(q < NQ ? pidx : 0);
            }
        if (eligible) {
            const size_t wbytes = wh.size() * sizeof(uint32_t);
            e = hipMalloc(&h->wtables, wbytes);
            if (e == hipSuccess) e = hipMemcpy(h->wtables, wh.data(), wbytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                if (h->wtables) (void)hipFree(h->wtables);
                (void)hipFree(h->tables);
                delete h;
                return fail(MADRL_EHIP, "wave tables: %s", hipGetErrorString(e));
            }
            pw::WaveDev &w = h->wdev;
            memset(&w, 0, sizeof(w));
            w.n_catch = d.n_catch; w.surround = d.surround; w.reward_global = d.reward_global;
            w.sample_maps = d.sample_maps; w.n_maps = d.n_maps; w.max_steps = d.max_steps; w.auto_reset = d.auto_reset;
            w.max_opponents = d.max_opponents;
            w.fmap_stride = fstride;
            w.k0 = d.k0; w.k1 = d.k1; w.gid_base = d.gid_base;
            w.catchr = d.catchr; w.term_pursuit = d.term_pursuit; w.urgency = d.urgency; w.cw = d.cw;
            w.n_envs = d.n_envs;
            w.fmaps = reinterpret_cast<const uint32_t *>(h->wtables);
            w.vtab = d.vtab;
            w.codes = reinterpret_cast<const uint32_t *>(h->wtables) + w_codes;
            w.cnt_tmpl = reinterpret_cast<const uint32_t *>(h->wtables) + w_tmpl;
            w.slot_tab = reinterpret_cast<const uint32_t *>(h->wtables) + w_slots;
            w.state = d.state;
            w.flags = d.flags;
            h->zmask = (uint8_t *)state_dev + zmask_offset(d.rec_bytes, n_envs);  // caller-owned, like the records
            w.zmask = reinterpret_cast<uint32_t *>(h->zmask);
        } else {
            h->wave = nullptr;
        }
    }

    h->walk_mode = 0;
    h->max_blocks = 0;
    rc = madrl_pursuit_set_launch(h, 0, 0);
    // the fork / join events of madrl_pursuit_step_sharded, on this handle's device (hipSetDevice above), each checked on its own
    if (!rc && (hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess)) {
        if (h->ev_done) (void)hipEventDestroy(h->ev_done);
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        rc = fail(MADRL_EHIP, "create: hipEventCreateWithFlags failed");
    }
    if (rc) {
        if (h->wtables) (void)hipFree(h->wtables);
        (void)hipFree(h->tables);
        delete h;
        return rc;
    }
    *out = h;
    return MADRL_OK;
}

int madrl_pursuit_set_walk(madrl_pursuit *h, int32_t mode) {
    if (!h || mode < 0 || mode > 2) return fail(MADRL_EINVAL, "set_walk: mode must be 0 (auto), 1 (alternate) or 2 (forward)");
    h->walk_mode = mode;
    return MADRL_OK;
}

int madrl_pursuit_set_launch(madrl_pursuit *h, int32_t threads, int64_t max_blocks) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const int thr = pick_threads(h->dev, threads);
    if (thr % 64 != 0 || thr < 64 || thr > 1024) return fail(MADRL_EINVAL, "threads=%d must be a multiple of 64 in 64..1024", thr);
    const int nt = (h->dev.D + thr - 1) / thr;
    if (nt > 8) return fail(MADRL_EINVAL, "obs_dim=%d needs more than 8 slots per thread at %d threads", h->dev.D, thr);
    const size_t lds = lds_bytes_for(h->dev);
    if (lds > 160 * 1024) return fail(MADRL_EINVAL, "configuration needs %zu B of LDS (> 160 KiB)", lds);
    if (max_blocks < 0) return fail(MADRL_EINVAL, "max_blocks < 0");
    h->threads = thr;
    h->nt = nt;
    h->lds_bytes = lds;
    h->max_blocks = max_blocks;
    if (lds > 64 * 1024) {
        // opt in to large dynamic LDS for every instantiation we may launch
        const int bytes = (int)lds;
#define MADRL_SET_LDS(NT) hipFuncSetAttribute((const void *)pursuit_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
        (void)MADRL_SET_LDS(1); (void)MADRL_SET_LDS(2); (void)MADRL_SET_LDS(3); (void)MADRL_SET_LDS(4);
        (void)MADRL_SET_LDS(5); (void)MADRL_SET_LDS(6); (void)MADRL_SET_LDS(7); (void)MADRL_SET_LDS(8);
#undef MADRL_SET_LDS
    }
    return MADRL_OK;
}

void madrl_pursuit_destroy(madrl_pursuit *h) {
    if (!h) return;
    if (h->tables) (void)hipFree(h->tables);
    if (h->wtables) (void)hipFree(h->wtables);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    delete h;
}

int madrl_pursuit_set_kernel(madrl_pursuit *h, int32_t kind) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (kind != MADRL_KERNEL_AUTO && kind != MADRL_KERNEL_GENERIC && kind != MADRL_KERNEL_WAVE)
        return fail(MADRL_EINVAL, "unknown kernel kind %d", kind);
    if (kind == MADRL_KERNEL_WAVE && !h->wave)
        return fail(MADRL_EINVAL, "no one-wavefront-per-env specialisation was compiled for this configuration "
                    "(see madrl_amd/csrc/pursuit_specializations.def)");
    h->kernel_kind = kind;
    return MADRL_OK;
}

int madrl_pursuit_kernel_kind(const madrl_pursuit *h, int32_t *out_kind) {
    if (!h || !out_kind) return fail(MADRL_EINVAL, "NULL argument");
    *out_kind = use_wave(h) ? MADRL_KERNEL_WAVE : MADRL_KERNEL_GENERIC;
    return MADRL_OK;
}

int madrl_pursuit_reset(madrl_pursuit *h, const uint8_t *mask_dev, const int32_t *inj_pos_dev,
                        const int32_t *inj_map_dev, float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "reset: handle/obs is NULL");
    PursuitIO io;
    memset(&io, 0, sizeof(io));
    io.mask = mask_dev;
    io.inj_pos = inj_pos_dev;
    io.inj_map = inj_map_dev;
    io.obs = obs_dev;
    return launch(h, io, 0, stream);
}

int madrl_pursuit_step(madrl_pursuit *h, const int32_t *actions_dev, const int32_t *inj_evader_actions_dev,
                       float *obs_dev, float *rew_dev, uint8_t *done_dev, int32_t *removed_dev, void *stream) {
    if (!h || !actions_dev || !obs_dev || !rew_dev || !done_dev || !removed_dev)
        return fail(MADRL_EINVAL, "step: NULL argument");
    PursuitIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions_dev;
    io.inj_eact = inj_evader_actions_dev;
    io.obs = obs_dev;
    io.rew = rew_dev;
    io.done = done_dev;
    io.removed = removed_dev;
    return launch(h, io, 1, stream);
}

int madrl_pursuit_step_sharded(madrl_pursuit *const *hs, const madrl_pursuit_shard_io *io, int32_t n_shards, void *caller_stream,
                               int32_t fork, int32_t join) {
    if (!hs || !io || n_shards < 1) return fail(MADRL_EINVAL, "step_sharded: NULL argument or n_shards < 1");
    for (int j = 0; j < n_shards; ++j) {
        if (!hs[j] || !io[j].actions || !io[j].obs || !io[j].rew || !io[j].done || !io[j].removed) return fail(MADRL_EINVAL, "step_sharded: shard %d has a NULL argument", j);
        if (!hs[j]->ev_done || !hs[j]->ev_fork) return fail(MADRL_EINVAL, "step_sharded: shard %d's handle has no events (madrl_pursuit_create makes them)", j);
    }
    hipStream_t cs = (hipStream_t)caller_stream;
    if (fork) {   // every sub-batch stream waits for what the caller's stream holds so far (the actions)
        MADRL_HIP_TRY(hipEventRecord(hs[0]->ev_fork, cs));
        for (int j = 0; j < n_shards; ++j) if ((hipStream_t)io[j].stream != cs) MADRL_HIP_TRY(hipStreamWaitEvent((hipStream_t)io[j].stream, hs[0]->ev_fork, 0));
    }
    int rc = MADRL_OK, launched = 0;
    for (; launched < n_shards && rc == MADRL_OK; ++launched) {
        const int j = launched;
        PursuitIO p;
        memset(&p, 0, sizeof(p));
        p.actions = io[j].actions; p.inj_eact = io[j].inj_evader_actions; p.obs = io[j].obs; p.rew = io[j].rew; p.done = io[j].done; p.removed = io[j].removed;
        rc = launch(hs[j], p, 1, io[j].stream);
        if (rc) break;
    }
    // ... and the caller's stream waits for every sub-batch -- also when a launch failed half way: the shards launched before it are
    // running, and the caller's stream must not be left unordered against them
    if (join || rc != MADRL_OK)
        for (int j = 0; j < launched; ++j) {
            if ((hipStream_t)io[j].stream == cs) continue;
            if (hipEventRecord(hs[j]->ev_done, (hipStream_t)io[j].stream) != hipSuccess || hipStreamWaitEvent(cs, hs[j]->ev_done, 0) != hipSuccess) {
                if (rc == MADRL_OK) rc = fail(MADRL_EHIP, "step_sharded: joining shard %d failed", j);
            }
        }
    return rc;
}

int madrl_pursuit_get_state(madrl_pursuit *h, int32_t *pos_p, int32_t *pos_e, uint8_t *gone, uint8_t *term_p,
                            uint8_t *term_e, int32_t *map_id, uint32_t *tick, int32_t *t, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(pursuit_get_state_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, pos_p,
                       pos_e, gone, term_p, term_e, map_id, tick, t);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_pursuit_set_state(madrl_pursuit *h, const int32_t *pos_p, const int32_t *pos_e, const uint8_t *gone,
                            const uint8_t *term_p, const uint8_t *term_e, const int32_t *map_id,
                            const uint32_t *tick, const int32_t *t, void *stream) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    const unsigned blocks = (unsigned)((h->dev.n_envs + 127) / 128);
    hipLaunchKernelGGL(pursuit_set_state_kernel, dim3(blocks), dim3(128), 0, (hipStream_t)stream, h->dev, pos_p,
                       pos_e, gone, term_p, term_e, map_id, tick, t);
    MADRL_HIP_TRY(hipGetLastError());
    // agents put somewhere else usually come with an observation buffer written from outside (a checkpoint restored): forget what is known
    // about it.  Episode clocks and RNG ticks alone (t, tick) say nothing about the buffer.
    if (pos_p || pos_e || gone || term_p || term_e || map_id) h->zmask_obs = nullptr;
    return MADRL_OK;
}

int madrl_pursuit_invalidate_obs(madrl_pursuit *h) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    h->zmask_obs = nullptr;  // the next fast-path launch starts from "no cell is known to be zero"
    return MADRL_OK;
}

int madrl_pursuit_declare_obs_zero(madrl_pursuit *h, const float *obs_dev, void *stream) {
    if (!h || !obs_dev) return fail(MADRL_EINVAL, "declare_obs_zero: NULL argument");
    if (h->zmask) {   // every cell of that buffer is known to hold +0.0f: no stale cell can need protecting
        MADRL_HIP_TRY(hipMemsetAsync(h->zmask, 0, (size_t)h->dev.n_envs * 256 * h->wave->g.waves * h->wave->g.mwords, (hipStream_t)stream));
        h->zmask_obs = obs_dev;
    }
    return MADRL_OK;
}

int madrl_pursuit_set_params(madrl_pursuit *h, double catchr, double constraint_window) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    if (!(constraint_window > 0.0 && constraint_window <= 1.0)) return fail(MADRL_EINVAL, "constraint_window must be in (0,1]");
    h->cfg.catchr = catchr; h->cfg.constraint_window = constraint_window;
    h->dev.catchr = catchr; h->dev.cw = constraint_window;
    return MADRL_OK;
}

int madrl_pursuit_set_curriculum(madrl_pursuit *h, const double *constraint_window_dev, const double *catchr_dev) {
    if (!h) return fail(MADRL_EINVAL, "handle is NULL");
    h->dev.cw_env = constraint_window_dev;
    h->dev.catchr_env = catchr_dev;
    return MADRL_OK;
}

}  // extern "C"
