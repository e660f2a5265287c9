// Hand-written policies on the device (reference: heuristics/pursuit.py:13-56, heuristics/waterworld.py:6-62,
// heuristics/multi_walker.py:10-86): pure functions of one agent's observation row, one row per lane group, reading the
// observation tensors the step kernels just wrote (no host round trip between step and act).
#include "common.hpp"

namespace madrl {
namespace {

enum : uint32_t { TAG_HEURISTIC_ACT = 3 };

// ---- PursuitHeuristicPolicy.sample_actions (pursuit.py:18-54): nearest evader in the window (first one in row-major
// order among equals), then a direction from atan2 quadrants.  The quadrant logic is a table over the evader's window cell,
// built on the host with the reference's own float64 expression (madrl_amd/heuristics.py); 255 in the table or an empty
// window = the reference's action_space.sample(), here Philox(row id, tick).
// 16 lanes per row.  Generic form (any obs_range, either layout): lane s looks at window cells s, s + 16, ... of the evader channel (four
// independent loads in flight, then the comparisons); the packed (distance^2, cell) keys are min-reduced inside the DPP row of 16 lanes.
// A block walks the rows in strides of the grid: few, long-lived wavefronts.  After its last row a launch advances the draw counter
// itself (the last workgroup to retire, policy_retire), so a rollout step is two kernels, not three.
// Measured for 524 288 rows (kernel under rocprofv3 / rollout step of scripts/rollout_bench.py): round 4's mapping -- 8 lanes per row, a
// lane per window row, load - compare - branch per cell -- 44 us; the same with its loads batched: 133 us per rollout step; 16 lanes per
// row, one short-lived wavefront per four rows: 56 / 132; the generic form below with a separate counter kernel behind it: 51 / 125 (two
// sub-batches: 119); pursuit_policy_rows_kernel: 28.4 us / 99 - 102 per rollout step.
__device__ __forceinline__ uint32_t min_row16(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// the draw counter of this launch, read once per wavefront: the device word (it survives hipGraph replay) cannot move before every workgroup
// of the launch has retired (policy_retire), so a plain cached load is enough -- a system-scope atomic load per row with an empty window
// went to memory uncached and doubled the kernel's time
__device__ __forceinline__ uint32_t policy_tick(uint32_t tick, const uint32_t *tick_dev) { return tick + (tick_dev ? *tick_dev : 0u); }

__device__ __forceinline__ void policy_emit(uint32_t key, int64_t row, const uint8_t *__restrict__ table, uint32_t k0, uint32_t k1,
                                            int64_t row_id_base, uint32_t tk, int32_t *__restrict__ actions) {
    int act = 255;
    if (key != 0xFFFFFFFFu) act = table[key & 0xFFFFu];
    if (act == 255) {
        const uint64_t id = (uint64_t)(row_id_base + row);
        const u32x4 r = philox4x32_10((uint32_t)id, tk, (uint32_t)(id >> 32), TAG_HEURISTIC_ACT, k0, k1);
        act = (int)__umulhi(r.x, 5u);
    }
    actions[row] = act;
}

// every row of this workgroup is done: the last workgroup of the launch advances the counter.  One shared count would be gridDim.x
// same-address atomics arriving together at the end of a launch of persistent workgroups -- measured 8.5 ns each, one after the other: +17 us
// with 2 048 workgroups, +35 us with 4 096 -- so the workgroups count in 64 groups on 64 cache lines, and the last one of each group counts
// the groups: two chains of <= 64.  Layout of tick_dev (MADRL_POLICY_COUNTER_WORDS uint32): [0] draw counter, [1] groups done, [32 (1 + g)]
// workgroups of group g done; all but [0] are zero again when the launch ends.
__device__ __forceinline__ void policy_retire(uint32_t *tick_dev) {
    if (!tick_dev) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // relaxed on purpose: an agent-scope release / acquire here is an L2 write-back + invalidate per workgroup (80 us per launch instead
        // of 30, and the step kernel on the other stream loses its cache with it); nothing but the counters is ordered by them, and a
        // workgroup has read tick_dev[0] (policy_tick) long before it arrives here
        const uint32_t G = gridDim.x < 64u ? gridDim.x : 64u, g = blockIdx.x % G, n_g = (gridDim.x - 1u - g) / G + 1u;
        uint32_t *mine = tick_dev + 32u * (1u + g);
        if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_g - 1u) {
            __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(tick_dev + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u) {
                __hip_atomic_store(tick_dev + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(tick_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(256) void pursuit_policy_kernel(const float *__restrict__ obs, int64_t n_rows, int R, int64_t row_stride,
                                                             int cell_stride, int ch_offset, const uint8_t *__restrict__ table,
                                                             uint32_t k0, uint32_t k1, int64_t row_id_base, uint32_t tick,
                                                             uint32_t *tick_dev, int32_t *__restrict__ actions) {
    const int sub = threadIdx.x & 15;
    const int c = R / 2;  // :23 (Python 2 integer division)
    const int cells = R * R;
    const uint32_t tk = policy_tick(tick, tick_dev);
    for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < n_rows; row += (int64_t)gridDim.x * 16) {
        uint32_t key = 0xFFFFFFFFu;
        const float *o = obs + row * row_stride + ch_offset;
        for (int kb = sub; kb < cells; kb += 64) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = kb + 16 * q < cells ? o[(int64_t)(kb + 16 * q) * cell_stride] : 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (v[q] != 0.0f) {
                    const int k = kb + 16 * q, i = k / R, j = k - i * R;
                    const uint32_t d2 = (uint32_t)((i - c) * (i - c) + (j - c) * (j - c));
                    key = min(key, (d2 << 16) | (uint32_t)k);
                }
        }
        key = min_row16(key);
        if (sub == 0) policy_emit(key, row, table, k0, k1, row_id_base, tk, actions);
    }
    policy_retire(tick_dev);
}

// Flatten rows with a window of 4 ... 64 cells (obs_range 3, 5, 7): the evader channel of a row is 4 * R * R consecutive bytes, lane s of
// the row's 16 takes cells 4 s ... 4 s + 3 in ONE 16-byte load (4-byte aligned: rows are 12 R^2 + 4 bytes apart; the last lanes step back
// to the final four cells, duplicates do not change a minimum); a lane's four (distance^2, cell) keys do not depend on the row and are
// computed once.  U rows per lane group in flight, lane u of a group finishes row u.  Measured for the 524 288 rows of configs[1]: 28.0 - 29.1 us
// with U = 2, 4, 8 and 4 or 8 blocks per CU alike (38.4 only with U = 2 and 4 blocks) -- not latency: the rows' 196 useful bytes at a
// 592-byte stride touch 168 MB of 128-byte lines (PMC FETCH_SIZE), i.e. 5.9 TB/s; what is left is the reference's row layout.
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };

template <int U>
__global__ __launch_bounds__(256) void pursuit_policy_rows_kernel(const float *__restrict__ obs, int64_t n_rows, int R, int64_t row_stride,
                                                                  int ch_offset, const uint8_t *__restrict__ table, uint32_t k0, uint32_t k1,
                                                                  int64_t row_id_base, uint32_t tick, uint32_t *tick_dev,
                                                                  int32_t *__restrict__ actions) {
    const int sub = threadIdx.x & 15;
    const int c = R / 2, cells = R * R;
    const uint32_t tk = policy_tick(tick, tick_dev);
    const int base = min(4 * sub, cells - 4);
    uint32_t keyq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = base + q, i = k / R, j = k - i * R;
        keyq[q] = ((uint32_t)((i - c) * (i - c) + (j - c) * (j - c)) << 16) | (uint32_t)k;
    }
    const int64_t stride = (int64_t)gridDim.x * 16;
    const float *o = obs + ch_offset + base;
    for (int64_t row0 = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row0 < n_rows; row0 += U * stride) {
        f4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {   // all U loads in flight before the first comparison
            const int64_t r = row0 + u * stride;
            v[u] = *reinterpret_cast<const f4u *>(o + (r < n_rows ? r : row0) * row_stride);
        }
        uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t k = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < 4; ++q) k = min(k, v[u].v[q] != 0.0f ? keyq[q] : 0xFFFFFFFFu);
            k = min_row16(k);
            mine = sub == u ? k : mine;   // lane u of the group finishes row u
        }
        const int64_t my_row = row0 + sub * stride;
        if (sub < U && my_row < n_rows) policy_emit(mine, my_row, table, k0, k1, row_id_base, tk, actions);
    }
    policy_retire(tick_dev);
}

// ---- WaterworldHeuristicPolicy.sample_actions (waterworld.py:11-58), one row per thread group of 8 lanes: each lane sums
// its sensors' contributions in float64, shuffles reduce.  K = D / 7 (:26), blocks 0 (obstacle, repulsive), 1 (evader,
// attractive), 3 (poison, repulsive), 5 (allies, attractive / 2); *1.5 when colliding (:43-44); unit norm (:46-50).
__global__ __launch_bounds__(256) void waterworld_policy_kernel(const float *__restrict__ obs, int64_t n_rows, int D,
                                                                const double *__restrict__ cs, float *__restrict__ actions) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const int K = D / 7;
    double ax[4] = {0, 0, 0, 0}, ay[4] = {0, 0, 0, 0};
    const float *o = obs + (row < n_rows ? row : 0) * (int64_t)D;
    for (int k = sub; k < K; k += 8) {
        const double cx = cs[2 * k], sy = cs[2 * k + 1];
        const double v0 = o[k], v1 = o[K + k], v3 = o[3 * K + k], v5 = o[5 * K + k];
        ax[0] += v0 * cx; ay[0] += v0 * sy;
        ax[1] += v1 * cx; ay[1] += v1 * sy;
        ax[2] += v3 * cx; ay[2] += v3 * sy;
        ax[3] += v5 * cx; ay[3] += v5 * sy;
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ax[q] += __shfl_xor(ax[q], m, 8);
            ay[q] += __shfl_xor(ay[q], m, 8);
        }
    if (row < n_rows && sub == 0) {
        const double fe = o[7 * K] > 0.0f ? 1.5 : 1.0, fp = o[7 * K + 1] > 0.0f ? 1.5 : 1.0;
        double x = -ax[0] + ax[1] * fe - ax[2] * fp + ax[3] / 2;
        double y = -ay[0] + ay[1] * fe - ay[2] * fp + ay[3] / 2;
        const double n = sqrt(x * x + y * y);
        if (n > 0) { x /= n; y /= n; } else { x = 0; y = 0; }
        actions[2 * row] = (float)x;
        actions[2 * row + 1] = (float)y;
    }
}

// ---- MultiWalkerHeuristicPolicy.sample_actions (multi_walker.py:16-86), one row per thread.  The gait state machine is
// re-initialised on every call in the reference (:23-25: state = STAY_ON_ONE_LEG, moving_leg = 0), so the policy is a
// pure function of the observation; `if target:` treats a 0.0 target as unset (:60-67).
__global__ __launch_bounds__(256) void multiwalker_policy_kernel(const float *__restrict__ obs, int64_t n_rows, int D,
                                                                 float *__restrict__ actions) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    const float *o = obs + row * (int64_t)D;
    double s[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) s[i] = (double)o[i];
    const double SPEED = 0.29, SKA = 0.1;
    // moving leg 0 (s[4..8]), supporting leg 1 (s[9..13])
    double hip_t0 = 1.1, knee_t0 = -0.6, hip_t1 = 0.0, knee_t1;
    bool has_hip1 = false;
    double ska = SKA + 0.03;
    if (s[2] > SPEED) ska += 0.03;
    ska = fmin(ska, SKA);
    knee_t1 = ska;
    int state = 1;
    if (s[9] < 0.10) state = 2;
    if (state == 2) {
        hip_t0 = 0.1; knee_t0 = SKA; knee_t1 = ska;
        if (s[8] != 0.0) { state = 3; ska = fmin(s[6], SKA); }
    }
    if (state == 3) { knee_t0 = ska; knee_t1 = 1.0; }
    double hip0 = 0, hip1 = 0, knee0 = 0, knee1 = 0;
    if (hip_t0 != 0.0) hip0 = 0.9 * (hip_t0 - s[4]) - 0.25 * s[5];
    if (has_hip1 && hip_t1 != 0.0) hip1 = 0.9 * (hip_t1 - s[9]) - 0.25 * s[10];
    if (knee_t0 != 0.0) knee0 = 4.0 * (knee_t0 - s[6]) - 0.25 * s[7];
    if (knee_t1 != 0.0) knee1 = 4.0 * (knee_t1 - s[11]) - 0.25 * s[12];
    const double head = 0.9 * (0 - s[0]) - 1.5 * s[1];
    hip0 -= head; hip1 -= head;
    knee0 -= 15.0 * s[3]; knee1 -= 15.0 * s[3];
    auto clip = [](double v) { return (float)fmax(-1.0, fmin(1.0, 0.5 * v)); };
    float *a = actions + 4 * row;
    a[0] = clip(hip0); a[1] = clip(knee0); a[2] = clip(hip1); a[3] = clip(knee1);
}

}  // namespace
}  // namespace madrl

using namespace madrl;

extern "C" {

int madrl_heuristic_pursuit(const float *obs, int64_t n_rows, int32_t obs_range, int64_t row_stride, int32_t cell_stride,
                            int32_t ch_offset, const uint8_t *table_dev, uint64_t seed, int64_t row_id_base, uint32_t tick,
                            uint32_t *tick_dev, int32_t *actions, void *stream) {
    if (!obs || !table_dev || !actions || n_rows < 1 || obs_range < 1 || obs_range > 255) return fail(MADRL_EINVAL, "heuristic_pursuit: bad argument");
    const int cells = obs_range * obs_range;
    if (cell_stride == 1 && cells >= 4 && cells <= 64) {
        constexpr int U = 2;
        int64_t blocks = (n_rows + 16 * U - 1) / (16 * U);
        if (blocks > 256 * 8) blocks = 256 * 8;   // 8 four-wavefront blocks per CU: every wavefront resident at once
        hipLaunchKernelGGL(pursuit_policy_rows_kernel<U>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs, n_rows, (int)obs_range,
                           row_stride, (int)ch_offset, table_dev, (uint32_t)seed, (uint32_t)(seed >> 32), row_id_base, tick, tick_dev, actions);
    } else {
        int64_t blocks = (n_rows + 15) / 16;
        if (blocks > 256 * 16) blocks = 256 * 16;   // 16 four-wavefront blocks per CU: two rounds of resident wavefronts
        hipLaunchKernelGGL(pursuit_policy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs, n_rows,
                           (int)obs_range, row_stride, (int)cell_stride, (int)ch_offset, table_dev, (uint32_t)seed, (uint32_t)(seed >> 32),
                           row_id_base, tick, tick_dev, actions);
    }
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_heuristic_waterworld(const float *obs, int64_t n_rows, int32_t obs_dim, const double *cos_sin_dev, float *actions, void *stream) {
    if (!obs || !cos_sin_dev || !actions || n_rows < 1 || obs_dim < 9) return fail(MADRL_EINVAL, "heuristic_waterworld: bad argument");
    const int64_t threads = n_rows * 8;
    hipLaunchKernelGGL(waterworld_policy_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, n_rows,
                       (int)obs_dim, cos_sin_dev, actions);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_heuristic_multiwalker(const float *obs, int64_t n_rows, int32_t obs_dim, float *actions, void *stream) {
    if (!obs || !actions || n_rows < 1 || obs_dim < 14) return fail(MADRL_EINVAL, "heuristic_multiwalker: bad argument");
    hipLaunchKernelGGL(multiwalker_policy_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs, n_rows,
                       (int)obs_dim, actions);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // extern "C"
