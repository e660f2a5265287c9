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
// 16 lanes per row: lane s looks at window cells s, s + 16, ... of the evader channel (four independent loads in flight, then the
// comparisons), so the 16 lanes of a row read 64 consecutive bytes per pass (flatten rows; the (R, R, 4) layout strides by its four
// channels); the packed (distance^2, cell) keys are min-reduced by shuffles.  A block walks the rows in strides of the grid: few,
// long-lived wavefronts.  The kernel reads one channel of every row -- a third of the bytes -- but nowhere near a third of the time of a
// full pass over the buffer (~50 us for its 310 MB at 65 536 envs).  Measured for those 524 288 rows (kernel under rocprofv3 / rollout
// step of scripts/rollout_bench.py): round 4's mapping -- 8 lanes per row, a lane per window row, load - compare - branch per cell --
// 44 us; the same with its loads batched: 133 us per rollout step; 16 lanes per row, one short-lived wavefront per four rows: 56 / 132;
// this form: 51 / 125 (two sub-batches: 119).
__global__ __launch_bounds__(256) void pursuit_policy_kernel(const float *__restrict__ obs, int64_t n_rows, int R, int64_t row_stride,
                                                             int cell_stride, int ch_offset, const uint8_t *__restrict__ table,
                                                             uint32_t k0, uint32_t k1, int64_t row_id_base, uint32_t tick,
                                                             const uint32_t *__restrict__ tick_dev, int32_t *__restrict__ actions) {
    const int sub = threadIdx.x & 15;
    const int c = R / 2;  // :23 (Python 2 integer division)
    const int cells = R * R;
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
        key = min(key, (uint32_t)__shfl_xor((int)key, 8, 16));
        key = min(key, (uint32_t)__shfl_xor((int)key, 1, 16));
        key = min(key, (uint32_t)__shfl_xor((int)key, 2, 16));
        key = min(key, (uint32_t)__shfl_xor((int)key, 4, 16));
        if (sub == 0) {
            int act = 255;
            if (key != 0xFFFFFFFFu) act = table[key & 0xFFFFu];
            if (act == 255) {
                const uint64_t id = (uint64_t)(row_id_base + row);
                const uint32_t tk = tick + (tick_dev ? *tick_dev : 0u);  // device counter: survives hipGraph replay
                const u32x4 r = philox4x32_10((uint32_t)id, tk, (uint32_t)(id >> 32), TAG_HEURISTIC_ACT, k0, k1);
                act = (int)__umulhi(r.x, 5u);
            }
            actions[row] = act;
        }
    }
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
                            const uint32_t *tick_dev, int32_t *actions, void *stream) {
    if (!obs || !table_dev || !actions || n_rows < 1 || obs_range < 1 || obs_range > 255) return fail(MADRL_EINVAL, "heuristic_pursuit: bad argument");
    int64_t blocks = (n_rows + 15) / 16;
    if (blocks > 256 * 16) blocks = 256 * 16;   // 16 four-wavefront blocks per CU: two rounds of resident wavefronts
    hipLaunchKernelGGL(pursuit_policy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs, n_rows,
                       (int)obs_range, row_stride, (int)cell_stride, (int)ch_offset, table_dev, (uint32_t)seed, (uint32_t)(seed >> 32),
                       row_id_base, tick, tick_dev, actions);
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
