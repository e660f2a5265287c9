// wrappers.hip -- the reference's env wrappers as elementwise epilogue kernels over the batched
// observation / reward tensors.  Each env instance keeps its own wrapper
// state, exactly as if N wrapped reference envs ran side by side.
//
// Reference (file:line in /root/reference/madrl_environments/__init__.py):
//   ObservationBuffer.step/reset :176-195 ; StandardizedEnv.update_*_estimate / standardize_* :242-271,
//   step :283-291 ; DiagnosticsWrapper.step :335-369, _discount_sum :392-393.
// HBM-bound streaming kernels: grid-stride, one element per lane, running statistics in float64
// like the reference (a float32 EMA with alpha = 1e-3 drifts by ~6e-5, beyond the 1e-5 tolerance).
#include "common.hpp"

namespace {
using namespace madrl;

__global__ void obsnorm_kernel(const float *obs_in, double *mean, double *var, float *obs_out, int64_t n,
                               int64_t per_env, const uint8_t *mask, double alpha, double eps) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask != nullptr && mask[i / per_env] == 0) continue;
        const double x = (double)obs_in[i];
        const double m = (1.0 - alpha) * mean[i] + alpha * x;              // :245-246
        const double d = x - m;
        const double v = (1.0 - alpha) * var[i] + alpha * (d * d);         // :247-249
        mean[i] = m;
        var[i] = v;
        obs_out[i] = (float)((x - m) / (sqrt(v) + eps));                   // :262-263
    }
}

// The same arithmetic on two elements per lane as 16-byte words (float64 pairs of mean and variance, an 8-byte pair of inputs and
// outputs), every access non-temporal (each byte is touched once per step), and the next iteration's loads issued BEFORE this
// iteration's stores: vmcnt retires loads and stores in issue order, so loads queued behind a batch of stores wait for those to be
// acknowledged.  scripts/ubench/stats_stream.hip at the Waterworld C3 size (34.9 M elements, 44 bytes each): 245.8 -> 224.5 us =
// 6.2 -> 6.8 TB/s.
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void obsnorm_one(double &m, double &v, float x, float &o, double alpha, double eps) {
    const double xd = (double)x;
    m = (1.0 - alpha) * m + alpha * xd;                                    // :245-246
    const double d = xd - m;
    v = (1.0 - alpha) * v + alpha * (d * d);                               // :247-249
    o = (float)((xd - m) / (sqrt(v) + eps));                               // :262-263
}
__global__ __launch_bounds__(256) void obsnorm_pairs_kernel(const float *obs_in, double *mean, double *var, float *obs_out, int64_t n_pairs,
                                                            double alpha, double eps) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    d2_t m = {0.0, 0.0}, v = {1.0, 1.0};
    f2_t x = {0.0f, 0.0f};
    if (p < n_pairs) {
        m = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(mean) + p);
        v = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(var) + p);
        x = __builtin_nontemporal_load(reinterpret_cast<const f2_t *>(obs_in) + p);
    }
    while (p < n_pairs) {
        const int64_t q = p + stride;
        d2_t m2 = {0.0, 0.0}, v2 = {1.0, 1.0};
        f2_t x2 = {0.0f, 0.0f};
        if (q < n_pairs) {
            m2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(mean) + q);
            v2 = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(var) + q);
            x2 = __builtin_nontemporal_load(reinterpret_cast<const f2_t *>(obs_in) + q);
        }
        double m0 = m.x, v0 = v.x, m1 = m.y, v1 = v.y;
        f2_t o;
        float o0, o1;
        obsnorm_one(m0, v0, x.x, o0, alpha, eps);
        obsnorm_one(m1, v1, x.y, o1, alpha, eps);
        m.x = m0; m.y = m1; v.x = v0; v.y = v1; o.x = o0; o.y = o1;
        __builtin_nontemporal_store(m, reinterpret_cast<d2_t *>(mean) + p);
        __builtin_nontemporal_store(v, reinterpret_cast<d2_t *>(var) + p);
        __builtin_nontemporal_store(o, reinterpret_cast<f2_t *>(obs_out) + p);
        m = m2; v = v2; x = x2; p = q;
    }
}

__global__ void rewnorm_kernel(const float *rew_in, double *mean, double *var, float *rew_out, int64_t n, int64_t per_env,
                               const uint8_t *mask, double alpha, double eps, double scale, int enable) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask != nullptr && mask[i / per_env] == 0) continue;
        double r = (double)rew_in[i];
        if (enable) {
            const double m = (1.0 - alpha) * mean[i] + alpha * r;          // :253-254
            const double d = r - m;
            const double v = (1.0 - alpha) * var[i] + alpha * (d * d);     // :255-257
            mean[i] = m;
            var[i] = v;
            r = r / (sqrt(v) + eps);                                       // :268-271 (the mean is not subtracted)
        }
        rew_out[i] = (float)(scale * r);                                   // :290
    }
}

__global__ void obsbuffer_kernel(const float *obs, float *buf, int64_t n, int64_t per_env, int k, const uint8_t *reset_mask,
                                 const uint8_t *active_mask) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (active_mask != nullptr && active_mask[i / per_env] == 0) continue;  // partial reset: the other envs keep their history
        float *b = buf + i * k;
        const float x = obs[i];
        if (reset_mask != nullptr && (reset_mask[i / per_env] & 0x7Fu) != 0) {   // a done byte's bit 7 (overflow report) resets nothing
            for (int j = 0; j < k; ++j) b[j] = x;                           // reset :190-192
        } else {
            for (int j = 0; j + 1 < k; ++j) b[j] = b[j + 1];               // step :179-181
            b[k - 1] = x;
        }
    }
}

// k == 4: an element's history is one 16-byte word
__global__ void obsbuffer4_kernel(const float *obs, float4 *buf, int64_t n, int64_t per_env, const uint8_t *reset_mask,
                                  const uint8_t *active_mask) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (active_mask != nullptr && active_mask[i / per_env] == 0) continue;
        const float x = obs[i];
        float4 b;
        if (reset_mask != nullptr && (reset_mask[i / per_env] & 0x7Fu) != 0) { b.x = x; b.y = x; b.z = x; b.w = x; }   // reset :190-192
        else { const float4 o = buf[i]; b.x = o.y; b.y = o.z; b.z = o.w; b.w = x; }                           // step :179-181
        buf[i] = b;
    }
}

__global__ void diagnostics_kernel(const float *rew, const uint8_t *done, double *ep_reward, int32_t *ep_len, double *disc_ret,
                                   double *disc_pow, int64_t N, int A, double discount, int max_traj_len, double *out_ep_reward,
                                   double *out_disc, int32_t *out_len, uint8_t *out_finished) {
    const int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int a = 0; a < A; ++a) {
        const double r = (double)rew[n * A + a];
        ep_reward[n * A + a] += r;                                          // :350
        s += r;
    }
    if (ep_len[n] == 0) disc_pow[n] = 1.0;
    disc_ret[n] += (s / (double)A) * disc_pow[n];                           // :360-361 accumulated step by step
    disc_pow[n] *= discount;
    ep_len[n] += 1;                                                         // :351
    // bits 0 / 1 of a done byte end an episode (terminal, time limit); bit 7 is the sticky "a capacity overflowed" report of the
    // Pursuit / MultiWalker kernels (include/madrl_hip.h) and starts no episode by itself
    const bool fin = ((done[n] & 0x03u) != 0) || ep_len[n] >= max_traj_len;   // :354
    out_finished[n] = fin ? 1 : 0;
    if (fin) {
        for (int a = 0; a < A; ++a) { out_ep_reward[n * A + a] = ep_reward[n * A + a]; ep_reward[n * A + a] = 0.0; }
        out_disc[n] = disc_ret[n];
        out_len[n] = ep_len[n];
        disc_ret[n] = 0.0;
        ep_len[n] = 0;                                                      // :365-367
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}
}  // namespace

// One thread per (env, agent) column; each time row is one coalesced sweep over [N][A].  HBM-bound:
// 4 (rew) + 4 (values) + 4 + 4 (returns, adv) bytes per element and step, the done byte is shared by A lanes.
__global__ void gae_kernel(const float *__restrict__ rew, const uint8_t *__restrict__ done, const float *__restrict__ values, int64_t T,
                           int64_t n_envs, int n_agents, double gamma, double lambda, float *__restrict__ returns,
                           float *__restrict__ adv) {
    const int64_t row = n_envs * n_agents;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row) return;
    const int64_t n = i / n_agents;
    double ret = values ? (double)values[T * row + i] : 0.0, a = 0.0;
    double vnext = ret;
    for (int64_t t = T - 1; t >= 0; --t) {
        const bool cut = (done[t * n_envs + n] & 0x03u) != 0;   // (bit 7 = overflow report: no episode boundary)
        const double r = (double)rew[t * row + i];
        ret = r + (cut ? 0.0 : gamma * ret);
        returns[t * row + i] = (float)ret;
        if (adv) {
            const double v = (double)values[t * row + i];
            const double delta = r + (cut ? 0.0 : gamma * vnext) - v;
            a = delta + (cut ? 0.0 : gamma * lambda * a);
            adv[t * row + i] = (float)a;
            vnext = v;
        }
    }
}

extern "C" {

int madrl_wrap_obsnorm(const float *obs_in, double *mean, double *var, float *obs_out, int64_t n_elems, int64_t elems_per_env,
                       const uint8_t *mask, double alpha, double eps, void *stream) {
    if (!obs_in || !mean || !var || !obs_out || n_elems < 1 || elems_per_env < 1) return fail(MADRL_EINVAL, "obsnorm: bad argument");
    // (round 2 measured two elements per lane with 16-byte accesses alone: no gain.  What counts is issuing the next loads before this
    // iteration's stores, and non-temporal accesses -- obsnorm_pairs_kernel.)
    const bool aligned = ((uintptr_t)obs_in % 8 == 0) && ((uintptr_t)obs_out % 8 == 0) && ((uintptr_t)mean % 16 == 0) && ((uintptr_t)var % 16 == 0);
    if (mask == nullptr && aligned && n_elems >= 2) {
        const int64_t n_pairs = n_elems / 2;
        int64_t blocks = (n_pairs + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(obsnorm_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, obs_in, mean, var, obs_out, n_pairs, alpha, eps);
        if (n_elems & 1)   // the odd last element
            hipLaunchKernelGGL(obsnorm_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, obs_in + (n_elems - 1), mean + (n_elems - 1), var + (n_elems - 1),
                               obs_out + (n_elems - 1), (int64_t)1, (int64_t)1, (const uint8_t *)nullptr, alpha, eps);
    } else
        hipLaunchKernelGGL(obsnorm_kernel, dim3(grid_for(n_elems)), dim3(256), 0, (hipStream_t)stream, obs_in, mean, var, obs_out,
                           n_elems, elems_per_env, mask, alpha, eps);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_wrap_rewnorm(const float *rew_in, double *mean, double *var, float *rew_out, int64_t n, int64_t per_env,
                       const uint8_t *mask, double alpha, double eps, double scale, int32_t enable_norm, void *stream) {
    if (!rew_in || !rew_out || n < 1 || per_env < 1 || (enable_norm && (!mean || !var))) return fail(MADRL_EINVAL, "rewnorm: bad argument");
    hipLaunchKernelGGL(rewnorm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, rew_in, mean, var, rew_out, n, per_env,
                       mask, alpha, eps, scale, (int)enable_norm);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_wrap_obsbuffer(const float *obs, float *buf, int64_t n_elems, int64_t elems_per_env, int32_t k, const uint8_t *reset_mask,
                         const uint8_t *active_mask, void *stream) {
    if (!obs || !buf || n_elems < 1 || elems_per_env < 1 || k < 1) return fail(MADRL_EINVAL, "obsbuffer: bad argument");
    if (k == 4 && (uintptr_t)buf % 16 == 0)
        hipLaunchKernelGGL(obsbuffer4_kernel, dim3(grid_for(n_elems)), dim3(256), 0, (hipStream_t)stream, obs, (float4 *)buf, n_elems,
                           elems_per_env, reset_mask, active_mask);
    else
        hipLaunchKernelGGL(obsbuffer_kernel, dim3(grid_for(n_elems)), dim3(256), 0, (hipStream_t)stream, obs, buf, n_elems,
                           elems_per_env, (int)k, reset_mask, active_mask);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_wrap_diagnostics(const float *rew, const uint8_t *done, double *ep_reward, int32_t *ep_len, double *disc_ret,
                           double *disc_pow, int64_t n_envs, int32_t n_agents, double discount, int32_t max_traj_len,
                           double *out_ep_reward, double *out_disc, int32_t *out_len, uint8_t *out_finished, void *stream) {
    if (!rew || !done || !ep_reward || !ep_len || !disc_ret || !disc_pow || !out_ep_reward || !out_disc || !out_len || !out_finished ||
        n_envs < 1 || n_agents < 1)
        return fail(MADRL_EINVAL, "diagnostics: bad argument");
    hipLaunchKernelGGL(diagnostics_kernel, dim3((unsigned)((n_envs + 127) / 128)), dim3(128), 0, (hipStream_t)stream, rew, done,
                       ep_reward, ep_len, disc_ret, disc_pow, n_envs, (int)n_agents, discount, (int)max_traj_len, out_ep_reward,
                       out_disc, out_len, out_finished);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

int madrl_rollout_gae(const float *rew, const uint8_t *done, const float *values, int64_t T, int64_t n_envs, int32_t n_agents,
                      double gamma, double lambda, float *returns, float *adv, void *stream) {
    if (!rew || !done || !returns || T < 1 || n_envs < 1 || n_agents < 1 || (adv && !values)) return fail(MADRL_EINVAL, "gae: bad argument");
    hipLaunchKernelGGL(gae_kernel, dim3(grid_for(n_envs * n_agents)), dim3(256), 0, (hipStream_t)stream, rew, done, values, T, n_envs,
                       (int)n_agents, gamma, lambda, returns, adv);
    MADRL_HIP_TRY(hipGetLastError());
    return MADRL_OK;
}

}  // extern "C"
