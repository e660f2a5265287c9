// Micro-benchmark behind DESIGN.md 4e: how fast can the StandardizedEnv statistics stream run?  Per observation element a float64
// running mean and variance are read and written (32 bytes) and one float32 leaves (4 bytes); the arithmetic is the wrapper's
// (madrl_environments/__init__.py:242-263: two EMAs, a float64 square root and a division).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/stats_stream.hip -o scripts/ubench/stats_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int ROW = 1065;   // Waterworld C3: 5 pursuers x 213 elements per env
typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float xval(int env, int e) { return (float)((env * 131 + e * 7) & 1023) * (1.0f / 1024.0f); }
__device__ __forceinline__ void one(double &m, double &v, float x, float &o, double alpha, double eps) {
    const double xd = (double)x;
    m = (1.0 - alpha) * m + alpha * xd;
    const double d = xd - m;
    v = (1.0 - alpha) * v + alpha * (d * d);
    o = (float)((xd - m) / (sqrt(v) + eps));
}

// one wavefront per env row, persistent (the fused kernel's shape)
// V 0: batches of 4 elements per lane, loads -> compute -> stores, batch after batch (the round-3 code)
// V 1: the same batches, software-pipelined: the next batch's loads are issued BEFORE this batch's stores
// V 2: pipelined, 16-byte pairs (two consecutive elements per lane), 2 pairs per batch
// V 3: like 1 but plain (cached) accesses instead of non-temporal ones
template <int V>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5)))
void row_kernel(double *mean, double *var, float *out, int n_envs, double alpha, double eps) {
    const int lane = threadIdx.x;
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        const size_t base = (size_t)env * ROW;
        double *gm = mean + base, *gv = var + base;
        float *go = out + base;
        if (V == 0) {
            for (int e0 = lane; e0 < ROW; e0 += 256) {
                double m[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int e = e0 + 64 * u; m[u] = e < ROW ? __builtin_nontemporal_load(&gm[e]) : 0.0; v[u] = e < ROW ? __builtin_nontemporal_load(&gv[e]) : 1.0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = e0 + 64 * u;
                    if (e < ROW) { float o; one(m[u], v[u], xval(env, e), o, alpha, eps); __builtin_nontemporal_store(m[u], &gm[e]); __builtin_nontemporal_store(v[u], &gv[e]); __builtin_nontemporal_store(o, &go[e]); }
                }
            }
        } else if (V == 1 || V == 3) {
            constexpr int NB = (ROW + 255) / 256;
            double m[2][4], v[2][4];
            auto load = [&](int b, int buf) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = b * 256 + 64 * u + lane;
                    if (V == 1) { m[buf][u] = e < ROW ? __builtin_nontemporal_load(&gm[e]) : 0.0; v[buf][u] = e < ROW ? __builtin_nontemporal_load(&gv[e]) : 1.0; }
                    else { m[buf][u] = e < ROW ? gm[e] : 0.0; v[buf][u] = e < ROW ? gv[e] : 1.0; }
                }
            };
            load(0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) load(b + 1, (b + 1) & 1);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int e = b * 256 + 64 * u + lane;
                    if (e < ROW) {
                        float o; one(m[b & 1][u], v[b & 1][u], xval(env, e), o, alpha, eps);
                        if (V == 1) { __builtin_nontemporal_store(m[b & 1][u], &gm[e]); __builtin_nontemporal_store(v[b & 1][u], &gv[e]); __builtin_nontemporal_store(o, &go[e]); }
                        else { gm[e] = m[b & 1][u]; gv[e] = v[b & 1][u]; go[e] = o; }
                    }
                }
            }
        } else if (V == 2) {
            // pairs: element 2p, 2p + 1 on lane p % 64 of batch p / 128 (2 pairs per lane and batch); the row's odd last element rides alone
            constexpr int NPAIR = (ROW + 1) / 2, NB = (NPAIR + 127) / 128;
            d2 m[2][2], v[2][2];
            auto load = [&](int b, int buf) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = b * 128 + 64 * u + lane;
                    d2 mm = {0.0, 0.0}, vv = {1.0, 1.0};
                    if (2 * p + 1 < ROW) { mm = __builtin_nontemporal_load((const d2 *)&gm[2 * p]); vv = __builtin_nontemporal_load((const d2 *)&gv[2 * p]); }
                    else if (2 * p < ROW) { mm.x = __builtin_nontemporal_load(&gm[2 * p]); vv.x = __builtin_nontemporal_load(&gv[2 * p]); }
                    m[buf][u] = mm; v[buf][u] = vv;
                }
            };
            load(0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (b + 1 < NB) load(b + 1, (b + 1) & 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = b * 128 + 64 * u + lane;
                    d2 mm = m[b & 1][u], vv = v[b & 1][u];
                    f2 o;
                    double a0 = mm.x, b0 = vv.x, a1 = mm.y, b1 = vv.y; float o0 = 0, o1 = 0;
                    if (2 * p < ROW) one(a0, b0, xval(env, 2 * p), o0, alpha, eps);
                    if (2 * p + 1 < ROW) one(a1, b1, xval(env, 2 * p + 1), o1, alpha, eps);
                    mm.x = a0; mm.y = a1; vv.x = b0; vv.y = b1; o.x = o0; o.y = o1;
                    if (2 * p + 1 < ROW) {
                        // (rows start at odd element offsets for odd env: 16-byte alignment holds for the float64 arrays only when base is even)
                        __builtin_nontemporal_store(mm, (d2 *)&gm[2 * p]); __builtin_nontemporal_store(vv, (d2 *)&gv[2 * p]);
                        __builtin_nontemporal_store(o0, &go[2 * p]); __builtin_nontemporal_store(o1, &go[2 * p + 1]);
                    } else if (2 * p < ROW) { __builtin_nontemporal_store(a0, &gm[2 * p]); __builtin_nontemporal_store(b0, &gv[2 * p]); __builtin_nontemporal_store(o0, &go[2 * p]); }
                }
            }
        }
    }
}

// V 4 of the row kernel: batches of 4 as in V 1, but every vector-memory instruction is issued by inline asm (the compiler then inserts no
// waits of its own) and the waits are EXACT: vmcnt is one in-order counter for loads and stores, so "batch b's statistics have
// arrived" = at most (the stores of batch b - 1) + (the loads of batch b + 1) instructions still outstanding.  Loads are never
// predicated (offsets clamped to the row), stores of a partly valid group run under an exec mask, groups past the row do not exist.
__device__ __forceinline__ void ld2_nt(double &r, const void *sbase, uint32_t voff) { asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=&v"(r) : "v"(voff), "s"(sbase)); }
__device__ __forceinline__ void st2_nt(const void *sbase, uint32_t voff, double v) { asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(voff), "v"(v), "s"(sbase) : "memory"); }
__device__ __forceinline__ void st1_nt(const void *sbase, uint32_t voff, float v) { asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(voff), "v"(v), "s"(sbase) : "memory"); }
__device__ __forceinline__ void st_masked(const void *bm, const void *bv, const void *bo, uint32_t voff8, uint32_t voff4, double m, double v, float o, uint64_t mask) {
    uint64_t sv;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1\n\tglobal_store_dwordx2 %2, %4, %7 nt\n\tglobal_store_dwordx2 %2, %5, %8 nt\n\tglobal_store_dword %3, %6, %9 nt\n\ts_mov_b64 exec, %0"
                 : "=&s"(sv) : "s"(mask), "v"(voff8), "v"(voff4), "v"(m), "v"(v), "v"(o), "s"(bm), "s"(bv), "s"(bo) : "scc", "memory");
}
template <int B, int G> struct Batch {
    static constexpr int left = ROW - B * 64 * G;
    static constexpr int groups = left <= 0 ? 0 : ((left + 63) / 64 > G ? G : (left + 63) / 64);   // groups of 64 elements with a valid lane
    static constexpr int loads = 2 * groups, stores = 3 * groups;
};
template <int N, int G> __device__ __forceinline__ void wait_vm(double (&m)[G], double (&v)[G]) {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N));
#pragma unroll
    for (int u = 0; u < G; ++u) asm volatile("" : "+v"(m[u]), "+v"(v[u]));
}
template <int B, int G> __device__ __forceinline__ void pipe_load(double (&m)[G], double (&v)[G], const void *gm, const void *gv, int lane) {
#pragma unroll
    for (int u = 0; u < G; ++u) if (u < Batch<B, G>::groups) {
        int e = B * 64 * G + 64 * u + lane; e = e < ROW ? e : ROW - 1;
        ld2_nt(m[u], gm, (uint32_t)e * 8u); ld2_nt(v[u], gv, (uint32_t)e * 8u);
    }
}
template <int B, int NB, int G> __device__ __forceinline__ void pipe_batches(double (&m0)[G], double (&v0)[G], double (&m1)[G], double (&v1)[G], const void *gm, const void *gv, const void *go,
                                                                      int env, int lane, double alpha, double eps) {
    if constexpr (B < NB) {
        if constexpr (B + 1 < NB) pipe_load<B + 1, G>(m1, v1, gm, gv, lane);
        wait_vm<(B > 0 ? Batch<B - 1, G>::stores : 0) + (B + 1 < NB ? Batch<B + 1, G>::loads : 0), G>(m0, v0);
#pragma unroll
        for (int u = 0; u < G; ++u) if (u < Batch<B, G>::groups) {
            const int e = B * 64 * G + 64 * u + lane;
            float o; one(m0[u], v0[u], xval(env, e < ROW ? e : ROW - 1), o, alpha, eps);
            if (B * 64 * G + 64 * u + 63 < ROW) { st2_nt(gm, (uint32_t)e * 8u, m0[u]); st2_nt(gv, (uint32_t)e * 8u, v0[u]); st1_nt(go, (uint32_t)e * 4u, o); }
            else st_masked(gm, gv, go, (uint32_t)(e < ROW ? e : ROW - 1) * 8u, (uint32_t)(e < ROW ? e : ROW - 1) * 4u, m0[u], v0[u], o, __ballot(e < ROW));
        }
        pipe_batches<B + 1, NB, G>(m1, v1, m0, v0, gm, gv, go, env, lane, alpha, eps);
    }
}
template <int G, int OCC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void row_kernel_asm(double *mean, double *var, float *out, int n_envs, double alpha, double eps) {
    const int lane = threadIdx.x;
    constexpr int NB = (ROW + 64 * G - 1) / (64 * G);
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        const size_t base = (size_t)__builtin_amdgcn_readfirstlane(env) * ROW;
        const void *gm = mean + base, *gv = var + base, *go = out + base;
        asm volatile("s_nop 4" : "+s"(gm), "+s"(gv), "+s"(go));
        double m0[G], v0[G], m1[G], v1[G];
        pipe_load<0, G>(m0, v0, gm, gv, lane);
        pipe_batches<0, NB, G>(m0, v0, m1, v1, gm, gv, go, env, lane, alpha, eps);
    }
    asm volatile("s_waitcnt vmcnt(0)");
}

// plain streaming kernels over the flat arrays (the stand-alone epilogue's shape)
// V 0: grid-stride, one element per thread and iteration (wrappers.hip, round 3)   V 1: two elements per thread as 16-byte words, the next
// iteration's loads issued before this iteration's stores, non-temporal
template <int V>
__global__ __launch_bounds__(256) void flat_kernel(const float *x, double *mean, double *var, float *out, size_t n, double alpha, double eps) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    if (V == 0) {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
            double m = mean[i], v = var[i]; float o;
            one(m, v, x[i], o, alpha, eps);
            mean[i] = m; var[i] = v; out[i] = o;
        }
    } else {
        const size_t np = n / 2;   // (n even in this benchmark)
        size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        d2 m = {0, 0}, v = {1, 1}; f2 xx = {0, 0};
        if (p < np) { m = __builtin_nontemporal_load((const d2 *)mean + p); v = __builtin_nontemporal_load((const d2 *)var + p); xx = __builtin_nontemporal_load((const f2 *)x + p); }
        while (p < np) {
            const size_t q = p + stride;
            d2 m2 = {0, 0}, v2 = {1, 1}; f2 x2 = {0, 0};
            if (q < np) { m2 = __builtin_nontemporal_load((const d2 *)mean + q); v2 = __builtin_nontemporal_load((const d2 *)var + q); x2 = __builtin_nontemporal_load((const f2 *)x + q); }
            double a0 = m.x, b0 = v.x, a1 = m.y, b1 = v.y; f2 o;
            float o0, o1;
            one(a0, b0, xx.x, o0, alpha, eps); one(a1, b1, xx.y, o1, alpha, eps);
            m.x = a0; m.y = a1; v.x = b0; v.y = b1; o.x = o0; o.y = o1;
            __builtin_nontemporal_store(m, (d2 *)mean + p); __builtin_nontemporal_store(v, (d2 *)var + p); __builtin_nontemporal_store(o, (f2 *)out + p);
            m = m2; v = v2; xx = x2; p = q;
        }
    }
}

int main(int argc, char **argv) {
    const int n_envs = argc > 1 ? atoi(argv[1]) : 32768;
    const size_t n = (size_t)n_envs * ROW;
    double *mean, *var; float *out, *x;
    hipMalloc(&mean, n * 8); hipMalloc(&var, n * 8); hipMalloc(&out, n * 4); hipMalloc(&x, n * 4);
    hipMemset(mean, 0, n * 8); hipMemset(x, 0, n * 4);
    { double *h = (double *)malloc(n * 8); for (size_t i = 0; i < n; ++i) h[i] = 1.0; hipMemcpy(var, h, n * 8, hipMemcpyHostToDevice); free(h); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto launch, double bytes_per_el) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        float best = 1e9f, tot = 0;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; tot += ms; if (ms < best) best = ms;
        }
        printf("%-58s %7.1f us (best %7.1f)  %6.0f GB/s\n", name, tot / 5 * 1e3, best * 1e3, bytes_per_el * n / (tot / 5 * 1e-3) / 1e9);
    };
    const double A = 0.001, E = 1e-8;
    const int blocks = argc > 2 ? atoi(argv[2]) : 5120;
    timeit("row, batches of 4, loads after stores (round 3)", [&] { hipLaunchKernelGGL(row_kernel<0>, dim3(blocks), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, batches of 4, next loads before stores", [&] { hipLaunchKernelGGL(row_kernel<1>, dim3(blocks), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, 16-byte pairs, next loads before stores", [&] { hipLaunchKernelGGL(row_kernel<2>, dim3(blocks), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, batches of 4 pipelined, cached accesses", [&] { hipLaunchKernelGGL(row_kernel<3>, dim3(blocks), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, asm + exact vmcnt, batches of 4, 4 waves/SIMD", [&] { hipLaunchKernelGGL((row_kernel_asm<4, 4>), dim3(4096), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, asm + exact vmcnt, batches of 2, 5 waves/SIMD", [&] { hipLaunchKernelGGL((row_kernel_asm<2, 5>), dim3(blocks), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, asm + exact vmcnt, batches of 2, 6 waves/SIMD", [&] { hipLaunchKernelGGL((row_kernel_asm<2, 6>), dim3(6144), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("row, asm + exact vmcnt, batches of 1, 8 waves/SIMD", [&] { hipLaunchKernelGGL((row_kernel_asm<1, 8>), dim3(8192), dim3(64), 0, 0, mean, var, out, n_envs, A, E); }, 36);
    timeit("flat, one element per thread (round 3 epilogue)", [&] { hipLaunchKernelGGL(flat_kernel<0>, dim3(4096), dim3(256), 0, 0, x, mean, var, out, n, A, E); }, 44);
    timeit("flat, 16-byte words, prefetched, non-temporal", [&] { hipLaunchKernelGGL(flat_kernel<1>, dim3(4096), dim3(256), 0, 0, x, mean, var, out, n, A, E); }, 44);
    timeit("flat, 16-byte words, prefetched, 8192 blocks", [&] { hipLaunchKernelGGL(flat_kernel<1>, dim3(8192), dim3(256), 0, 0, x, mean, var, out, n, A, E); }, 44);
    return 0;
}
