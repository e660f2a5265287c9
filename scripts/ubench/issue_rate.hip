// Micro-benchmark: instruction issue rates of a gfx950 CU as this kernel family sees them (one-wave workgroups).
//   VALU-only loop, SALU-only loop, mixed loop, at 1/2/4/8 waves per SIMD; also s_memtime ticks vs wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint64_t *ticks) {
    uint32_t a = threadIdx.x, b = blockIdx.x, c = 3, d = 5;
    uint32_t s0 = blockIdx.x, s1 = 7, s2 = 11, s3 = 13;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 64 independent-ish VALU ops (4 chains)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(i));
        } else if (KIND == 1) {  // 64 SALU ops
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("s_add_u32 %0, %0, %4\n s_xor_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_xor_b32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(i) : "scc");
        } else if (KIND == 2) {  // 32 VALU + 32 SALU interleaved
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 %2, %2, %5\n v_xor_b32 %1, %1, %4\n s_xor_b32 %3, %3, %5" : "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : "v"(i), "s"(i) : "scc");
        } else if (KIND == 3) {  // 64 dependent VALU ops (1 chain)
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(i));
        } else if (KIND == 4) {  // v_readlane x64
#pragma unroll
            for (int u = 0; u < 16; ++u)
                asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a));
        } else if (KIND == 5) {  // 64 f64 adds
            double x = a, y = b;
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(x), "+v"(y) : "v"((double)i));
            a += (uint32_t)x + (uint32_t)y;
        } else if (KIND == 6) {  // v_mul_lo_u32 / v_mul_hi_u32 (Philox)
#pragma unroll
            for (int u = 0; u < 32; ++u) asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_hi_u32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(i | 1));
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int KIND> void run(const char *name, uint32_t *out, uint64_t *ticks) {
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * 4 * wps;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, 100, ticks);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, ticks);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        uint64_t t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        double instr_per_simd = (double)wps * iters * 64;
        printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz); s_memtime %.1f MHz\n", name, wps, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4, (double)t / (ms * 1e3));
    }
}
int main() {
    uint32_t *out; uint64_t *ticks; (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4); (void)hipMalloc(&ticks, 8);
    run<0>("valu 4 chains", out, ticks); run<3>("valu 1 chain", out, ticks); run<1>("salu", out, ticks); run<2>("valu+salu 1:1", out, ticks);
    run<4>("v_readlane", out, ticks); run<5>("v_add_f64", out, ticks); run<6>("v_mul_lo/hi_u32", out, ticks);
    return 0;
}
