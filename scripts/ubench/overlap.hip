// Micro-benchmark: can a persistent one-wave workgroup overlap a compute phase with the row stores of the previous env?
//   A  = compute only (VALU + SALU + LDS mix, ~dynamics of the Pursuit kernel)      B = stores only (4736 B per env)
//   AB = both, per env: compute then store                                          ABL = AB + a prefetch load consumed before the stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int STORE>
__global__ __launch_bounds__(64) void k(float *out, const uint32_t *in, int n_envs, int work) {
    __shared__ uint32_t L[2048];
    const int lane = threadIdx.x;
    uint32_t a = lane, b = blockIdx.x, c = 1, d = 7;
    uint32_t cur = in[blockIdx.x * 32 + (lane & 31)];
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        uint32_t nxt = 0;
        if (MODE == 3 && env + (int)gridDim.x < n_envs) nxt = in[(size_t)(env + gridDim.x) * 32 + (lane & 31)];
        a ^= cur;
        if (MODE != 1) {  // compute phase
            for (int i = 0; i < work; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    asm volatile("v_add_u32 %0, %0, %2\n s_nop 0\n v_xor_b32 %1, %1, %0\n v_mul_lo_u32 %2, %2, %3\n v_add_u32 %3, %3, %1"
                                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
                L[(a & 1023) + lane] = b;
                __builtin_amdgcn_wave_barrier();
                c += L[(b & 1023) + 63 - lane];
            }
        }
        if (MODE == 3) asm volatile("" : "+v"(nxt));
        if (MODE != 0) {  // store phase
            float *row = out + (size_t)env * 1184;
            const float v = __uint_as_float((a + c) & 0x3FFFFFFF);
            if (STORE == 0) {
#pragma unroll
                for (int s = 0; s < 5; ++s) { int q = lane + 64 * s; v4f x = {v, v, v, v}; if (q < 296) __builtin_nontemporal_store(x, &((v4f *)row)[q]); }
            } else {
#pragma unroll
                for (int t = 0; t < 19; ++t) { int e = lane + 64 * t; if (e < 1184 && ((a >> (t & 15)) & 7) != 0) row[e] = v; }
            }
        }
        cur = nxt;
    }
    if (a == 0xdeadbeef) out[0] = (float)(a + b + c + d);
}
template <int MODE, int STORE> float run(float *out, uint32_t *in, int work) {
    const int n_envs = 65536, blocks = 4096;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, STORE>), dim3(blocks), dim3(64), 0, 0, out, in, n_envs, work);
    (void)hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, STORE>), dim3(blocks), dim3(64), 0, 0, out, in, n_envs, work);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 20 * 1000.f;
}
int main() {
    float *out; uint32_t *in; (void)hipMalloc(&out, (size_t)65536 * 1184 * 4); (void)hipMalloc(&in, (size_t)65536 * 32 * 4);
    (void)hipMemset(in, 1, (size_t)65536 * 32 * 4);
    for (int work : {4, 8, 12}) {
        printf("work=%2d | x4 nt stores:   A %.1f  B %.1f  AB %.1f  AB+prefetch %.1f us", work, run<0, 0>(out, in, work), run<1, 0>(out, in, work), run<2, 0>(out, in, work), run<3, 0>(out, in, work));
        printf(" | dword masked stores:   B %.1f  AB %.1f  AB+prefetch %.1f us\n", run<1, 1>(out, in, work), run<2, 1>(out, in, work), run<3, 1>(out, in, work));
    }
    return 0;
}
