// (Read exec_passes3.hip with this: the 4x found here for masks of <= 8 lanes sets in only after ~0.3 ms of NOTHING BUT narrow masks on a SIMD.)
// Does a wave64 VALU instruction cost fewer cycles when whole 16-lane quarters of the exec mask are off?  One wavefront per SIMD runs a chain of
// dependent v_mul_f32 / v_add_f32 under exec masks with 1, 2, 3, 4 non-empty quarters (and with the active lanes SPREAD over all quarters).
//   hipcc --offload-arch=gfx950 -O3 -o exec_passes exec_passes.hip && ./exec_passes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void chain(float *out, uint64_t mask, int iters) {
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f, b = 0.999f;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) { a = a * b; a = a + 1e-7f; }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + lane] = a;
    if (lane == 0 && blockIdx.x == 0) reinterpret_cast<uint64_t *>(out)[4096] = t1 - t0;
}
int main() {
    float *d; hipMalloc(&d, 1 << 20);
    const struct { const char *name; uint64_t m; } cases[] = {
        {"lanes 0-15  (1 quarter)", 0xFFFFull}, {"lanes 0-31  (2 quarters)", 0xFFFFFFFFull}, {"lanes 0-47  (3 quarters)", 0xFFFFFFFFFFFFull},
        {"all 64 lanes (4 quarters)", ~0ull}, {"16 lanes spread: every 4th lane (4 quarters)", 0x1111111111111111ull},
        {"4 lanes: one per quarter", 0x0001000100010001ull}, {"lanes 0-3 only", 0xFull}, {"lanes 16-19 only", 0xF0000ull}, {"lanes 0-3 and 32-35", 0xF0000000Full},
        {"lane 0 only", 1ull}, {"lanes 0-1", 3ull}, {"lanes 0-7", 0xFFull}, {"lanes 0-8", 0x1FFull}, {"lanes 0-11", 0xFFFull}, {"lanes 0-14", 0x7FFFull},
        {"8 lanes spread: every 8th", 0x0101010101010101ull}, {"12 lanes: 3 of every quad in quarter 0 + 4", 0x7777ull},
        {"all 64 lanes again", ~0ull}, {"lanes 0-3 only again", 0xFull}, {"48 lanes: 3 of every quad", 0x7777777777777777ull}, {"32 lanes: 2 of every quad", 0x3333333333333333ull}};
    for (auto &c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(1024), dim3(64), 0, 0, d, c.m, 2000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint64_t clk; hipMemcpy(&clk, reinterpret_cast<uint64_t *>(d) + 4096, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-48s %8.3f ms  = %5.2f ns per dependent VALU instruction\n", c.name, ms, ms * 1e6 / (2000.0 * 128));
        }
    }
    return 0;
}
