// Follow-up to exec_passes.hip: is the 4x of a VALU instruction under an exec mask of <= 8 lanes a LATENCY (dependent issue) or a THROUGHPUT effect,
// and does a second resident wavefront hide it?   ILP = independent chains interleaved in one wavefront; WPS = wavefronts per SIMD (256 CUs x 4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int ILP>
__global__ __launch_bounds__(64) void chain(float *out, uint64_t mask, int iters) {
    const int lane = threadIdx.x;
    float a[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) a[j] = 1.0f + lane * 1e-3f + j;
    const float b = 0.999f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64 / ILP; ++k) {
#pragma unroll
                for (int j = 0; j < ILP; ++j) a[j] = a[j] * b;
#pragma unroll
                for (int j = 0; j < ILP; ++j) a[j] = a[j] + 1e-7f;
            }
        }
    }
    float s = 0; 
#pragma unroll
    for (int j = 0; j < ILP; ++j) s += a[j];
    out[blockIdx.x * 64 + lane] = s;
}
template <int ILP> void run(float *d, const char *name, uint64_t m, int wps) {
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain<ILP>, dim3(1024 * wps), dim3(64), 0, 0, d, m, 2000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("ILP %d, %d wavefront(s) per SIMD, %-22s %8.3f ms = %5.2f ns per VALU instruction and wavefront\n", ILP, wps, name, ms, ms * 1e6 / (2000.0 * 128));
    }
}
int main() {
    float *d; hipMalloc(&d, 8 << 20);
    const struct { const char *name; uint64_t m; } cases[] = {{"all 64 lanes", ~0ull}, {"lanes 0-8", 0x1FFull}, {"lanes 0-7", 0xFFull}, {"lanes 0-3", 0xFull}};
    for (int wps : {1, 2, 4}) for (auto &c : cases) { run<1>(d, c.name, c.m, wps); run<2>(d, c.name, c.m, wps); run<4>(d, c.name, c.m, wps); }
    return 0;
}
