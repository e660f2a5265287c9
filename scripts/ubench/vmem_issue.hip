// Micro-benchmark: cost of one vector-memory store instruction through a CU's address pipeline (TA) as a function of
// width, active-lane count and address pattern; the target lines stay L2-resident (each wave rewrites its own 4 KB).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(64) void k(float *buf, int iters) {
    const int lane = threadIdx.x;
    float *mine = buf + (size_t)blockIdx.x * 1024;  // 4 KB per wave
    const float v = (float)iters;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) mine[lane + 64 * (u & 3)] = v;                                       // dword, 64 lanes, contiguous
            else if (KIND == 1) { if ((lane & 3) == 0) mine[lane * 4 + (u & 3)] = v; }           // dword, 16 lanes, 64-B stride
            else if (KIND == 2) { if (lane == 5) mine[lane * 4 + (u & 3)] = v; }                 // dword, 1 lane
            else if (KIND == 3) { v4f x = {v, v, v, v}; ((v4f *)mine)[lane + 64 * (u & 3)] = x; } // dwordx4, 64 lanes
            else if (KIND == 4) mine[lane * 4 + (u & 3)] = v;                                    // dword, 64 lanes, 16-B stride
            else if (KIND == 5) { if (lane < 16) mine[lane * 4 + (u & 3)] = v; }                 // dword, 16 lanes (one quarter-wave), 16-B stride
            else if (KIND == 6) { v4f x = {v, v, v, v}; __builtin_nontemporal_store(x, &((v4f *)mine)[lane + 64 * (u & 3)]); }
            asm volatile("" ::: "memory");
        }
    }
}
template <int KIND> void run(const char *name, float *buf) {
    const int iters = 2000;
    for (int wps : {1, 4}) {
        int blocks = 256 * 4 * wps;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, buf, 10);
        (void)hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, buf, iters);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        double per_cu = (double)wps * 4 * iters * 16;
        printf("%-44s waves/SIMD=%d  %.3f ms -> %.1f ns (%.0f cycles @2.4GHz) per store instr per CU\n", name, wps, ms, ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    }
}
int main() {
    float *buf; (void)hipMalloc(&buf, (size_t)256 * 4 * 4 * 4096);
    run<0>("dword  64 lanes contiguous", buf); run<4>("dword  64 lanes 16-B stride", buf); run<1>("dword  16 lanes (every 4th) 64-B stride", buf);
    run<5>("dword  16 lanes (first quarter) 16-B stride", buf); run<2>("dword  1 lane", buf); run<3>("dwordx4 64 lanes", buf); run<6>("dwordx4 64 lanes nt", buf);
    return 0;
}
