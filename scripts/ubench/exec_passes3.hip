// Follow-up 2: does the 4x of narrow exec masks (<= 8 lanes) need the mask to STAY narrow?  One wavefront per SIMD alternates blocks of B dependent
// VALU instructions under the full mask with blocks of B under a 4-lane mask.  If every narrow instruction cost 4x whatever came before, the
// time per pair of blocks would be B x (1 + 4) units for every B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int B>
__global__ __launch_bounds__(64) void alt(float *out, int iters, int narrow_lanes) {
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f; const float b = 0.999f;
    const bool narrow = lane < narrow_lanes;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < B; ++k) { a = a * b; a = a + 1e-7f; }          // full mask
        if (narrow) {
#pragma unroll
            for (int k = 0; k < B; ++k) { a = a * b; a = a + 1e-7f; }      // narrow mask
        }
        asm volatile("" : "+v"(a));
    }
    out[blockIdx.x * 64 + lane] = a;
}
// long blocks: rolled loops of `reps` x 64 instructions each
__global__ __launch_bounds__(64) void alt_long(float *out, int iters, int reps, int narrow_lanes) {
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-3f; const float b = 0.999f;
    const bool narrow = lane < narrow_lanes;
    for (int i = 0; i < iters; ++i) {
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { a = a * b; a = a + 1e-7f; }
        }
        if (narrow) {
            for (int r = 0; r < reps; ++r) {
#pragma unroll
                for (int k = 0; k < 32; ++k) { a = a * b; a = a + 1e-7f; }
            }
        }
        asm volatile("" : "+v"(a));
    }
    out[blockIdx.x * 64 + lane] = a;
}
void run_long(float *d, int reps, int narrow_lanes) {
    const int total = 1 << 20;
    const int iters = total / (64 * reps);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(alt_long, dim3(1024), dim3(64), 0, 0, d, iters, reps, narrow_lanes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("blocks of %7d instructions, narrow part on %2d lanes: %7.3f ms for 2 x %d instructions = %5.2f ns per instruction on average\n", 64 * reps, narrow_lanes, ms, total, ms * 1e6 / (2.0 * total));
}
template <int B> void run(float *d, int narrow_lanes) {
    const int total = 1 << 17;   // instructions of each kind per wavefront
    const int iters = total / (2 * B);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(alt<B>, dim3(1024), dim3(64), 0, 0, d, iters, narrow_lanes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("blocks of %4d instructions, narrow part on %2d lanes: %7.3f ms for 2 x %d instructions = %5.2f ns per instruction on average\n", 2 * B, narrow_lanes, ms, total, ms * 1e6 / (2.0 * total));
}
int main() {
    float *d; hipMalloc(&d, 1 << 20);
    for (int nl : {64, 4}) { run<4>(d, nl); run<16>(d, nl); run<64>(d, nl); run<256>(d, nl); run<1024>(d, nl); }
    for (int nl : {64, 4}) for (int reps : {16, 64, 256, 1024, 4096, 16384}) run_long(d, reps, nl);
    return 0;
}
