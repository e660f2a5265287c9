// Micro-benchmark: how fast can MI355X absorb the observation-store patterns?
// Build: hipcc --offload-arch=gfx950 -O3 store_patterns.hip -o store_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));

// each wave writes ROWS rows of 1184 floats (one env's observation row), persistent over envs
template <int VARIANT>
__global__ __launch_bounds__(64) void k(float *out, int n_envs, float val) {
    const int lane = threadIdx.x;
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        float *row = out + (size_t)env * 1184;
        if (VARIANT == 0 || VARIANT == 1) {  // float4, lane-contiguous
            v4f *r4 = (v4f *)row;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int q = lane + 64 * s;
                v4f v = {val, val + s, val, val};
                if (q < 296) { if (VARIANT == 0) __builtin_nontemporal_store(v, &r4[q]); else r4[q] = v; }
            }
        } else if (VARIANT == 2 || VARIANT == 3) {  // dword, lane-contiguous
#pragma unroll
            for (int s = 0; s < 19; ++s) {
                int q = lane + 64 * s;
                if (q < 1184) { if (VARIANT == 3) __builtin_nontemporal_store(val + s, &row[q]); else row[q] = val + s; }
            }
        } else if (VARIANT == 4) {  // dword, lane-contiguous, ~20% of lanes skipped (pseudo-random)
#pragma unroll
            for (int s = 0; s < 19; ++s) {
                int q = lane + 64 * s;
                unsigned h = (unsigned)(q * 2654435761u + env * 40503u) >> 24;
                if (q < 1184 && h > 51) row[q] = val + s;
            }
        } else if (VARIANT == 5) {  // float4 fast path + 4 strided dword stores for 25% of lanes (current kernel)
            v4f *r4 = (v4f *)row;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int q = lane + 64 * s;
                unsigned h = (unsigned)(q * 2654435761u + env * 40503u) >> 24;
                bool slow = h < 64;
                v4f v = {val, val + s, val, val};
                if (q < 296) {
                    if (!slow) __builtin_nontemporal_store(v, &r4[q]);
                    else {
                        float *o = (float *)&r4[q];
                        if (h & 1) o[0] = val;
                        if (h & 2) o[1] = val;
                        if (h & 4) o[2] = val;
                        if (h & 8) o[3] = val;
                    }
                }
            }
        } else if (VARIANT == 6) {  // per pursuer: 3 dword stores (148 = 64+64+20), 8 pursuers
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    int r = lane + 64 * t;
                    if (r < 148) row[p * 148 + r] = val + t;
                }
        }
    }
}

template <int V> float run(float *buf, int n_envs, int blocks, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, buf, n_envs, 1.0f);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, buf, n_envs, 1.0f + i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / iters * 1000.f;
}

int main() {
    const int n_envs = 65536; float *buf; hipMalloc(&buf, (size_t)n_envs * 1184 * 4);
    hipMemset(buf, 0, (size_t)n_envs * 1184 * 4);
    const double mb = n_envs * 1184.0 * 4 / 1e6;
    for (int blocks : {2048, 4096, 8192, 16384}) {
        printf("blocks=%5d  x4nt %.1f us | x4 %.1f | dw %.1f | dw nt %.1f | dw 20%%skip %.1f | x4+partial %.1f | per-pursuer dw %.1f   (%.0f MB)\n", blocks,
               run<0>(buf, n_envs, blocks, 50), run<1>(buf, n_envs, blocks, 50), run<2>(buf, n_envs, blocks, 50), run<3>(buf, n_envs, blocks, 50),
               run<4>(buf, n_envs, blocks, 50), run<5>(buf, n_envs, blocks, 50), run<6>(buf, n_envs, blocks, 50), mb);
    }
    return 0;
}
