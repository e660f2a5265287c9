// Micro-benchmark: cost of leaving "stale" observation cells untouched (partial cache lines)
// vs reading the old float4 and writing full lines, with the real geometry of the Pursuit rows
// (8 pursuers x 148 floats; stale cells = out-of-map rows/columns of channels 1-2).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// element r of pursuer p in env: is it stale?  pursuer position uniform on 16x16
__device__ inline bool stale(int env, int p, int r) {
    if (r >= 147 || r < 49) return false;
    unsigned h = hash(env * 8 + p);
    int x = h & 15, y = (h >> 4) & 15;
    int ij = r % 49, i = ij / 7, j = ij % 7;
    int gx = x - 3 + i, gy = y - 3 + j;
    return gx < 0 || gx > 15 || gy < 0 || gy > 15;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void k(float *out, int n_envs, float val) {
    const int lane = threadIdx.x;
    for (int env = blockIdx.x; env < n_envs; env += gridDim.x) {
        float *row = out + (size_t)env * 1184;
        v4f *r4 = (v4f *)row;
        if (VARIANT == 0) {  // write everything (no stale semantics)
#pragma unroll
            for (int s = 0; s < 5; ++s) { int q = lane + 64 * s; v4f v = {val, val, val, val}; if (q < 296) __builtin_nontemporal_store(v, &r4[q]); }
        } else if (VARIANT == 1) {  // current kernel: nt x4 when clean, plain dwords when partially stale
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int q = lane + 64 * s;
                if (q < 296) {
                    int p = q / 37, f = q % 37;
                    bool s0 = stale(env, p, 4 * f), s1 = stale(env, p, 4 * f + 1), s2 = stale(env, p, 4 * f + 2), s3 = stale(env, p, 4 * f + 3);
                    v4f v = {val, val, val, val};
                    if (!(s0 | s1 | s2 | s3)) __builtin_nontemporal_store(v, &r4[q]);
                    else { float *o = (float *)&r4[q]; if (!s0) o[0] = val; if (!s1) o[1] = val; if (!s2) o[2] = val; if (!s3) o[3] = val; }
                }
            }
        } else if (VARIANT == 2) {  // explicit read-merge-write: every store is a full float4
            v4f old[5]; bool need[5];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int q = lane + 64 * s; need[s] = false;
                if (q < 296) { int p = q / 37, f = q % 37; need[s] = stale(env, p, 4 * f) | stale(env, p, 4 * f + 1) | stale(env, p, 4 * f + 2) | stale(env, p, 4 * f + 3); }
                if (need[s]) old[s] = r4[q];
            }
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                int q = lane + 64 * s;
                if (q < 296) {
                    int p = q / 37, f = q % 37;
                    v4f v = {val, val, val, val};
                    if (need[s]) { if (stale(env, p, 4 * f)) v.x = old[s].x; if (stale(env, p, 4 * f + 1)) v.y = old[s].y; if (stale(env, p, 4 * f + 2)) v.z = old[s].z; if (stale(env, p, 4 * f + 3)) v.w = old[s].w; }
                    __builtin_nontemporal_store(v, &r4[q]);
                }
            }
        } else if (VARIANT == 3) {  // lane-contiguous dwords, stale lanes skipped
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int t = 0; t < 3; ++t) { int r = lane + 64 * t; if (r < 148 && !stale(env, p, r)) row[p * 148 + r] = val; }
        }
    }
}
template <int V> float run(float *buf, int n_envs, int blocks, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, buf, n_envs, 1.0f);
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, buf, n_envs, 1.0f + i);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / iters * 1000.f;
}
int main() {
    const int n_envs = 65536; float *buf; (void)hipMalloc(&buf, (size_t)n_envs * 1184 * 4); (void)hipMemset(buf, 0, (size_t)n_envs * 1184 * 4);
    for (int blocks : {4096, 8192, 16384})
        printf("blocks=%5d  all-cells x4nt %.1f us | x4nt+partial dwords %.1f | explicit read-merge-write %.1f | contiguous dwords, stale skipped %.1f\n", blocks,
               run<0>(buf, n_envs, blocks, 30), run<1>(buf, n_envs, blocks, 30), run<2>(buf, n_envs, blocks, 30), run<3>(buf, n_envs, blocks, 30));
    return 0;
}
