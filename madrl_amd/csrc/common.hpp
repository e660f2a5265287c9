// common.hpp -- shared host/device helpers for libmadrl_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/madrl_hip.h"

namespace madrl {

// ---------------------------------------------------------------- error reporting
char *last_error_buf();  // thread-local, defined in abi.hip

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define MADRL_HIP_TRY(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return ::madrl::fail(MADRL_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
                                 hipGetErrorString(_e), __FILE__, __LINE__);                 \
    } while (0)

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based generator (Salmon, Moraes, Dror, Shaw, SC'11).  One call = 4 x 32 random
// bits addressed by (counter, key); no state to carry, so every (env, tick, agent) draw is
// independent of launch shape and of the number of GPUs.
struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

#ifndef MADRL_PHILOX_VARIANT
#define MADRL_PHILOX_VARIANT 0
#endif
// a ^ b ^ c: one v_bitop3_b32 on gfx950 (truth table 0x96)
__host__ __device__ inline uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

// Each round is two 32 x 32 -> 64 products (one v_mad_u64_u32 each on the device, instead of a v_mul_hi_u32 / v_mul_lo_u32
// pair) and two three-way XORs: 4 VALU ops per round for per-lane counters.
__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
#if MADRL_PHILOX_VARIANT == 1   // separate high / low products
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = xor3(hi1, c1, k0); c1 = lo1; c2 = xor3(hi0, c3, k1); c3 = lo0;
#elif MADRL_PHILOX_VARIANT == 2  // and two-way XORs (round 1)
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
#else
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = xor3((uint32_t)(p1 >> 32), c1, k0);
        c1 = (uint32_t)p1;
        c2 = xor3((uint32_t)(p0 >> 32), c3, k1);
        c3 = (uint32_t)p0;
#endif
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// RNG contract tags: counter = (global env id, tick, index, tag | attempt << 8)
enum : uint32_t { TAG_EVADER_ACT = 0, TAG_RESET_POS = 1, TAG_RESET_ENV = 2, TAG_PURSUER_ACT = 3 };

// uniform double in [0,1) from 53 random bits
__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
    return (double)(((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The flag plane of a step launch (include/madrl_hip.h, madrl_pursuit_flags_offset): the done byte split into one 0 / 1 byte per
// meaning, so that the host side hands out bool views of it instead of launching kernels that mask bits.
//   byte 0 = bit 0 (episode over)   byte 1 = bit 1 (max_steps reached)   byte 2 = bit 7 (a capacity overflowed)   byte 3 = the done byte
__host__ __device__ inline uint32_t done_flag_word(uint32_t done_byte) {
    return (done_byte & 1u) | ((done_byte & 2u) << 7) | ((done_byte & 0x80u) << 9) | (done_byte << 24);
}

// XCD-aware walk of the env range by a grid of persistent one-wavefront workgroups.  Workgroups are dealt round-robin to
// the 8 XCDs (block b runs on XCD b % 8, each with its own L2), so the plain walk env = b + k * gridDim hands NEIGHBOURING
// envs to DIFFERENT L2s: every cache line shared by two envs' rows / records / reward words is then written back partially
// by two or more L2s.  Here XCD x owns the contiguous eighth [x * per, (x + 1) * per) of the envs and its workgroups stride
// through that chunk, so lines shared by neighbours merge in one L2.  (Measured on the hostage world: 4 107 -> 2 3xx bytes
// written per env-step; speed-only -- env results do not depend on the order in which they are processed.)
struct EnvWalk {
    int64_t base, first, stride, lim;  // env = base + li for li = first, first + stride, ... < lim
};
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ EnvWalk env_walk(int64_t n_envs) {
    EnvWalk w;
    const int64_t G = gridDim.x, b = blockIdx.x;
    if ((G & 7) == 0 && n_envs >= 8 * 8) {
        const int64_t per = (n_envs + 7) / 8;
        w.base = (b & 7) * per;
        w.first = b >> 3;
        w.stride = G >> 3;
        w.lim = (w.base + per <= n_envs ? per : n_envs - w.base);
        if (w.lim < 0) w.lim = 0;
    } else {
        w.base = 0; w.first = b; w.stride = G; w.lim = n_envs;
    }
    return w;
}
#endif

}  // namespace madrl
