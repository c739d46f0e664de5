// Shared device helpers for librpo_hip.so (gfx950 / CDNA4 only: wave64, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpo_hip.h"

#define RPO_BLOCK 256
#define RPO_WAVE 64
// Memory-bound kernels are launched with at most this many workgroups (256 CUs x 8 resident blocks) and
// grid-stride over the rest: bounds the number of per-block atomics on the shared statistics words.
#define RPO_MAX_GRID 2048

#define RPO_LAUNCH_CHECK()                        \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

static inline int rpo_grid_for(long long n, int per_block = RPO_BLOCK) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > RPO_MAX_GRID) g = RPO_MAX_GRID;
    return (int)g;
}

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct rpo_u4 {
    uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ rpo_u4 rpo_philox(uint64_t seed, uint32_t id, uint32_t index, uint32_t stream) {
    uint32_t c0 = id, c1 = index, c2 = stream, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return rpo_u4{c0, c1, c2, c3};
}

// [0,1) with 24 random bits
__host__ __device__ __forceinline__ float rpo_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// standard normal by Box-Muller from two words; u1 in (0,1]
__device__ __forceinline__ float rpo_normal(uint32_t a, uint32_t b) {
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-08f;
    const float u2 = rpo_u01(b);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ float rpo_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, RPO_WAVE);
    return v;
}
__device__ __forceinline__ float rpo_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, RPO_WAVE));
    return v;
}

// non-negative float max through the integer ordering of IEEE bit patterns
__device__ __forceinline__ void rpo_atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// The *_step kernels own ctrl[RPO_CTRL_T]: the last workgroup to finish advances it and clears the statistics row
// of the next vector step.  Every workgroup read t at its start, before any workgroup can have arrived last.
__device__ __forceinline__ void rpo_step_epilogue(long long* ctrl, long long t, float* stats, int stats_cap) {
    __syncthreads();
    if (threadIdx.x == 0 && ctrl != nullptr) {
        const unsigned long long arrived =
            atomicAdd(reinterpret_cast<unsigned long long*>(ctrl + RPO_CTRL_ARRIVE), 1ull);
        if (arrived == (unsigned long long)gridDim.x - 1ull) {
            ctrl[RPO_CTRL_ARRIVE] = 0;
            ctrl[RPO_CTRL_T] = t + 1;
            if (stats != nullptr && stats_cap > 1) {
                float* nxt = stats + ((t + 1) % stats_cap) * RPO_STATS_LEN;
                for (int k = 0; k < RPO_STATS_LEN; ++k) nxt[k] = 0.0f;
            }
        }
    }
}
