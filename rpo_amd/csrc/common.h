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

// Kernel-variant switches (include/rpo_hip.h: rpo_tuning, RPO_TUNE_*): one process-wide table, defined in train_ops.hip,
// read by the launch code on every call (no getenv in the library).
extern int g_rpo_tune[RPO_TUNE_COUNT];
static inline int rpo_tune(int key) { return g_rpo_tune[key]; }

// relu as ONE instruction: the integer maximum of the bit pattern with 0 (positive floats order like their bits, negative
// floats and -0.0 are negative integers).  fmaxf(x, 0.0f) compiles to two v_max_f32 -- the first quiets a signalling NaN the
// value could be -- which matters next to f32 MFMAs: vector instructions share their lanes.  Same bits for every non-NaN x.
__device__ __forceinline__ float rpo_relu_bits(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// Compute units of the CURRENT device, cached per device id (the persistent kernels launch one workgroup per CU; ADVICE r05:
// one cache for the streaming forward, backward and rollout instead of a query per launch / a first-device-only static).
static inline int rpo_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int v = 0;
        cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cached[dev];
}

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

__host__ __device__ __forceinline__ rpo_u4 rpo_philox(uint64_t seed, uint32_t id, uint32_t index, uint32_t stream,
                                                       uint32_t sub = 0u) {
    uint32_t c0 = id, c1 = index, c2 = stream, c3 = sub;
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
// The env / head row functions live in headers and are inlined into several kernels (single-stage launches and the
// fused pipelines).  With the default -ffp-contract=fast the compiler picks mul+add fusions per inlining context, so
// the same source could round differently in two kernels; RPO_FP_STRICT at the top of a function body switches
// contraction off for it, which makes "the pipeline equals the launches it replaces" hold bit for bit by construction
// (and matches the unfused numpy / torch-CPU arithmetic of the reference).
#define RPO_FP_STRICT _Pragma("clang fp contract(off)")

// Sum over the 16 lanes of a DPP row, valid in the row's lane 0 (lane & 15 == 0), with the association of the xor
// butterfly v += shfl_xor(v, 1 | 2 | 4 | 8) at that lane -- ((v0 + v1) + (v2 + v3)) per quad, then (Q0 + Q1) + (Q2 + Q3) --
// so it is a bitwise drop-in wherever only lane 0 of a row keeps the result.  __shfl_xor is a ds_bpermute + s_waitcnt per
// step (32 of them in series cost 1.7 us at the end of every MLP slab); these are four VALU instructions.
template <int CTRL>
__device__ __forceinline__ float rpo_dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rpo_quad_sum(float v) {        // every lane of the quad ends with the quad's sum
    v += rpo_dpp_mov<0xB1>(v);                                   // quad_perm:[1,0,3,2]  (== shfl_xor 1)
    v += rpo_dpp_mov<0x4E>(v);                                   // quad_perm:[2,3,0,1]  (== shfl_xor 2)
    return v;
}
__device__ __forceinline__ float rpo_row16_sum_lane0(float v) {
    v = rpo_quad_sum(v);
    v += rpo_dpp_mov<0x104>(v);                                  // row_shl:4: lane l reads lane l + 4 (0 beyond the row)
    v += rpo_dpp_mov<0x108>(v);                                  // row_shl:8
    return v;
}

// The same with the association of rpo_wave_sum (offsets 8, 4, 2, 1 in that order; its steps 32 and 16 only add the other
// rows): a bitwise drop-in for rpo_wave_sum(v) where only lanes 0..15 of the wave hold non-zero values and lane 0 keeps
// the result -- the 16 per-row terms of a tile (losses, Lagrangian partials).
__device__ __forceinline__ float rpo_row16_sum_desc_lane0(float v) {
    v += rpo_dpp_mov<0x108>(v);                                  // row_shl:8
    v += rpo_dpp_mov<0x104>(v);                                  // row_shl:4
    v += rpo_dpp_mov<0x4E>(v);                                   // quad_perm:[2,3,0,1]
    v += rpo_dpp_mov<0xB1>(v);                                   // quad_perm:[1,0,3,2]
    return v;
}

// K statistics of the 16 lanes of row 0 (lanes 0..15; the rest of the wave holds zeros), the results in EVERY lane: the
// DPP form of rpo_wave_reduce_many<K, 8> (same offsets in the same order -> same bits), without its LDS round trips.
template <int K>
__device__ __forceinline__ void rpo_row16_reduce_many(float (&v)[K], unsigned max_mask) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float x = v[k];
        if ((max_mask >> k) & 1u) {
            x = fmaxf(x, rpo_dpp_mov<0x108>(x)); x = fmaxf(x, rpo_dpp_mov<0x104>(x));
            x = fmaxf(x, rpo_dpp_mov<0x4E>(x)); x = fmaxf(x, rpo_dpp_mov<0xB1>(x));
        } else {
            x = rpo_row16_sum_desc_lane0(x);
        }
        v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 0));
    }
}

// Maximum of non-negative values over the wave, the same in every lane (order-independent, exact): DPP inside the rows,
// v_readlane across them -- no LDS round trips (rpo_wave_max: six ds_bpermute + wait in series).
__device__ __forceinline__ float rpo_wave_max_nonneg(float v) {
    v = fmaxf(v, rpo_dpp_mov<0x108>(v));
    v = fmaxf(v, rpo_dpp_mov<0x104>(v));
    v = fmaxf(v, rpo_dpp_mov<0x4E>(v));
    v = fmaxf(v, rpo_dpp_mov<0xB1>(v));                          // lane 0 of each row: the row's maximum (0 beyond the row)
    const int b = __float_as_int(v);
    const float m0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), m1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float m2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), m3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

// Sum over the wave, the same in every lane, without LDS round trips (rpo_wave_sum: six dependent ds_bpermute): DPP inside the
// 16-lane rows (row_shl fills with zeros), v_readlane across them.  Association: per row ((l + l+8) + (l+4 + l+12)) ... then
// (r0 + r1) + (r2 + r3) -- NOT rpo_wave_sum's bits.
__device__ __forceinline__ float rpo_wave_sum_rows(float v) {
    v = rpo_row16_sum_desc_lane0(v);
    const int b = __float_as_int(v);
    const float s0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), s1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float s2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), s3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (s0 + s1) + (s2 + s3);
}

// torch.clamp / np.clip propagate a NaN; fminf(fmaxf(x, lo), hi) would return `lo` for it -- and a diverged actor's NaN
// output would become a legal action at the edge of its box instead of reaching the step kernels' failure word
// (rpo_flag_nonfinite below; the reference's assert, cartpole.py:170-174).  Same bits for every non-NaN x.
__device__ __forceinline__ float rpo_clamp(float x, float lo, float hi) {
    const float c = fminf(fmaxf(x, lo), hi);
    return x != x ? x : c;
}

// take_action's exploration (agent/ddpg_pa.py:108-110): clip(ap + eps_t * noise, lo, hi), unfused like the
// RPO_NOISE_PHILOX / RPO_NOISE_EXPLICIT branches of the *_explore_project functions
__device__ __forceinline__ float rpo_explore_clip(float ap, float eps_t, float noise, float lo, float hi) {
    RPO_FP_STRICT
    return rpo_clamp(ap + eps_t * noise, lo, hi);
}

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

// K wave reductions at once, step by step: the shuffles of one butterfly step are issued together and waited for once
// (K back-to-back rpo_wave_sum calls wait for every ds_bpermute separately: 11 statistics cost 2.9 us that way).  Per value
// the same xor butterfly, so the results are bitwise those of rpo_wave_sum / rpo_wave_max.  max_mask bit k: maximum.
// FIRST < 32: only lanes [0, 2 * FIRST) hold data (the others hold zeros, which the skipped steps would only add).
template <int K, int FIRST = 32>
__device__ __forceinline__ void rpo_wave_reduce_many(float (&v)[K], unsigned max_mask) {
#pragma unroll
    for (int off = FIRST; off > 0; off >>= 1) {
        float o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) o[k] = __shfl_xor(v[k], off, RPO_WAVE);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = ((max_mask >> k) & 1u) ? fmaxf(v[k], o[k]) : v[k] + o[k];
    }
}

// non-negative float max through the integer ordering of IEEE bit patterns
__device__ __forceinline__ void rpo_atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// Episode bookkeeping of a lane (gym TimeLimit counter, running return, episode index of the reset stream): loaded as
// early as the kernel allows -- the step itself is a dependent chain, and these three loads would otherwise sit at its end.
struct RpoEpisode { int len; float ret; unsigned count; };
__device__ __forceinline__ RpoEpisode rpo_load_episode(const int* ep_len, const float* ep_ret, const unsigned* ep_count, int i) {
    return RpoEpisode{ep_len[i], ep_ret[i], ep_count[i]};
}

// Failure detection (include/rpo_hip.h: RPO_CTRL_NONFINITE).  The reference stops on a NaN action (`assert
// self.action_space.contains(action_fixed)`: cartpole.py:170-174, pendulum.py:85-89; an infinite action is clipped and passes)
// and everything after it is garbage; here a diverged actor would fill the ring with NaN transitions silently.  `bad` is the
// lane's own test; the word keeps the FIRST offending vector step + 1.  Clean path: one compare + a not-taken branch.
__device__ __forceinline__ void rpo_flag_nonfinite(long long* ctrl, bool bad) {
    if (bad && ctrl != nullptr)
        atomicCAS(reinterpret_cast<unsigned long long*>(ctrl + RPO_CTRL_NONFINITE), 0ull, (unsigned long long)(ctrl[RPO_CTRL_T] + 1));
}

// Statistics rows are split into RPO_STATS_SUB sub-rows (one cache line each); a workgroup adds into sub-row
// blockIdx % RPO_STATS_SUB.  Up to 16 workgroups therefore never share an address (bitwise reproducible sums, summed by
// the host in a fixed order); large grids spread their same-address atomics (~8 ns each, serialised) 16 ways.
__device__ __forceinline__ float* rpo_stats_row_at(float* stats, int stats_cap, long long t, unsigned blk) {
    return stats + ((t % stats_cap) * RPO_STATS_SUB + (blk % RPO_STATS_SUB)) * RPO_STATS_LEN;
}
__device__ __forceinline__ float* rpo_stats_row(float* stats, int stats_cap, long long t) {
    return rpo_stats_row_at(stats, stats_cap, t, blockIdx.x);
}

// The calling wave holds K statistics (the same values in every lane): lane k adds / maxes value k into row[slot[k]] -- ONE
// or two atomic instructions per workgroup instead of K.  Atomics of many workgroups on one 64-byte sub-row are served one
// request after the other by its L2 channel (16 workgroups x 11 single-lane requests per sub-row held the rollout kernel's
// last wave for 2.9 us); a request carrying 11 lanes of the same line counts once.
template <int K>
__device__ __forceinline__ void rpo_stats_commit(const float (&r)[K], unsigned max_mask, const int (&slot)[K], float* row) {
    const int lane = threadIdx.x & (RPO_WAVE - 1);
    float mine = 0.0f;
    int s = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane == k) { mine = r[k]; s = slot[k]; }
    const bool mx = lane < K && ((max_mask >> lane) & 1u);
    if (lane < K && !mx && mine != 0.0f) atomicAdd(row + s, mine);
    if (mx && mine > 0.0f) rpo_atomic_max_nonneg(row + s, mine);
}

// Flush kStats per-thread partials (sums for k < n_sum, maxima after) of a 256-thread workgroup into `row` slots.
template <int K>
__device__ __forceinline__ void rpo_stats_flush(const float (&v)[K], int n_sum, const int (&slot)[K], float* row,
                                                float* lds /* [4 * K] */) {
    const int lane = threadIdx.x & (RPO_WAVE - 1), wave = threadIdx.x / RPO_WAVE;
    float r[K];
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = v[k];
    rpo_wave_reduce_many<K>(r, ~0u << n_sum);
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (lane == 0) lds[wave * K + k] = r[k];
    __syncthreads();
    if (threadIdx.x < K) {
        const int k = threadIdx.x;
        float r = lds[k];
        for (int w = 1; w < RPO_BLOCK / RPO_WAVE; ++w) r = (k < n_sum) ? r + lds[w * K + k] : fmaxf(r, lds[w * K + k]);
        if (k < n_sum) {
            if (r != 0.0f) atomicAdd(row + slot[k], r);
        } else if (r > 0.0f) {
            rpo_atomic_max_nonneg(row + slot[k], r);
        }
    }
}

// The *_step kernels own ctrl[RPO_CTRL_T]: the last workgroup to finish advances it and clears the statistics rows of
// the next vector step.  Every workgroup read t at its start, before any workgroup can have arrived last.  Arrival is
// hierarchical -- 16 sub-counters on separate cache lines, then one top counter -- because 2048 returning atomics on
// ONE address cost ~16 us (measured) while 128 per address cost ~1 us.
// (`blk` of `nblk`: the stepping workgroups may be a subset of a larger launch -- the riding stages of nsplit.hip)
__device__ __forceinline__ void rpo_step_epilogue_at(long long* ctrl, long long t, float* stats, int stats_cap, unsigned blk,
                                                     unsigned nblk) {
    __syncthreads();
    if (threadIdx.x == 0 && ctrl != nullptr) {
        const unsigned sub = blk % RPO_STATS_SUB;
        const unsigned subs = nblk < RPO_STATS_SUB ? nblk : RPO_STATS_SUB;
        const unsigned long long in_sub = (nblk - sub + RPO_STATS_SUB - 1) / RPO_STATS_SUB;
        unsigned long long* sub_ctr = reinterpret_cast<unsigned long long*>(ctrl + RPO_CTRL_SUB0 + RPO_CTRL_SUB_STRIDE * sub);
        // (up to 16 workgroups: every sub-counter would see one arrival -- they go to the top counter directly)
        if (nblk <= RPO_STATS_SUB || atomicAdd(sub_ctr, 1ull) == in_sub - 1ull) {
            if (nblk > RPO_STATS_SUB) *sub_ctr = 0;
            if (atomicAdd(reinterpret_cast<unsigned long long*>(ctrl + RPO_CTRL_ARRIVE), 1ull) == subs - 1ull) {
                ctrl[RPO_CTRL_ARRIVE] = 0;
                ctrl[RPO_CTRL_T] = t + 1;
                if (stats != nullptr && stats_cap > 1) {
                    float* nxt = stats + ((t + 1) % stats_cap) * RPO_STATS_SUB * RPO_STATS_LEN;
                    for (int k = 0; k < RPO_STATS_SUB * RPO_STATS_LEN; ++k) nxt[k] = 0.0f;
                }
            }
        }
    }
}
__device__ __forceinline__ void rpo_step_epilogue(long long* ctrl, long long t, float* stats, int stats_cap) {
    rpo_step_epilogue_at(ctrl, t, stats, stats_cap, blockIdx.x, gridDim.x);
}
