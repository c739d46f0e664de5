// Layer-by-layer form of the MLP for the WIDE networks of EVOPF-v0 (E = 256: actor 57 -> 256 -> 256 -> 14, concatenating
// critic (57 -> 256 | 43 -> 256) -> 256 -> 1; model/policy.py:24-33,48-71, model/value.py:5-31,61-100) at update-batch sizes.
//
// The row-tile kernels (mlp.hip) give 16 rows to ONE workgroup through all three layers: at batch 256 that is 16 workgroups,
// each pulling the whole 256 / 512 KB hidden matrix through one compute unit behind a 57-input first layer on the vector
// ALU -- 25 us (actor), 36 us (Q_targ || Q), 28 us (backward rows) per launch, all on the critical path of an iteration whose
// other long pole is the power-flow projection.  Here every layer is its own small GEMM launch that uses the width of the
// chip -- 16-row x 64-column tiles, one 16 x 16 f32 MFMA tile per wave, the k loop an ordered chain (exact f32, bitwise
// reproducible) -- and the pre-activations x0 / h1 travel through memory (the backward needs them saved anyway):
//   forward   x0 = [s Ws^T + bs | a Wa^T + ba]  ->  h1 = relu(x0) W0^T + b0  ->  out = relu(h1) W1^T + b1
//   backward  dh = (dout W1) * 1[h1 > 0]  ->  dx0 = (dh W0) * 1[x0 > 0]  ->  da = dx0[:, action half] Wa
// (the parameter gradients keep mlp_bwd_weights_kernel).  A launch is ~3-5 us, three of them replace one 25-36 us launch.
#pragma once
#include "mlp_bwd.h"

namespace rpo_mlp_dev {

// C[m][n] = (sum_k opA(A[m][k]) W(k, n) [+ sum_k A2[m][k] W2(k, n)]) + bias[n], zeroed where mask[m][n] <= 0.
// W_NK: W is [N][K] (torch's [out][in]: forward layers); else [K][N] (backward: the same matrices walked the other way).
struct GemmArgs {
    const float* A; int lda; int K;
    const float* W; int ldw;
    const float* A2; int lda2; int K2;         // optional second operand pair (the "add" critic's action embedding; the
    const float* W2; int ldw2;                 //  second head of a Gaussian policy in the backward)
    const float* bias;                         // [N] or NULL
    const float* bias2;                        // added too when given ("add" critics: bs + ba)
    float* C; int ldc;
    const float* mask; int ldmask;             // NULL, or C is zeroed where mask <= 0 (the relu of the saved pre-activation)
    int M, N;
    int relu_a;                                // opA = relu (the consumer applies the producer's activation)
    TdArgs td;                                 // td.q != NULL: A (one column, K = 1) is not read but PRODUCED here -- the TD target +
                                               // Huber loss of the tile's rows (the prologue of mlp_bwd_rows_body): dLoss/dQ into the
                                               // A tile, and (column block 0) into td.dq_out with the tile's loss share
};
constexpr int kGemmProblems = 8;               // problems per launch (blockIdx.z): RPOSAC's four critics x {state, action} halves
struct GemmArgs4 { GemmArgs g[kGemmProblems]; };

constexpr int kGemmThreads = 256, kGemmCols = 64, kGemmMaxK = 512;

#ifndef RPO_GEMM_SKIP
#define RPO_GEMM_SKIP 0            // debug builds only (tools/probe/build_stream_variants.sh KIND=gemm): timing with a phase left
#endif                             // out -- 1 no A loads, 2 no weight loads, 4 no MFMA chain, 8 no stores, 16 no epilogue operands

// One operand pair of a 16 x 16 output tile.  The k range runs in chunks of CH blocks of 16: the weight operands of chunk c + 1
// are requested before the MFMAs of chunk c are issued (register ping-pong), in a ROLLED loop.  Two things this replaces, both
// measured at 24 us for the hidden layer (K = 512): loads inside the k loop (32 dependent L2 round trips), and the fully
// unrolled form (3 700 instructions executed once per workgroup: bound by instruction fetch, every line a cold miss).
// Addresses are clamped instead of guarded (no branches); the A tile in LDS is zero beyond K, so operands beyond K multiply zeros.
// VEC: W is [N][K] with 16-byte aligned rows and K % 16 == 0.
template <bool W_NK, bool VEC, int CH>
__device__ __forceinline__ void gemm_load_chunk(float (&b)[CH][4], const float* __restrict__ W, int ldw, int K, int nc, int c, int lg) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const int kb = (c * CH + u) * 16 + lg * 4;
        if (RPO_GEMM_SKIP & 2) {
            b[u][0] = b[u][1] = b[u][2] = b[u][3] = 0.001f * (float)(kb + nc);
        } else if (VEC) {
            const int kc = kb + 3 < K ? kb : 0;
            const float4 v = *reinterpret_cast<const float4*>(&W[(size_t)nc * ldw + kc]);
            b[u][0] = v.x; b[u][1] = v.y; b[u][2] = v.z; b[u][3] = v.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kb + q, kc = k < K ? k : K - 1;
                b[u][q] = W_NK ? W[(size_t)nc * ldw + kc] : W[(size_t)kc * ldw + nc];
            }
        }
    }
}

template <int CH>
__device__ __forceinline__ f32x4 gemm_mfma_chunk(const float (&b)[CH][4], const float* a_s, int ldk, int c, int li, int lg, f32x4 acc) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
        const float4 a4 = *reinterpret_cast<const float4*>(&a_s[li * ldk + (c * CH + u) * 16 + lg * 4]);
        if (RPO_GEMM_SKIP & 4) { acc[0] += a4.x * b[u][0]; acc[1] += a4.y * b[u][1]; acc[2] += a4.z * b[u][2]; acc[3] += a4.w * b[u][3]; continue; }
        acc = mfma4(a4.x, b[u][0], acc);
        acc = mfma4(a4.y, b[u][1], acc);
        acc = mfma4(a4.z, b[u][2], acc);
        acc = mfma4(a4.w, b[u][3], acc);
    }
    return acc;
}

// chunks [c0, c1) of the k range (the whole range: 0, ceil(K / (16 CH)))
template <bool W_NK, bool VEC, int CH>
__device__ __forceinline__ f32x4 gemm_chain_range(const float* a_s, int ldk, const float* __restrict__ W, int ldw, int K, int N, int n,
                                                  int li, int lg, f32x4 acc, int c0, int c1) {
    const int nc = n < N ? n : N - 1;
    float b0[CH][4], b1[CH][4];
    if (c0 < c1) gemm_load_chunk<W_NK, VEC, CH>(b0, W, ldw, K, nc, c0, lg);
    for (int c = c0; c < c1; c += 2) {                            // (columns n >= N compute garbage that is never stored)
        if (c + 1 < c1) gemm_load_chunk<W_NK, VEC, CH>(b1, W, ldw, K, nc, c + 1, lg);
        acc = gemm_mfma_chunk<CH>(b0, a_s, ldk, c, li, lg, acc);
        if (c + 1 < c1) {
            if (c + 2 < c1) gemm_load_chunk<W_NK, VEC, CH>(b0, W, ldw, K, nc, c + 2, lg);
            acc = gemm_mfma_chunk<CH>(b1, a_s, ldk, c + 1, li, lg, acc);
        }
    }
    return acc;
}

template <bool W_NK, bool VEC, int CH>
__device__ __forceinline__ f32x4 gemm_chain(const float* a_s, int ldk, const float* __restrict__ W, int ldw, int K, int N, int n, int li,
                                            int lg, f32x4 acc) {
    return gemm_chain_range<W_NK, VEC, CH>(a_s, ldk, W, ldw, K, N, n, li, lg, acc, 0, (K + 16 * CH - 1) / (16 * CH));
}

// The 16 rows of an A operand into LDS, zero-padded to `kw` columns (a multiple of 16 CH).  Eight loads per thread are in
// flight before the first LDS write; rows that allow it move as 16-byte vectors; clamped addresses, no branches around loads.
__device__ __forceinline__ void gemm_stage(float* a_s, int ldk, int kw, const float* __restrict__ A, int lda, int K, int m0, int M,
                                           bool relu, bool vec) {
    const int tid = threadIdx.x;
    if (RPO_GEMM_SKIP & 1) {
        for (int idx = tid; idx < kRows * kw; idx += kGemmThreads) a_s[(idx / kw) * ldk + idx % kw] = 0.5f;
        return;
    }
    if (vec) {                                                   // (uniform: rows are 16-byte aligned and K % 4 == 0)
        const int q = kw / 4, total = kRows * q;
        for (int base = tid; base < total; base += kGemmThreads * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * kGemmThreads, r = (idx / q) & (kRows - 1), k = (idx % q) * 4;
                const int rc = m0 + r < M ? m0 + r : M - 1, kc = k + 3 < K ? k : 0;
                v[u] = *reinterpret_cast<const float4*>(&A[(size_t)rc * lda + kc]);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * kGemmThreads, r = idx / q, k = (idx % q) * 4;
                float4 w = (m0 + r < M && k + 3 < K) ? v[u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (relu) { w.x = fmaxf(w.x, 0.0f); w.y = fmaxf(w.y, 0.0f); w.z = fmaxf(w.z, 0.0f); w.w = fmaxf(w.w, 0.0f); }
                if (idx < total) *reinterpret_cast<float4*>(&a_s[r * ldk + k]) = w;
            }
        }
        return;
    }
    const int total = kRows * kw;
    for (int base = tid; base < total; base += kGemmThreads * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kGemmThreads, r = (idx / kw) & (kRows - 1), k = idx % kw;
            const int rc = m0 + r < M ? m0 + r : M - 1, kc = k < K ? k : K - 1;
            v[u] = A[(size_t)rc * lda + kc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * kGemmThreads, r = idx / kw, k = idx % kw;
            const float w = (m0 + r < M && k < K) ? v[u] : 0.0f;
            if (idx < total) a_s[r * ldk + k] = relu ? fmaxf(w, 0.0f) : w;
        }
    }
}

// grid (ceil(N / 64), ceil(M / 16), problems); 256 threads = 4 waves = 4 output tiles of 16 x 16.  CH = k blocks of 16 per
// chunk: 1 for K <= 16 (the heads' backward), 4 for K <= 64 (first layers), 8 beyond; all problems of a launch share it.
template <bool W_NK, int CH>
__global__ __launch_bounds__(kGemmThreads) void mlp_gemm_kernel(GemmArgs4 all, int vec_a, int vec_w) {
    const GemmArgs& p = all.g[blockIdx.z];
    // These launches are the update branch of the EVOPF windows: a few hundred waves next to the 1024 resident, issue-bound waves
    // of the rollout's projection on the other branch (which has slack).  Raised priority gives them the issue slots.
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __attribute__((aligned(16))) float a_s[];   // [16][ldk] (+ [16][ldk2])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * kRows, n = (blockIdx.x * 4 + wave) * 16 + li;
    if ((int)blockIdx.x * kGemmCols >= p.N) return;
    const int kw = (p.K + 16 * CH - 1) / (16 * CH) * (16 * CH), ldk = kw + 4;
    const int kw2 = p.A2 ? (p.K2 + 16 * CH - 1) / (16 * CH) * (16 * CH) : 0, ldk2 = kw2 + 4;
    float* a2_s = a_s + kRows * ldk;
    if (p.td.q) {
        for (int idx = tid; idx < kRows * kw; idx += kGemmThreads) a_s[(idx / kw) * ldk + idx % kw] = 0.0f;
        __syncthreads();
        if (tid < 64) {
            const TdArgs& t = p.td;
            const int i = m0 + tid;
            float dq = 0.0f, hub = 0.0f;
            if (tid < kRows && i < p.M) {
                const float qn = rpo_head_dev::td_next_value(t.qn1[i], t.qn2 ? t.qn2[i] : 0.0f, t.qn2 != nullptr,
                                                             t.logp ? t.logp[i] : 0.0f, t.logp != nullptr, t.alpha);
                const float y = rpo_head_dev::td_target(t.reward[(size_t)i * t.reward_stride], t.done[(size_t)i * t.done_stride],
                                                        t.gamma, qn);
                dq = rpo_head_dev::td_huber_row(t.q[i], y, 1.0f / (float)p.M, &hub);
                a_s[tid * ldk] = dq;
                if (blockIdx.x == 0) t.dq_out[i] = dq;
            }
            const float sum = rpo_row16_sum_desc_lane0(hub);     // (the same 16-term sum as the rows kernel's prologue)
            if (tid == 0 && blockIdx.x == 0) t.loss_partial[blockIdx.y] = sum;
        }
    } else {
        gemm_stage(a_s, ldk, kw, p.A, p.lda, p.K, m0, p.M, p.relu_a != 0, vec_a != 0);
    }
    if (p.A2) gemm_stage(a2_s, ldk2, kw2, p.A2, p.lda2, p.K2, m0, p.M, false, false);
    // (the epilogue's operands too: they do not depend on the chain)
    const int nc = n < p.N ? n : p.N - 1;
    const float bv = (RPO_GEMM_SKIP & 16) ? 0.0f : (p.bias ? p.bias[nc] : 0.0f) + (p.bias2 ? p.bias2[nc] : 0.0f);
    float mk[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (p.mask && !(RPO_GEMM_SKIP & 16)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lg * 4 + i, mc = m < p.M ? m : p.M - 1;
            mk[i] = p.mask[(size_t)mc * p.ldmask + nc];
        }
    }
    __syncthreads();
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (W_NK && vec_w) acc = gemm_chain<W_NK, W_NK, CH>(a_s, ldk, p.W, p.ldw, p.K, p.N, n, li, lg, acc);
    else acc = gemm_chain<W_NK, false, CH>(a_s, ldk, p.W, p.ldw, p.K, p.N, n, li, lg, acc);      // state inputs first ...
    if (p.A2) acc = gemm_chain<W_NK, false, CH>(a2_s, ldk2, p.W2, p.ldw2, p.K2, p.N, n, li, lg, acc);   // ... then the action's
    // acc[i] = C[m0 + 4 lg + i][n]
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lg * 4 + i;
            if (m < p.M) {
                float v = acc[i] + bv;
                if (p.mask && !(mk[i] > 0.0f)) v = 0.0f;
                if ((RPO_GEMM_SKIP & 8) && v != 12345.678f) continue;
                p.C[(size_t)m * p.ldc + n] = v;
            }
        }
    }
}

// The k range split over the four waves of a workgroup (round 4).  A 16 x 16 output tile is an ORDERED chain of K / 4 MFMA steps
// on one SIMD -- 3.4 us for K = 256, 6.8 us for the concatenating critic's K = 512 -- and that chain, not a launch fee, is what
// the hidden-layer / head / dx0 launches of the EVOPF windows cost (5-9 us each, seven of them on the critical path of an
// iteration).  Here a workgroup owns ONE column tile and its waves take a quarter of the k chunks each (chunks of 64); the four
// partial tiles meet in LDS and wave 0 adds them in the fixed order ((q0 + q1) + q2) + q3 and runs the epilogue: a quarter of
// the chain, four times the workgroups (the chip is a quarter full otherwise), bitwise reproducible, OTHER summation order than
// the one-chain kernel (float32 round-off; the comparisons with torch hold at the same tolerances).  Problems without a second
// operand pair and without the TD prologue, K >= 256.
template <bool W_NK>
__global__ __launch_bounds__(kGemmThreads) void mlp_gemm_ksplit_kernel(GemmArgs4 all, int vec_a, int vec_w) {
    const GemmArgs& p = all.g[blockIdx.z];
    __builtin_amdgcn_s_setprio(2);
    extern __shared__ __attribute__((aligned(16))) float a_s[];   // [16][ldk] | partial tiles [4][64] x f32x4
    constexpr int CH = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int m0 = blockIdx.y * kRows, n = blockIdx.x * 16 + li;
    if ((int)blockIdx.x * 16 >= p.N) return;
    const int kw = (p.K + 16 * CH - 1) / (16 * CH) * (16 * CH), ldk = kw + 4;
    f32x4* red = reinterpret_cast<f32x4*>(a_s + kRows * ldk);
    gemm_stage(a_s, ldk, kw, p.A, p.lda, p.K, m0, p.M, p.relu_a != 0, vec_a != 0);
    const int nc = n < p.N ? n : p.N - 1;
    const float bv = (RPO_GEMM_SKIP & 16) ? 0.0f : (p.bias ? p.bias[nc] : 0.0f) + (p.bias2 ? p.bias2[nc] : 0.0f);
    float mk[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (p.mask && wave == 0 && !(RPO_GEMM_SKIP & 16)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lg * 4 + i, mc = m < p.M ? m : p.M - 1;
            mk[i] = p.mask[(size_t)mc * p.ldmask + nc];
        }
    }
    __syncthreads();
    const int chunks = kw / (16 * CH), per = (chunks + 3) / 4, c0 = wave * per, c1 = c0 + per < chunks ? c0 + per : chunks;
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (W_NK && vec_w) acc = gemm_chain_range<W_NK, W_NK, CH>(a_s, ldk, p.W, p.ldw, p.K, p.N, n, li, lg, acc, c0, c1);
    else acc = gemm_chain_range<W_NK, false, CH>(a_s, ldk, p.W, p.ldw, p.K, p.N, n, li, lg, acc, c0, c1);
    red[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
    if (n < p.N) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lg * 4 + i;
            if (m < p.M) {
                float v = acc[i] + bv;
                if (p.mask && !(mk[i] > 0.0f)) v = 0.0f;
                if ((RPO_GEMM_SKIP & 8) && v != 12345.678f) continue;
                p.C[(size_t)m * p.ldc + n] = v;
            }
        }
    }
}

static inline bool gemm_ksplit_enabled() {
    return rpo_tune(RPO_TUNE_GEMM_KSPLIT) != 0;
}

template <bool W_NK>
static inline int gemm_launch(const GemmArgs* g, int count, hipStream_t stream) {
    if (count < 1 || count > kGemmProblems) return RPO_ERR_ARG;
    GemmArgs4 all{};
    int maxN = 0, maxM = 0, maxK = 0;
    bool vec_a = true, vec_w = W_NK;
    for (int i = 0; i < count; ++i) {
        all.g[i] = g[i];
        if (g[i].K > kGemmMaxK || g[i].K <= 0 || (g[i].A2 && (g[i].K2 <= 0 || g[i].K2 > kGemmMaxK))) return RPO_ERR_ARG;
        maxN = g[i].N > maxN ? g[i].N : maxN;
        maxM = g[i].M > maxM ? g[i].M : maxM;
        maxK = g[i].K > maxK ? g[i].K : maxK;
        if (g[i].A2 && g[i].K2 > maxK) maxK = g[i].K2;
        vec_a = vec_a && (g[i].lda & 3) == 0 && (g[i].K & 3) == 0 && (reinterpret_cast<uintptr_t>(g[i].A) & 15u) == 0;
        vec_w = vec_w && (g[i].ldw & 3) == 0 && (g[i].K & 15) == 0 && (reinterpret_cast<uintptr_t>(g[i].W) & 15u) == 0;
    }
    bool ksplit = gemm_ksplit_enabled();
    for (int i = 0; i < count; ++i) ksplit = ksplit && g[i].K >= 256 && !g[i].A2 && !g[i].td.q;
    if (ksplit) {
        size_t lds = 0;
        for (int i = 0; i < count; ++i) {
            const size_t l = (size_t)kRows * ((g[i].K + 63) / 64 * 64 + 4) * sizeof(float) + 4 * 64 * sizeof(f32x4);
            lds = l > lds ? l : lds;
        }
        const dim3 grid((maxN + 15) / 16, (maxM + kRows - 1) / kRows, count);
        hipLaunchKernelGGL((mlp_gemm_ksplit_kernel<W_NK>), grid, dim3(kGemmThreads), lds, stream, all, (int)vec_a, (int)vec_w);
        RPO_LAUNCH_CHECK();
        return 0;
    }
    const int ch = maxK <= 16 ? 1 : maxK <= 64 ? 4 : 8;
    size_t lds = 0;
    for (int i = 0; i < count; ++i) {
        const int kw = (g[i].K + 16 * ch - 1) / (16 * ch) * (16 * ch), kw2 = g[i].A2 ? (g[i].K2 + 16 * ch - 1) / (16 * ch) * (16 * ch) : 0;
        const size_t l = (size_t)kRows * (kw + 4 + kw2 + 4) * sizeof(float);
        lds = l > lds ? l : lds;
    }
    const dim3 grid((maxN + kGemmCols - 1) / kGemmCols, (maxM + kRows - 1) / kRows, count);
    if (ch == 1) hipLaunchKernelGGL((mlp_gemm_kernel<W_NK, 1>), grid, dim3(kGemmThreads), lds, stream, all, (int)vec_a, (int)vec_w);
    else if (ch == 4) hipLaunchKernelGGL((mlp_gemm_kernel<W_NK, 4>), grid, dim3(kGemmThreads), lds, stream, all, (int)vec_a, (int)vec_w);
    else hipLaunchKernelGGL((mlp_gemm_kernel<W_NK, 8>), grid, dim3(kGemmThreads), lds, stream, all, (int)vec_a, (int)vec_w);
    RPO_LAUNCH_CHECK();
    return 0;
}

struct GemmFwd { Mlp net; int n; const float* s; int s_stride; const float* a; int a_stride; float* out; float* x0; float* h1; };

// (Round 4 also built the first layer INSIDE the hidden layer's launch -- every workgroup forming the x0 tile of its 16 rows
// itself, bit-equal -- and measured it slower in the EVOPF windows, 4.99 -> 4.74 M env-steps/s: the separate first-layer launch
// spreads its tiles over 64-128 workgroups, fused every wave walks 4-8 of them in front of the hidden layer's chain.  Removed;
// DESIGN.md 4b, commit "mlp_gemm: first layer inside the hidden layer's launch".)

// The layer-by-layer path applies to the wide networks at update-batch sizes when the caller provides x0 / h1 buffers
// (rpo_tuning(RPO_TUNE_MLP_GEMM, 0) keeps the row-tile kernels: the A/B switch of the tests).
static inline bool gemm_path_enabled() {
    return rpo_tune(RPO_TUNE_MLP_GEMM) != 0;                    // (read per call: the tests switch it inside one process)
}
static inline bool gemm_path_ok(const Mlp& net, int n, const void* x0, const void* h1, int out_mode) {
    return gemm_path_enabled() && net.E == 256 && net.H == 256 && x0 && h1 && out_mode == 0 && n <= 16384;
}

// whether `count` (<= 4) networks of this shape fit the problems of one launch (second heads / action halves are problems too)
static inline bool gemm_fits(const Mlp& net, int count) {
    return count <= 4 && count * (net.cat ? 2 : 1) <= kGemmProblems && count * (net.n_out > 1 ? 2 : 1) <= kGemmProblems;
}

// forward of `count` same-shaped networks: three launches (first layer | hidden layer | heads)

static inline int gemm_forward(const GemmFwd* f, int count, hipStream_t stream) {
    GemmArgs g1[4], g1b[4], g2[4], g3[4], g3b[4];
    const Mlp& n0 = f[0].net;
    const int ein = n0.cat ? 2 * n0.E : n0.E, heads = n0.hd > 1 ? n0.hd : 1;
    for (int i = 0; i < count; ++i) {
        const Mlp& net = f[i].net;
        GemmArgs z{};
        // x0[:, 0:E] = s Ws^T + bs (+ a Wa^T + ba for "add"); x0[:, E:2E] = a Wa^T + ba for "cat"
        g1[i] = z;
        g1[i].A = f[i].s; g1[i].lda = f[i].s_stride; g1[i].K = net.S; g1[i].W = net.Ws; g1[i].ldw = net.S; g1[i].bias = net.bs;
        g1[i].C = f[i].x0; g1[i].ldc = ein; g1[i].M = f[i].n; g1[i].N = net.E;
        if (net.A > 0 && !net.cat) {
            g1[i].A2 = f[i].a; g1[i].lda2 = f[i].a_stride; g1[i].K2 = net.A; g1[i].W2 = net.Wa; g1[i].ldw2 = net.A;
            g1[i].bias2 = net.ba;
        }
        g1b[i] = z;
        if (net.cat) {
            g1b[i].A = f[i].a; g1b[i].lda = f[i].a_stride; g1b[i].K = net.A; g1b[i].W = net.Wa; g1b[i].ldw = net.A;
            g1b[i].bias = net.ba; g1b[i].C = f[i].x0 + net.E; g1b[i].ldc = ein; g1b[i].M = f[i].n; g1b[i].N = net.E;
        }
        g2[i] = z;
        g2[i].A = f[i].x0; g2[i].lda = ein; g2[i].K = ein; g2[i].relu_a = 1; g2[i].W = net.W0; g2[i].ldw = ein; g2[i].bias = net.b0;
        g2[i].C = f[i].h1; g2[i].ldc = net.H; g2[i].M = f[i].n; g2[i].N = net.H;
        g3[i] = z;
        g3[i].A = f[i].h1; g3[i].lda = net.H; g3[i].K = net.H; g3[i].relu_a = 1; g3[i].W = net.W1; g3[i].ldw = net.H; g3[i].bias = net.b1;
        g3[i].C = f[i].out; g3[i].ldc = net.n_out * heads; g3[i].M = f[i].n; g3[i].N = heads;
        g3b[i] = g3[i];
        if (net.n_out > 1) { g3b[i].W = net.W1b; g3b[i].bias = net.b1b; g3b[i].C = f[i].out + heads; }
    }
    // (a second output head / the action half of a concatenating critic are more problems of the same launch)
    GemmArgs l1[kGemmProblems], l3[kGemmProblems];
    int c1 = 0, c3 = 0;
    if (!gemm_fits(n0, count)) return RPO_ERR_ARG;
    for (int i = 0; i < count; ++i) { l1[c1++] = g1[i]; if (n0.cat) l1[c1++] = g1b[i]; }
    for (int i = 0; i < count; ++i) { l3[c3++] = g3[i]; if (n0.n_out > 1) l3[c3++] = g3b[i]; }
    if (int e = gemm_launch<true>(l1, c1, stream)) return e;
    if (int e = gemm_launch<true>(g2, count, stream)) return e;
    return gemm_launch<true>(l3, c3, stream);
}


// rows part of the backward of `count` (1 or 2) same-shaped networks: [TD +] dh | dx0 | [da]
static inline int gemm_backward_rows(const BwdArgs* b, int count, hipStream_t stream) {
    const Mlp& n0 = b[0].net;
    const int ein = n0.cat ? 2 * n0.E : n0.E, heads = n0.hd > 1 ? n0.hd : 1, outs = n0.n_out * heads, n = b[0].n;
    GemmArgs gh[4], gx[4], ga[4];
    for (int i = 0; i < count; ++i) {
        const Mlp& net = b[i].net;
        GemmArgs z{};
        gh[i] = z;                                               // dh = (dout W1 [+ dout_b W1b]) * 1[h1 > 0]
        gh[i].A = b[i].dout; gh[i].lda = outs; gh[i].K = heads; gh[i].W = net.W1; gh[i].ldw = net.H;
        if (net.n_out > 1) { gh[i].A2 = b[i].dout + heads; gh[i].lda2 = outs; gh[i].K2 = heads; gh[i].W2 = net.W1b; gh[i].ldw2 = net.H; }
        gh[i].C = b[i].dh; gh[i].ldc = net.H; gh[i].mask = b[i].h1; gh[i].ldmask = net.H; gh[i].M = n; gh[i].N = net.H;
        gh[i].td = b[i].td;                                      // (critic update: dout = dLoss/dQ is produced by this launch)
        gx[i] = z;                                               // dx0 = (dh W0) * 1[x0 > 0]
        gx[i].A = b[i].dh; gx[i].lda = net.H; gx[i].K = net.H; gx[i].W = net.W0; gx[i].ldw = ein; gx[i].C = b[i].dx0; gx[i].ldc = ein;
        gx[i].mask = b[i].x0; gx[i].ldmask = ein; gx[i].M = n; gx[i].N = ein;
        ga[i] = z;                                               // da = dx0[:, action columns] Wa
        ga[i].A = b[i].dx0 + (net.cat ? net.E : 0); ga[i].lda = ein; ga[i].K = net.E; ga[i].W = net.Wa; ga[i].ldw = net.A;
        ga[i].C = b[i].da; ga[i].ldc = net.A; ga[i].M = n; ga[i].N = net.A;
    }
    if (int e = gemm_launch<false>(gh, count, stream)) return e;
    if (int e = gemm_launch<false>(gx, count, stream)) return e;
    if (b[0].da) return gemm_launch<false>(ga, count, stream);
    return 0;
}

}  // namespace rpo_mlp_dev
