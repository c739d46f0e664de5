// Column-split ("N-split") form of the MLP tile for the batch-256 update kernels (see nsplit.hip).
//
// mlp_tile.h gives one workgroup 16 rows through the WHOLE hidden layer: 128 f32 MFMAs per SIMD (~1.7 us of matrix
// pipe on one CU) behind a 128 KB weight stream, on 16 of the 256 CUs.  Here a workgroup of 2 waves owns 16 rows x 32
// hidden columns -- exactly the slab ONE wave of the row-tile kernel owns -- so a batch of 256 rows becomes
// 16 row tiles x 8 column groups = 128 workgroups per network, each with a 16 KB weight slice and a 32-MFMA chain.
// The price is that the head (a dot product over all hidden columns) cannot finish inside the launch: every workgroup
// leaves the per-row partial its slab contributes, and the CONSUMER adds the 8 partials in the fixed order of the
// row-tile kernel's wave loop.  Same arithmetic, same order -> bitwise the same outputs as mlp_tile.h
// (tests/test_mlp_gpu.py::test_split_forward_is_bitwise_the_tile_forward).
#pragma once
#include "mlp_tile.h"

namespace rpo_mlp_dev {

constexpr int kNsThreads = 128;            // 2 waves: wave w owns column tile c = w of the group's 32 columns
constexpr int kNsGroups = 8;               // column groups == waves of the row-tile forward (H / 32)

template <int EIN>
struct NsLds {
    __attribute__((aligned(16))) float x1[kRows * (EIN + 4)];   // relu(first layer) of the 16 rows
    float in_s[kRows * 8];                                        // state inputs (S <= 8)
    float in_a[kRows * 8];                                        // action inputs (A <= 8)
    float xch[2 * 4 * 64];                                        // wave 0 -> wave 1: head partials after column tile 0
};

// What a thread keeps in registers: its first-layer operands and its wave's 16 x EIN slice of W0.
// First layer on the matrix cores (round 5, as mlp_tile.h tile_l1_mfma): wave w owns first-layer columns [64 w, 64 w + 64) as
// four 16-column tiles, x0^T[e][row] = Wcat[e][u] in^T[u][row] with the k slots u = (bs x 1 | ba x 1 | state inputs | zero pad)
// in two steps, then (action inputs | zero pad) in one: wk[t][s] is this lane's A operand of step s.  The biases enter as the
// FIRST terms of the chain -- (0 + bs) + ba is the vector form's bias = bs + ba, exactly -- so this is its k-ordered fmaf chain
// (bias, state inputs in order, action inputs in order): no bit changes, on 12 registers instead of 17.  What was measured on
// the way (bench A/B on one box each): a separate accumulator initialiser (16 more registers) -- RPODDPG +3.4 %, RPOSAC -6 %
// (its riding launches went over 128 registers); the sum bs + ba formed where the weights are loaded, under a per-lane branch --
// the wait for those two loads lands in front of every other weight request: -2.5 .. -5 % on the riding workloads; a fourth
// step for wider inputs with its operands fetched where they are used -- the same wait, between the steps.  Hence: every
// operand ONE predicated load, no arithmetic on loaded values before the first layer, six state and four action inputs
// (split_ok; wider inputs keep the row-tile pipelines).
template <int EIN>
struct NsWeights {
    float wk[4][3];
    float4 w0[EIN / 16];
    float b0v, w1av, w1bv;
};

// A operand of first-layer column e at k slot u of the state steps (slots 0 / 1: the two biases) / at action input u
__device__ __forceinline__ float ns_l1_state_w(const Mlp& net, int e, int u) {
    const float* ptr = u == 0 ? net.bs + e : (u == 1 ? net.ba + e : net.Ws + e * net.S + (u - 2));
    const bool live = u == 0 || (u == 1 ? net.A > 0 : u - 2 < net.S);
    return live ? *ptr : 0.0f;
}
__device__ __forceinline__ float ns_l1_action_w(const Mlp& net, int e, int u) {
    return (net.A > 0 && u < net.A) ? net.Wa[e * net.A + u] : 0.0f;
}

// Issue every weight load of the slab (they return underneath the input staging / gather of the caller).
template <int EIN, int H>
__device__ __forceinline__ void ns_load_weights(const Mlp& net, int g, NsWeights<EIN>& w) {
    static_assert(EIN == kNsThreads, "one first-layer column per thread");
    const int tid = threadIdx.x & (kNsThreads - 1), lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;   // (two slab units may share a 256-thread workgroup)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = 64 * wave + 16 * t + li;                   // this lane's row of the A operand
        w.wk[t][0] = ns_l1_state_w(net, e, lg);
        w.wk[t][1] = ns_l1_state_w(net, e, 4 + lg);
        w.wk[t][2] = ns_l1_action_w(net, e, lg);
    }
    const int j = g * 32 + wave * 16 + li;                       // hidden column of this lane's B operand / outputs
#pragma unroll
    for (int it = 0; it < EIN / 16; ++it)
        w.w0[it] = *reinterpret_cast<const float4*>(&net.W0[(size_t)j * EIN + it * 16 + lg * 4]);
    w.b0v = net.b0[j];
    w.w1av = net.W1[j];
    w.w1bv = net.n_out > 1 ? net.W1b[j] : 0.0f;
}

// Hidden slab of column group g for the 16 rows staged in lds.in_s / lds.in_a (the caller wrote them; this function
// synchronises before reading).  Writes part[(g * n + row) * 2 + o] for rows < n (o < n_out), h1_save columns of the
// group, x0_save (group 0 only).  S <= 6, A <= 4 (split_ok), "add" critics / actors with EIN = E = 128.
// The state half of layer 1 (bias, then the state inputs in order) of the rows staged in lds.in_s: what ns_hidden starts
// with.  A caller whose ACTION inputs arrive late (a consumer inside a fused launch) runs this before it waits and hands
// the accumulators to ns_hidden (`pre`): same operations in the same order.  The caller synchronises before (staging).
#ifndef RPO_NS_SKIP
#define RPO_NS_SKIP 0              // timing-only builds: 1 = no first-layer fmaf loops in the column-split stages
#endif
// (acc1: the four C tiles of the lane, acc1[4 t + i] = x0[row li][64 w + 16 t + 4 lg + i] so far)
template <int EIN>
__device__ __forceinline__ void ns_layer1_state(const Mlp& net, const NsWeights<EIN>& w, const NsLds<EIN>& lds, float (&acc1)[kRows]) {
    const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
    // B operands: slot u of row li -- 1.0 for the two bias slots, state input u - 2, zero beyond S
    const float b0 = lg < 2 ? 1.0f : (lg - 2 < net.S ? lds.in_s[li * 8 + lg - 2] : 0.0f);
    const float b1 = 2 + lg < net.S ? lds.in_s[li * 8 + 2 + lg] : 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (!(RPO_NS_SKIP & 1)) {
            c = mfma4(w.wk[t][0], b0, c);
            c = mfma4(w.wk[t][1], b1, c);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc1[4 * t + i] = c[i];
    }
}

template <int EIN, int H>
__device__ __forceinline__ void ns_hidden(const Mlp& net, const NsWeights<EIN>& w, NsLds<EIN>& lds, int g, int row0, int n,
                                          float* part, float* x0_save, float* h1_save, const float* pre = nullptr,
                                          float (*hp)[4] = nullptr) {
    constexpr int LDX = EIN + 4;
    const int tid = threadIdx.x & (kNsThreads - 1), lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;   // (two slab units may share a 256-thread workgroup)
    __syncthreads();
    // ---- layer 1 (MFMA, see NsWeights): the state steps (or `pre`), then the action steps; same fmaf order as tile_compute
    float acc1[kRows];
    if (pre) {
#pragma unroll
        for (int r = 0; r < kRows; ++r) acc1[r] = pre[r];
    } else {
        ns_layer1_state<EIN>(net, w, lds, acc1);
    }
    const bool act = net.A > 0 && !(RPO_NS_SKIP & 1);
    const float b2 = (act && lg < net.A) ? lds.in_a[li * 8 + lg] : 0.0f;
    const bool x0_vec = (reinterpret_cast<uintptr_t>(x0_save) & 15u) == 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x4 c = f32x4{acc1[4 * t], acc1[4 * t + 1], acc1[4 * t + 2], acc1[4 * t + 3]};
        if (act) {
            c = mfma4(w.wk[t][2], b2, c);
        }
        const int e0 = 64 * wave + 16 * t + 4 * lg;              // c[i] = x0[row li][e0 + i]
        if (x0_save && g == 0 && row0 + li < n) {
            float* dst = x0_save + (size_t)(row0 + li) * EIN + e0;
            if (x0_vec) *reinterpret_cast<f32x4*>(dst) = c;
            else { dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; dst[3] = c[3]; }
        }
        *reinterpret_cast<f32x4*>(&lds.x1[li * LDX + e0]) =
            f32x4{fmaxf(c[0], 0.0f), fmaxf(c[1], 0.0f), fmaxf(c[2], 0.0f), fmaxf(c[3], 0.0f)};
    }
    __syncthreads();
    // ---- layer 2 (MFMA): one 16 x 16 output tile per wave, k-ordered chain of EIN / 4 instructions
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int it = 0; it < EIN / 16; ++it) {
        const float4 a4 = *reinterpret_cast<const float4*>(&lds.x1[li * LDX + it * 16 + lg * 4]);
        acc = mfma4(a4.x, w.w0[it].x, acc);
        acc = mfma4(a4.y, w.w0[it].y, acc);
        acc = mfma4(a4.z, w.w0[it].z, acc);
        acc = mfma4(a4.w, w.w0[it].w, acc);
    }
    // acc[i] = h1[row = 4 lg + i][col = 32 g + 16 wave + li] (before bias)
    const int col = g * 32 + wave * 16 + li;
    float hr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = acc[i] + w.b0v;
        const int row = row0 + lg * 4 + i;
        if (h1_save && row < n) h1_save[(size_t)row * H + col] = h;
        hr[i] = fmaxf(h, 0.0f);
    }
    // ---- head partial of the slab: the row-tile kernel's wave runs fmaf over its two column tiles in order, so wave 0
    //      hands its term to wave 1, which adds its own and reduces over the 16 lanes of a row
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lds.xch[(0 * 4 + i) * 64 + lane] = fmaf(hr[i], w.w1av, 0.0f);
            lds.xch[(1 * 4 + i) * 64 + lane] = fmaf(hr[i], w.w1bv, 0.0f);
        }
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = fmaf(hr[i], o == 0 ? w.w1av : w.w1bv, lds.xch[(o * 4 + i) * 64 + lane]);
                v = rpo_row16_sum_lane0(v);                      // (same association as the xor butterfly, at li == 0)
                const int row = row0 + lg * 4 + i;
                if (li == 0 && row < n && o < net.n_out) part[((size_t)g * n + row) * 2 + o] = v;
                if (hp) hp[o][i] = v;                           // (wave 1, li == 0: the partial of row 4 lg + i, output o)
            }
    }
}

// The consumer's side of the seam: head output o of `row` from the 8 slab partials (fixed order == the wave loop of
// tile_compute), bias first.
__device__ __forceinline__ float ns_head(const float* part, int n, int row, int o, float bias) {
    float v = bias;
#pragma unroll
    for (int g = 0; g < kNsGroups; ++g) v += part[((size_t)g * n + row) * 2 + o];
    return v;
}

}  // namespace rpo_mlp_dev
