// Large-batch forward of the 128 -> 256 scalar-head networks: WEIGHTS STATIONARY IN LDS, ROWS STREAM THROUGH, WAVES INDEPENDENT
// (round 5; reference work: rpo/algo/rpo_ddpg.py:163-205 at SURVEY 8d-iii's batch of 256 * N rows).
//
// The 64-row tile kernel (mlp_tile.h, RT = 4) gives every tile a workgroup that pulls W0 (128 KB) from L2 again, splits the
// hidden COLUMNS over its eight waves and meets at four barriers per tile: 0.63 of the f32 MFMA peak at 2^20 rows, the MFMA
// phase itself at ~0.7 (rounds 3-4).  Here one persistent workgroup per CU stages W0 ONCE into LDS (256 x 136 floats =
// 136 KB of the CU's 160 KB; row stride 136: the ds_read_b128 of the B operand -- lane (li, lg) reads W0[16 c + li][16 it +
// 4 lg ..+3] -- is conflict-free for 136 / 4 = 2 mod 16, the padding 132 of the row-tile kernels has a two-way conflict), and
// every WAVE owns whole 16-row tiles end to end:
//   * first layer on the vector ALU straight into the A-operand layout: lane (li, lg) forms x0[row li][16 it + 4 lg + m] for
//     the k-group it is about to feed (weights of the <= 11 inputs from a packed LDS table, the row's inputs in registers);
//   * hidden layer: all 16 column tiles of the row tile in 64 accumulator registers, B from LDS (one ds_read_b128 per four
//     MFMAs), consecutive MFMAs on four different accumulators;
//   * head, pre-activation stores and output from the accumulators.
// No barrier after the staging, no LDS traffic between waves: a SIMD's MFMA pipe is fed by three or four independent
// instruction streams.  Arithmetic = the row-tile kernels' to the bit: the same k-ordered MFMA chains from a zero
// accumulator, bias afterwards; the first layer's fmaf order (bias, state inputs, action inputs); the head as eight
// 32-column partial sums (the old waves' shares: fma over the two tiles, quad / row butterfly, added to the bias in wave
// order).  (Inputs beyond S + A are zero-weighted zeros: fmaf(0, 0, x) == x except for x == -0.0 -> +0.0, a sign of zero.)
#pragma once
#include "mlp_tile.h"

namespace rpo_mlp_dev {

constexpr int kStreamLdW = 136;        // floats between rows of W0 in LDS
constexpr int kStreamIn = 11;          // S + A <= 11 inputs; slot 11 of a first-layer table row holds the bias

template <int H>
struct StreamLds {
    __attribute__((aligned(16))) float w0[H * kStreamLdW];
    __attribute__((aligned(16))) float fl[128 * 12];            // [e][w_0 .. w_10 | bias]
    __attribute__((aligned(16))) float hb[H * 4];               // [col][b0, W1, W1b, 0]: one ds_read_b128 per column tile of the epilogue
};

static inline bool stream_shape_ok(const Mlp& net) {
    return !net.cat && net.E == 128 && net.H == 256 && net.hd <= 1 && net.S + net.A <= kStreamIn && net.S > 0;
}

// One 16-row tile, start to finish, by one wave.  FULL: every row of the tile exists (no guards: a guarded store is a branch
// of its own, 80 per tile); otherwise the inputs come from a clamped row and the stores are guarded.
template <int H, bool FULL, int CH, class ARGS>
__device__ __forceinline__ void stream_tile(const ARGS& p, const StreamLds<H>& lds, int row0, int li, int lg, float b1a, float b1b) {
    constexpr int EIN = 128;
    const Mlp& net = p.net;
    const int nin = net.S + net.A;
    const bool two = net.n_out > 1;
    const int row = row0 + li;
    const bool live = FULL || row < p.n;
    const int rc = FULL ? row : (row < p.n ? row : p.n - 1);
    // ---- the row's inputs (wave-uniform selects of the source; no divergent control flow)
    const float* sp = p.s + (size_t)rc * p.s_stride;
    const float* ap = net.A > 0 ? p.a + (size_t)rc * p.a_stride : sp;
    float in[kStreamIn];
#pragma unroll
    for (int u = 0; u < kStreamIn; ++u) {
        const bool is_s = u < net.S, is_a = !is_s && u < nin;
        const float* q = is_s ? sp + u : (is_a ? ap + (u - net.S) : sp);
        const float v = *q;
        in[u] = (is_s || is_a) ? v : 0.0f;
    }
    f32x4 acc[H / 16];
#pragma unroll
    for (int c = 0; c < H / 16; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int it = 0; it < EIN / 16; ++it) {                      // (rolled: fully unrolled, the scheduler hoisted every LDS read of
        // ---- first layer for the four k's this lane feeds: e = 16 it + 4 lg + m            the tile and spilled 700 registers)
        float x0v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float* wrow = &lds.fl[(it * 16 + lg * 4 + m) * 12];
            const float4 wA = *reinterpret_cast<const float4*>(wrow);
            const float4 wB = *reinterpret_cast<const float4*>(wrow + 4);
            const float4 wC = *reinterpret_cast<const float4*>(wrow + 8);
            float a1 = wC.w;
            a1 = fmaf(in[0], wA.x, a1); a1 = fmaf(in[1], wA.y, a1); a1 = fmaf(in[2], wA.z, a1); a1 = fmaf(in[3], wA.w, a1);
            a1 = fmaf(in[4], wB.x, a1); a1 = fmaf(in[5], wB.y, a1); a1 = fmaf(in[6], wB.z, a1); a1 = fmaf(in[7], wB.w, a1);
            a1 = fmaf(in[8], wC.x, a1); a1 = fmaf(in[9], wC.y, a1); a1 = fmaf(in[10], wC.z, a1);
            x0v[m] = a1;
        }
        if (p.x0_save && live)
            __builtin_nontemporal_store(f32x4{x0v[0], x0v[1], x0v[2], x0v[3]},
                                        reinterpret_cast<f32x4*>(&p.x0_save[(size_t)row * EIN + it * 16 + lg * 4]));
        float a4[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a4[m] = fmaxf(x0v[m], 0.0f);
        // ---- hidden layer: CH column tiles at a time (4; 2 in the 16-wave form, whose budget is 128 registers: an accumulator is
        // revisited every CH MFMAs = 64+ cycles >= the 40 of a dependent pair), consecutive MFMAs on different accumulators; the B operands of
        // chunk cc + 1 are requested BEFORE the MFMAs of chunk cc (register ping-pong; the scheduling barriers pin that order:
        // left alone the compiler issues the reads behind the chunk's last MFMAs, or -- unrolled -- all of them up front)
        const float* wb = &lds.w0[li * kStreamLdW + it * 16 + lg * 4];
        float4 bcur[CH], bnxt[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) bcur[q] = *reinterpret_cast<const float4*>(wb + 16 * q * kStreamLdW);
#pragma unroll
        for (int cc = 0; cc < H / 16; cc += CH) {
            if (cc + CH < H / 16) {
#pragma unroll
                for (int q = 0; q < CH; ++q) bnxt[q] = *reinterpret_cast<const float4*>(wb + 16 * (cc + CH + q) * kStreamLdW);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(a4[0], bcur[q].x, acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(a4[1], bcur[q].y, acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(a4[2], bcur[q].z, acc[cc + q]);
#pragma unroll
            for (int q = 0; q < CH; ++q) acc[cc + q] = mfma4(a4[3], bcur[q].w, acc[cc + q]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < CH; ++q) bcur[q] = bnxt[q];
        }
    }
    // ---- epilogue: acc[c][i] = h1[row0 + 4 lg + i][16 c + li] before the bias
    float v0[4] = {b1a, b1a, b1a, b1a}, v1[4] = {b1b, b1b, b1b, b1b};
    float* h1p = p.h1_save ? p.h1_save + (size_t)(row0 + lg * 4) * H + li : nullptr;
#pragma unroll
    for (int w8 = 0; w8 < H / 32; ++w8) {                        // the 32-column shares of the row-tile kernels' eight waves
        float p0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, p1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 2 * w8; c < 2 * w8 + 2; ++c) {
            const float4 hb = *reinterpret_cast<const float4*>(&lds.hb[(16 * c + li) * 4]);
            const float b0 = hb.x, wa = hb.y, wb = hb.z;
            float h[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) h[i] = acc[c][i] + b0;
            if (h1p) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (FULL || row0 + lg * 4 + i < p.n) __builtin_nontemporal_store(h[i], h1p + (size_t)i * H + 16 * c);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float hr = fmaxf(h[i], 0.0f);
                p0[i] = fmaf(hr, wa, p0[i]);
                p1[i] = fmaf(hr, wb, p1[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) v0[i] += rpo_row16_sum_lane0(p0[i]);
        if (two) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v1[i] += rpo_row16_sum_lane0(p1[i]);
        }
    }
    if (li == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + lg * 4 + i;
            if (FULL || r < p.n) {
                float o0 = v0[i];
                if (p.out_mode == 1) o0 = p.scale * tanhf(o0) + p.base;
                p.out[(size_t)r * net.n_out] = o0;
                if (two) p.out[(size_t)r * net.n_out + 1] = v1[i];
            }
        }
    }
}

// FwdArgs is declared by the includer (mlp.hip); the kernel takes the FwdArgs4 of the multi-network launches
template <int H, int NW, class ARGS4>
__device__ __forceinline__ void mlp_forward_stream_body(const ARGS4& p4) {
    constexpr int EIN = 128;
    __shared__ StreamLds<H> lds;
    const auto& p = p4.net[blockIdx.y];
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nin = net.S + net.A;
    // ---- staging, once per workgroup
    for (int idx = tid; idx < H * (EIN / 4); idx += NW * 64) {
        const int j = idx / (EIN / 4), q = idx - j * (EIN / 4);
        *reinterpret_cast<float4*>(&lds.w0[j * kStreamLdW + q * 4]) =
            *reinterpret_cast<const float4*>(&net.W0[(size_t)j * EIN + q * 4]);
    }
    for (int idx = tid; idx < EIN * 12; idx += NW * 64) {
        const int e = idx / 12, u = idx - e * 12;
        float v = 0.0f;
        if (u < net.S) v = net.Ws[e * net.S + u];
        else if (u < nin) v = net.Wa[e * net.A + (u - net.S)];
        else if (u == 11) v = net.A > 0 ? net.bs[e] + net.ba[e] : net.bs[e];
        lds.fl[idx] = v;
    }
    for (int idx = tid; idx < H; idx += NW * 64)
        *reinterpret_cast<float4*>(&lds.hb[idx * 4]) =
            make_float4(net.b0[idx], net.W1[idx], net.n_out > 1 ? net.W1b[idx] : 0.0f, 0.0f);
    __syncthreads();
    const float b1a = net.b1[0], b1b = net.n_out > 1 ? net.b1b[0] : 0.0f;
    const int tiles = (p.n + kRows - 1) / kRows;
    for (int t = blockIdx.x * NW + wave; t < tiles; t += gridDim.x * NW) {
        const int row0 = t * kRows;
        constexpr int CH = NW > 12 ? 2 : 4;
        if (row0 + kRows <= p.n) stream_tile<H, true, CH>(p, lds, row0, li, lg, b1a, b1b);
        else stream_tile<H, false, CH>(p, lds, row0, li, lg, b1a, b1b);
    }
}

}  // namespace rpo_mlp_dev
