// Large-batch backward of the 128 -> 256 scalar-head networks in the style of mlp_stream.h (round 5): persistent workgroups,
// waves that own whole 16-row tiles, matrix work on MFMA in layouts that need no transposition, as few vector instructions as
// the arithmetic allows (v_mfma_f32_16x16x4_f32 shares the vector FMA lanes: vector instructions ADD to the MFMA time).
//
// The one-pass kernel of round 3 (mlp_bwd.h) walks a slice of the batch with two groups of four waves that meet at three
// barriers per tile and do ~1.5 k vector instructions per tile per wave (dh formation thread-per-column, first-layer sums,
// LDS staging): 0.57 of the f32 MFMA peak.  Here the backward is TWO launches over the saved activations (h1 and x0 are read
// twice, 3 GB at 2^20 rows: 0.5 ms of HBM time under 1.1 ms of MFMA work):
//
//   rows kernel (bwd_stream_rows_kernel): W0 stationary in LDS ([j][e], row stride 128: the ds_read_b128 of lane (li, lg) at
//     W0[j][64 h + 4 li ..+3] is conflict-free exactly when 4 rows are a multiple of 64 floats), one wave per 16-row tile:
//       A = mask (.) W1 for the tile, from h1 loaded as 16-byte chunks in the forward's C layout (lane (li, lg): row li,
//           columns 16 jt + 4 lg + m);  dx0 = dout * (A W0) * 1[x0 > 0] on 8 interleaved output tiles (column e = 64 h + 4 n + c:
//           one ds_read_b128 of W0 feeds 4 MFMAs);  the C layout of dx0 -- lane (li, lg): rows 4 lg + i, e = 64 h + 4 li + c --
//       is the A-operand layout of the first-layer gradients  d[Ws | Wa | b]^T[e][u] += dx0^T[e][row] in[row][u]  (K = rows in
//       the order i, lg; u = 11 is a column of ones: the bias gradient), so those are 32 more MFMAs per tile on the registers
//       dx0 lands in.  Optionally da = dx0 Wa (the policy step's dQ/da).  Two-head networks (the Gaussian actor) form
//       dh = mask (da_ w1a + db_ w1b) on the vector ALU instead of scaling at the end.
//   weights kernel (bwd_stream_weights_kernel): dW0[j][e] += dh[row][j] relu(x0)[row][e] with K = rows: eight waves own the
//       256 x 128 result as 4 x 4 interleaved tiles each (64 accumulator registers; lane (li, lg) loads h1[row 4 ks + lg][64 jq +
//       4 li ..+3] and x0[row][64 eh + 4 li ..+3]: one 16-byte load per operand feeds 16 MFMAs), two such sets per workgroup on
//       alternating tiles; db0 / dW1 / db1 as per-lane partial sums on the side.
// Every workgroup is one slice of the split-K plan (mlp_bwd.h): its sums go into its copy of the gradient span, added by
// splitk_reduce_kernel in slice order -- one owner per element, fixed orders: bitwise reproducible.  Not bitwise the two-pass
// result (other associations; the single-head rows kernel multiplies by dout after the k-sum): the tests hold it to the plain
// pass at 1e-5 of a tensor's largest entry like the one-pass kernel.
#pragma once
#include "mlp_bwd.h"

namespace rpo_mlp_dev {

#ifndef RPO_BWDS_SKIP
#define RPO_BWDS_SKIP 0            // timing-only builds (tools/probe/build_stream_variants.sh): weights kernel 1 no loads, 2 no MFMAs;
#endif                             // rows kernel 4 no h1 loads, 8 no epilogue (x0 loads, first-layer gradients) -- wrong results
constexpr int kBwdStreamWaves = 16;

struct BwdStreamLds {
    __attribute__((aligned(16))) float w0[256 * 128];           // [j][e], no padding (see above)
    __attribute__((aligned(16))) float w1a[256], w1b[256];
    __attribute__((aligned(16))) float wa[128 * 8];             // Wa[e][u'] for da (<= 8 action inputs)
};

__device__ __forceinline__ float bwd_xlane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

// TD target + Huber loss of every row (the prologue of the rows passes, as a launch of its own in front of the two kernels):
// dq_out[i] and the 16-row tiles' loss shares, the same functions and the same order as td prologue of mlp_bwd_rows_body.
static __global__ __launch_bounds__(256) void bwd_stream_td_kernel(TdArgs t, int n) {   // (static: not a symbol of the shared library)
    const int i = blockIdx.x * 256 + threadIdx.x;
    float dq = 0.0f, hub = 0.0f;
    if (i < n) {
        const float qn = rpo_head_dev::td_next_value(t.qn1[i], t.qn2 ? t.qn2[i] : 0.0f, t.qn2 != nullptr,
                                                     t.logp ? t.logp[i] : 0.0f, t.logp != nullptr, t.alpha);
        const float y = rpo_head_dev::td_target(t.reward[(size_t)i * t.reward_stride], t.done[(size_t)i * t.done_stride], t.gamma, qn);
        dq = rpo_head_dev::td_huber_row(t.q[i], y, 1.0f / (float)n, &hub);
        t.dq_out[i] = dq;
    }
    const float sum = rpo_row16_sum_desc_lane0(hub);             // (lanes beyond n hold 0)
    if ((threadIdx.x & 15) == 0 && (i >> 4) < (n + 15) / 16) t.loss_partial[i >> 4] = sum;
}

// ------------------------------------------------------------------------------------------------------------ rows kernel
// GRADS: 0 none, 1 every first-layer gradient, 2 state part only (shared embedding inside the actor loss).  TWO: second head.
template <int GRADS, bool TWO, bool WANT_DA>
__global__ __launch_bounds__(kBwdStreamWaves * 64, kBwdStreamWaves / 4) void bwd_stream_rows_kernel(BwdArgs p, SplitK k) {
    constexpr int EIN = 128, H = 256, NW = kBwdStreamWaves;
    __shared__ BwdStreamLds lds;
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the tile arithmetic stays on the scalar unit)
    const int nin = net.S + net.A;
    for (int idx = tid; idx < H * EIN / 4; idx += NW * 64)
        reinterpret_cast<float4*>(lds.w0)[idx] = reinterpret_cast<const float4*>(net.W0)[idx];
    for (int idx = tid; idx < H; idx += NW * 64) {
        lds.w1a[idx] = net.W1[idx];
        lds.w1b[idx] = TWO ? net.W1b[idx] : 0.0f;
    }
    if (WANT_DA)
        for (int idx = tid; idx < EIN * 8; idx += NW * 64) lds.wa[idx] = (idx & 7) < net.A ? net.Wa[(idx >> 3) * net.A + (idx & 7)] : 0.0f;
    __syncthreads();
    const int z = blockIdx.x, Z = gridDim.x;
    const long long tiles = ((long long)p.n + kRows - 1) / kRows;
    const long long t_lo = tiles * z / Z, t_hi = tiles * (z + 1) / Z;
    const int outs = TWO ? 2 : 1;
    f32x4 g[2][4];                                               // first-layer gradients: [e = 64 h + 4 (4 lg + i) + c][u = li]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) g[h][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    typedef const f32x4 __attribute__((address_space(3))) * lds_f4;
    typedef const float __attribute__((address_space(3))) * lds_f;
    unsigned w1p = (unsigned)(__UINTPTR_TYPE__)((lds_f)(&lds.w1a[0])) + lg * 16;
    asm volatile("" : "+v"(w1p));                                // (one base register, immediate offsets: see mlp_stream.h)
    // (the first h1 chunk of a tile is requested during the PREVIOUS tile's epilogue: at the tile's start the load would expose
    //  a full memory latency per tile)
    auto h1_row = [&](long long t) {
        const long long r = t * kRows + li;
        return p.h1 + (size_t)(r < p.n ? r : (long long)p.n - 1) * H + lg * 4;
    };
    f32x4 hfirst = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (t_lo + wave < t_hi) hfirst = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h1_row(t_lo + wave)));
    for (long long t = t_lo + wave; t < t_hi; t += NW) {
        const long long row0 = t * kRows;
        const long long rowA = row0 + li < p.n ? row0 + li : (long long)p.n - 1;       // this lane's row of the h1 chunks
        const float* h1p = p.h1 + (size_t)rowA * H + lg * 4;
        float doa = 0.0f, dob = 0.0f;                            // two-head form: the head gradients of row li
        if (TWO) { doa = p.dout[(size_t)rowA * 2]; dob = p.dout[(size_t)rowA * 2 + 1]; }
        f32x4 acc[2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[h][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 hcur = hfirst;
#pragma unroll 1
        for (int jt = 0; jt < H / 16; ++jt) {
            const int jn = jt + 1 < H / 16 ? jt + 1 : jt;        // (no branch around the prefetch: the last chunk is read twice)
            const f32x4 hnext = (RPO_BWDS_SKIP & 4) ? f32x4{1.0f, -1.0f, 2.0f, (float)jn} : __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h1p + 16 * jn));
            __builtin_amdgcn_sched_barrier(0);                   // (the request stays HERE: the scheduler otherwise sinks it behind the
                                                                 //  chunk's MFMAs to reuse hcur's registers -- a full latency per chunk)
            const f32x4 wv = *(lds_f4)(__UINTPTR_TYPE__)(w1p + 64 * jt);
            float a4[4];
            if (TWO) {
                const f32x4 wv2 = *(lds_f4)(__UINTPTR_TYPE__)(w1p + 64 * jt + 4 * H);
#pragma unroll
                for (int m = 0; m < 4; ++m) a4[m] = hcur[m] > 0.0f ? fmaf(dob, wv2[m], doa * wv[m]) : 0.0f;
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) a4[m] = hcur[m] > 0.0f ? wv[m] : 0.0f;
            }
            const float* wrow = &lds.w0[(16 * jt + 4 * lg) * EIN + 4 * li];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 b0 = *reinterpret_cast<const float4*>(wrow + m * EIN);
                const float4 b1 = *reinterpret_cast<const float4*>(wrow + m * EIN + 64);
                acc[0][0] = mfma4(a4[m], b0.x, acc[0][0]);
                acc[0][1] = mfma4(a4[m], b0.y, acc[0][1]);
                acc[0][2] = mfma4(a4[m], b0.z, acc[0][2]);
                acc[0][3] = mfma4(a4[m], b0.w, acc[0][3]);
                acc[1][0] = mfma4(a4[m], b1.x, acc[1][0]);
                acc[1][1] = mfma4(a4[m], b1.y, acc[1][1]);
                acc[1][2] = mfma4(a4[m], b1.z, acc[1][2]);
                acc[1][3] = mfma4(a4[m], b1.w, acc[1][3]);
            }
            hcur = hnext;
        }
        hfirst = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(h1_row(t + NW < t_hi ? t + NW : t)));
        if (RPO_BWDS_SKIP & 8) {
            float tsum = 0.0f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c) tsum += acc[h][c][0] + acc[h][c][1] + acc[h][c][2] + acc[h][c][3];
            if (tsum == 12345.678f) g[0][0][0] += tsum;
            continue;
        }
        // ---- dx0[row 4 lg + i][e = 64 h + 4 li + c] = dout * acc * 1[x0 > 0]; rows beyond n contribute zeros
        float dsc[4];
        long long rowC[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = row0 + 4 * lg + i;
            rowC[i] = r < p.n ? r : (long long)p.n - 1;
            dsc[i] = r < p.n ? (TWO ? 1.0f : p.dout[(size_t)rowC[i] * outs]) : 0.0f;
        }
        // k-step i of the first-layer gradients holds rows {i, 4 + i, 8 + i, 12 + i} in its slots lg: d[W | b]^T[e][u] +=
        // dx0^T[e][row] in[row][u]; one e-half at a time (16 live registers of dx0 instead of 32)
        float inv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (GRADS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long long r = rowC[i];
                if (li < net.S) inv[i] = p.s[(size_t)r * p.s_stride + li];
                else if (GRADS == 1 && li < nin) inv[i] = p.a[(size_t)r * p.a_stride + (li - net.S)];
                else if (li == 11) inv[i] = 1.0f;
            }
        }
        float pa[4][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};   // da partials (A <= 2)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 dxv[4];                                        // [i] -> components c
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.x0 + (size_t)rowC[i] * EIN + 64 * h + 4 * li));
#pragma unroll
                for (int c = 0; c < 4; ++c) dxv[i][c] = x[c] > 0.0f ? acc[h][c][i] * dsc[i] : 0.0f;
            }
            if (GRADS) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) g[h][c] = mfma4(dxv[i][c], inv[i], g[h][c]);
            }
            if (WANT_DA) {                                       // da[row][u'] = sum_e dx0[row][e] Wa[e][u']: lane-local part
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float2 w = *reinterpret_cast<const float2*>(&lds.wa[(64 * h + 4 * li + c) * 8]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pa[i][0] = fmaf(dxv[i][c], w.x, pa[i][0]);
                        pa[i][1] = fmaf(dxv[i][c], w.y, pa[i][1]);
                    }
                }
            }
        }
        if (WANT_DA) {                                           // ... then the 16 lanes of the row group
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float v = rpo_row16_sum_lane0(pa[i][u]);
                    const long long r = row0 + 4 * lg + i;
                    if (li == 0 && u < net.A && r < p.n) p.da[(size_t)r * net.A + u] = v;
                }
        }
    }
    if (!GRADS) return;
    // ---- the workgroup's first-layer gradients: waves in order through LDS (W0 is no longer needed), into the slice's copy
    __syncthreads();
    float* red = lds.w0;                                         // [wave][e][16]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(wave * EIN + 64 * h + 4 * (4 * lg + i) + c) * 16 + li] = g[h][c][i];
    __syncthreads();
    float* base = k.scratch + (long long)z * k.stride;
    for (int idx = tid; idx < EIN * 16; idx += NW * 64) {
        const int e = idx >> 4, u = idx & 15;
        float tot = red[idx];
#pragma unroll
        for (int w = 1; w < NW; ++w) tot += red[w * EIN * 16 + idx];
        if (u < net.S) base[(p.g.Ws - k.lo) + e * net.S + u] = tot;
        else if (GRADS == 1 && u < nin) base[(p.g.Wa - k.lo) + e * net.A + (u - net.S)] = tot;
        else if (u == 11) {
            base[(p.g.bs - k.lo) + e] = tot;
            if (GRADS == 1 && net.A > 0) base[(p.g.ba - k.lo) + e] = tot;
        }
    }
}

// --------------------------------------------------------------------------------------------------------- weights kernel

// Eight waves per workgroup (two per SIMD, <= 256 registers): wave (set, jq) owns dW0[64 jq .. 64 jq + 63][all 128 e] of its
// set's tiles as 4 x 8 interleaved tiles (128 accumulator registers): per k-step ONE 16-byte load of h1 and two of x0 feed 32
// MFMAs.  (The first cut -- sixteen waves with 4 x 4 tiles, two loads per 16 MFMAs, every x0 chunk read by four waves and
// every h1 chunk by two -- ran at 0.63 of the peak: 1.9 vector instructions and 0.19 load instructions per MFMA; this
// form has 1.1 and 0.09.)
constexpr int kBwdWeightsWaves = 8;

template <bool TWO>
__global__ __launch_bounds__(kBwdWeightsWaves * 64, kBwdWeightsWaves / 4) void bwd_stream_weights_kernel(BwdArgs p, SplitK k) {
    constexpr int EIN = 128, H = 256;
    __shared__ __attribute__((aligned(16))) float red[H * EIN];  // the second set's dW0 at the end
    __shared__ float vec[2][3][H];                               // [set][db0 | dW1a | dW1b][j]
    __shared__ float sc[2][2];                                   // [set][db1a | db1b]
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: the tile arithmetic below stays on the scalar unit)
    const int set = wave >> 2, jq = wave & 3;
    const int z = blockIdx.x, Z = gridDim.x;
    const long long tiles = ((long long)p.n + kRows - 1) / kRows;
    const long long t_lo = tiles * z / Z, t_hi = tiles * (z + 1) / Z;
    constexpr int outs = TWO ? 2 : 1;
    const int j0 = 64 * jq + 4 * li, e0 = 4 * li;
    const f32x4 w1a = *reinterpret_cast<const f32x4*>(net.W1 + j0);
    f32x4 w1b = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (TWO) w1b = *reinterpret_cast<const f32x4*>(net.W1b + j0);
    f32x4 acc[4][8];                                             // [cj][4 eh + ce]: see the store below
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 gb0 = f32x4{0.0f, 0.0f, 0.0f, 0.0f}, gw1 = gb0, gw2 = gb0;      // per-lane partial sums over its rows (slot lg)
    float gb1a = 0.0f, gb1b = 0.0f;
    // one k-step = 4 rows (slot lg): 32 MFMAs on the operands of three 16-byte loads
    auto step = [&](const f32x4& hv, const f32x4& xa, const f32x4& xb, float da_, float db_) {
        f32x4 dh;
        float x1[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = TWO ? fmaf(db_, w1b[c], da_ * w1a[c]) : da_ * w1a[c];
            dh[c] = hv[c] > 0.0f ? d : 0.0f;
            x1[c] = rpo_relu_bits(xa[c]);
            x1[4 + c] = rpo_relu_bits(xb[c]);
        }
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int ce = 0; ce < 8; ++ce) {
                if (RPO_BWDS_SKIP & 2) acc[cj][ce][0] += dh[cj] + x1[ce];
                else acc[cj][ce] = mfma4(dh[cj], x1[ce], acc[cj][ce]);
            }
        gb0 += dh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float hr = rpo_relu_bits(hv[c]);
            gw1[c] = fmaf(da_, hr, gw1[c]);
            if (TWO) gw2[c] = fmaf(db_, hr, gw2[c]);
        }
        if (jq == 0 && li == 0) { gb1a += da_; gb1b += db_; }
    };
    // this set's tiles: t_lo + set, + 2, ...; the FULL ones (every row exists) run through a ring of four k-steps whose loads
    // are issued without a branch from a scalar base + a per-lane constant offset; a ragged last tile takes the guarded path
    const long long n_t = t_hi > t_lo + set ? (t_hi - t_lo - set + 1) / 2 : 0;
    const bool ragged = n_t > 0 && (t_lo + set + 2 * (n_t - 1) + 1) * kRows > p.n;
    const int n_steps = (int)((n_t - (ragged ? 1 : 0)) * 4);      // (32-bit: the scalar unit has no ordered 64-bit compare)
    const int tile0 = (int)(t_lo + set);
    // (addresses = a SCALAR row base + a 32-bit per-lane byte offset: the loads take the base from scalar registers, no vector
    //  address arithmetic between the MFMAs)
    unsigned ho = 4u * (unsigned)(lg * H + j0), xo = 4u * (unsigned)(lg * EIN + e0), dof = 4u * (unsigned)(lg * outs);
    auto load = [&](int s, f32x4& hv, f32x4& xa, f32x4& xb, float& da_, float& db_) {
        const int sc2 = s < n_steps ? s : n_steps - 1;           // (scalar clamp: the last loads are repeated, not skipped)
        const long long rb = (long long)(tile0 + 2 * (sc2 >> 2)) * kRows + 4 * (sc2 & 3);   // scalar row base
        if (RPO_BWDS_SKIP & 1) {
            const float f = (float)(rb & 7) + (float)li;
            hv = f32x4{f, -f, f, f}; xa = f32x4{f, f, -f, f}; xb = xa; da_ = f; db_ = f;
            return;
        }
        const char* hrow = reinterpret_cast<const char*>(p.h1 + (size_t)rb * H);
        const char* xrow = reinterpret_cast<const char*>(p.x0 + (size_t)rb * EIN);
        const char* drow = reinterpret_cast<const char*>(p.dout + (size_t)rb * outs);
        asm volatile("" : "+v"(ho), "+v"(xo), "+v"(dof));        // opaque (no instruction): keeps `scalar base + 32-bit lane
                                                                 // offset` at the load -- hoisted, it is a 64-bit vector add per load
        hv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(hrow + ho));
        xa = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow + xo));
        xb = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow + xo + 256u));
        da_ = *reinterpret_cast<const float*>(drow + dof);
        db_ = TWO ? *reinterpret_cast<const float*>(drow + dof + 4u) : 0.0f;
    };
    if (n_steps > 0) {
        constexpr int RING = 4;
        f32x4 hq[RING], xaq[RING], xbq[RING];
        float daq[RING], dbq[RING];
#pragma unroll
        for (int b = 0; b < RING - 1; ++b) load(b, hq[b], xaq[b], xbq[b], daq[b], dbq[b]);
        for (int s = 0; s < n_steps; s += RING) {
#pragma unroll
            for (int b = 0; b < RING; ++b) {                     // three steps of loads in flight behind the MFMAs of the current one
                                                                 // (n_steps is a multiple of RING: four k-steps per tile)
                load(s + b + RING - 1, hq[(b + RING - 1) % RING], xaq[(b + RING - 1) % RING], xbq[(b + RING - 1) % RING],
                     daq[(b + RING - 1) % RING], dbq[(b + RING - 1) % RING]);
                step(hq[b], xaq[b], xbq[b], daq[b], dbq[b]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (ragged) {
        const long long t = t_lo + set + 2 * (n_t - 1);
        for (int ks = 0; ks < 4; ++ks) {
            const long long r = t * kRows + 4 * ks + lg, rc = r < p.n ? r : (long long)p.n - 1;
            const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h1 + (size_t)rc * H + j0);
            const f32x4 xa = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)rc * EIN + e0);
            const f32x4 xb = *reinterpret_cast<const f32x4*>(p.x0 + (size_t)rc * EIN + e0 + 64);
            const float da_ = r < p.n ? p.dout[(size_t)rc * outs] : 0.0f;
            const float db_ = (TWO && r < p.n) ? p.dout[(size_t)rc * outs + 1] : 0.0f;
            step(hv, xa, xb, da_, db_);
        }
    }
    // ---- results.  acc[cj][4 eh + ce][i] = dW0[j = 64 jq + 4 (4 lg + i) + cj][e = 64 eh + 4 li + ce]: the second set through
    // LDS, the first adds and writes 16-byte chunks along e into the slice's copy of the gradient span
    float* base = k.scratch + (long long)z * k.stride;
    if (set == 1) {
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh)
                    *reinterpret_cast<f32x4*>(&red[(64 * jq + 4 * (4 * lg + i) + cj) * EIN + 64 * eh + e0]) =
                        f32x4{acc[cj][4 * eh][i], acc[cj][4 * eh + 1][i], acc[cj][4 * eh + 2][i], acc[cj][4 * eh + 3][i]};
    }
    // small vectors: the four slot lanes (lg) of a column, then the sets
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s0 = gb0[c] + bwd_xlane(gb0[c], lane ^ 16);
        s0 = s0 + bwd_xlane(s0, lane ^ 32);
        float s1 = gw1[c] + bwd_xlane(gw1[c], lane ^ 16);
        s1 = s1 + bwd_xlane(s1, lane ^ 32);
        if (lg == 0) { vec[set][0][j0 + c] = s0; vec[set][1][j0 + c] = s1; }
        if (TWO) {
            float s2 = gw2[c] + bwd_xlane(gw2[c], lane ^ 16);
            s2 = s2 + bwd_xlane(s2, lane ^ 32);
            if (lg == 0) vec[set][2][j0 + c] = s2;
        }
    }
    if (jq == 0 && li == 0) {
        float s1 = gb1a + bwd_xlane(gb1a, lane ^ 16);
        s1 = s1 + bwd_xlane(s1, lane ^ 32);
        float s2 = gb1b + bwd_xlane(gb1b, lane ^ 16);
        s2 = s2 + bwd_xlane(s2, lane ^ 32);
        if (lg == 0) { sc[set][0] = s1; sc[set][1] = s2; }
    }
    __syncthreads();
    if (set == 0) {
        float* dst = base + (p.g.W0 - k.lo);
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
                    const int j = 64 * jq + 4 * (4 * lg + i) + cj;
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&red[j * EIN + 64 * eh + e0]);
                    *reinterpret_cast<f32x4*>(dst + (size_t)j * EIN + 64 * eh + e0) =
                        f32x4{acc[cj][4 * eh][i] + o[0], acc[cj][4 * eh + 1][i] + o[1], acc[cj][4 * eh + 2][i] + o[2],
                              acc[cj][4 * eh + 3][i] + o[3]};
                }
    }
    if (tid < H) {
        base[(p.g.b0 - k.lo) + tid] = vec[0][0][tid] + vec[1][0][tid];
        base[(p.g.W1 - k.lo) + tid] = vec[0][1][tid] + vec[1][1][tid];
        if (TWO) base[(p.g.W1b - k.lo) + tid] = vec[0][2][tid] + vec[1][2][tid];
    }
    if (tid == 0) {
        base[p.g.b1 - k.lo] = sc[0][0] + sc[1][0];
        if (TWO) base[p.g.b1b - k.lo] = sc[0][1] + sc[1][1];
    }
}

// The streaming backward applies to what the one-pass kernel took (every parameter gradient, no action-input gradient) and,
// beyond it, to the policy step's pass through the critic: da wanted, parameter gradients none or the state part of a shared
// embedding's first layer.
static inline bool bwd_stream_applies(const BwdArgs& a, const SplitK& k) {
    const Mlp& net = a.net;
    if (!rpo_tune(RPO_TUNE_BWD_STREAM) || net.cat || net.E != 128 || net.H != 256 || net.hd > 1 || net.S + net.A > 11 || net.A > 8) return false;
    if (a.n < RPO_SPLITK_FROM) return false;
    if (!a.param_grads && !a.da) return false;                   // (nothing the streaming kernels produce is asked for: dh / dx0
                                                                 //  only -- the two-pass kernels write those)
    if (a.param_grads && k.Z < 2) return false;                  // (needs the slices' scratch)
    if (a.da && (net.n_out > 1 || net.A > 2)) return false;
    if (a.param_grads && !a.first_layer_state_only && a.da) return false;   // (no caller; keep the two-pass form)
    // 16-byte loads of the saved activations and of W1, 16-byte stores into the slices' copies of the gradient span
    uintptr_t bits = reinterpret_cast<uintptr_t>(a.x0) | reinterpret_cast<uintptr_t>(a.h1) | reinterpret_cast<uintptr_t>(net.W1) |
                     reinterpret_cast<uintptr_t>(net.W1b) | reinterpret_cast<uintptr_t>(net.W0);
    if (a.param_grads) bits |= reinterpret_cast<uintptr_t>(k.scratch) | (uintptr_t)(4 * (a.g.W0 - k.lo));
    if (bits & 15u) return false;
    return true;
}

static inline int launch_bwd_stream(const BwdArgs& args_in, const SplitK& k, hipStream_t stream) {
    BwdArgs a = args_in;
    const bool two = a.net.n_out > 1;
    if (a.td.q) {                                                // TD / Huber prologue -> dq_out, read as dout by both kernels
        hipLaunchKernelGGL(bwd_stream_td_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a.td, a.n);
        RPO_LAUNCH_CHECK();
        a.dout = a.td.dq_out;
    }
    const int cus = rpo_cu_count();
    const bool grads = a.param_grads != 0;
    if (grads && splitk_zero(k, stream) != 0) return RPO_ERR_ARG;
    const int Z = grads ? k.Z : cus;                             // rows only: nothing is reduced, every CU takes a share
    const dim3 grid(Z), block(kBwdStreamWaves * 64);
    const bool want_da = a.da != nullptr;
    const int gmode = !grads ? 0 : (a.first_layer_state_only ? 2 : 1);
#define RPO_ROWS(G, T, D) hipLaunchKernelGGL((bwd_stream_rows_kernel<G, T, D>), grid, block, 0, stream, a, k)
    if (gmode == 1 && !two && !want_da) RPO_ROWS(1, false, false);
    else if (gmode == 1 && two && !want_da) RPO_ROWS(1, true, false);
    else if (gmode == 0 && !two && want_da) RPO_ROWS(0, false, true);
    else if (gmode == 2 && !two && want_da) RPO_ROWS(2, false, true);
    else if (gmode == 2 && !two && !want_da) RPO_ROWS(2, false, false);
    else if (gmode == 2 && two && !want_da) RPO_ROWS(2, true, false);
    else return RPO_ERR_ARG;                                     // (unreachable: bwd_stream_applies admits exactly the six forms above)
#undef RPO_ROWS
    RPO_LAUNCH_CHECK();
    if (gmode == 1) {
        const dim3 wblock(kBwdWeightsWaves * 64);
        if (two) hipLaunchKernelGGL((bwd_stream_weights_kernel<true>), grid, wblock, 0, stream, a, k);
        else hipLaunchKernelGGL((bwd_stream_weights_kernel<false>), grid, wblock, 0, stream, a, k);
        RPO_LAUNCH_CHECK();
    }
    if (grads) {
        long long blocks = (k.span + RPO_BLOCK / 4 - 1) / (RPO_BLOCK / 4);
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL((splitk_reduce_kernel<0>), dim3((int)blocks), dim3(RPO_BLOCK), 0, stream, k, a.gradmax);
        RPO_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace rpo_mlp_dev
