// Backward pass of the MLP tile (see mlp.hip for the operand-layout notes): device bodies shared by the stand-alone
// kernels (mlp.hip) and the fused actor-update pipeline (fused.hip).
#pragma once
#include <stdlib.h>
#include "mlp_tile.h"
#include "heads_dev.h"

namespace rpo_mlp_dev {

// ------------------------------------------------------------------------------------------------- backward, rows
// Per row tile: dh = (dout W1) * 1[h1 > 0]  -> global (for the weights pass) and LDS; dW1 / db1 / db0 partial sums
// (one atomic per value per workgroup); dx0 = (dh W0) * 1[x0 > 0] -> global; optionally da = dx0_a Wa.
// Optional TD / Huber prologue of the rows pass (critic update): dout = dLoss/dQ is computed here from the critic's and
// the target critics' outputs instead of being read, written to dq_out for the weights pass, and the row tile's share of
// the loss goes to loss_partial[blockIdx.x].  q == NULL: no prologue.
struct TdArgs {
    const float* q;            // [n] Q(s, a) of this network
    const float* qn1;          // [n] Q_targ(s', a')
    const float* qn2;          // [n] second target critic or NULL
    const float* logp;         // [n] log pi(a'|s') or NULL
    const float* reward; int reward_stride;
    const float* done; int done_stride;
    float alpha, gamma;
    float* dq_out;             // [n]
    float* loss_partial;       // [ceil(n / 16)]
};

struct BwdArgs {
    Mlp net;
    MlpGrad g;
    int n;
    const float* s; int s_stride;
    const float* a; int a_stride;
    const float* x0;       // [n, Ein] saved by forward
    const float* h1;       // [n, H]
    const float* dout;     // [n, n_out]
    float* dh;             // [n, H]   scratch
    float* dx0;            // [n, Ein] scratch
    float* da;             // [n, A] or NULL: gradient w.r.t. the action input
    int param_grads;       // 0: only dx0 / da are needed (critic inside the actor loss, non-shared embedding)
    int first_layer_state_only;   // 1: of the parameter gradients only dWs / dbs are accumulated (shared embedding)
    float* gradmax;        // NULL, or where the weights pass leaves max |gradient element written| (clip_grad_norm_(inf))
    TdArgs td;
};

template <int EIN, int H>
__device__ __forceinline__ void mlp_bwd_rows_body(const BwdArgs& p) {
    constexpr int LDH = H + 4;
    __shared__ __attribute__((aligned(16))) float dh_s[kRows * LDH];
    __shared__ __attribute__((aligned(16))) float red[kRows * EIN];              // dx0 of the tile (waves add into it)
    __shared__ float dout_s[kRows * kWideOut];
    const Mlp& net = p.net;
    const int row0 = blockIdx.x * kRows;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool wide_head = net.hd > 1;
    const int outs = wide_head ? net.n_out * net.hd : net.n_out;
    if (p.td.q) {                                                  // TD target + Huber of the tile's rows (n_out == 1)
        if (tid < 64) {
            float dq = 0.0f, hub = 0.0f;
            const int i = row0 + tid;
            if (tid < kRows && i < p.n) {
                const TdArgs& t = p.td;
                const float qn = rpo_head_dev::td_next_value(t.qn1[i], t.qn2 ? t.qn2[i] : 0.0f, t.qn2 != nullptr,
                                                             t.logp ? t.logp[i] : 0.0f, t.logp != nullptr, t.alpha);
                const float y = rpo_head_dev::td_target(t.reward[(size_t)i * t.reward_stride],
                                                        t.done[(size_t)i * t.done_stride], t.gamma, qn);
                dq = rpo_head_dev::td_huber_row(t.q[i], y, 1.0f / (float)p.n, &hub);
                t.dq_out[i] = dq;
            }
            if (tid < kRows) { dout_s[tid * 2] = dq; dout_s[tid * 2 + 1] = 0.0f; }
            const float sum = rpo_row16_sum_desc_lane0(hub);     // (== rpo_wave_sum: only lanes 0..15 hold terms)
            if (tid == 0) p.td.loss_partial[blockIdx.x] = sum;
        }
    } else if (!wide_head) {
        if (tid < kRows * 2) {
            const int r = tid >> 1, o = tid & 1;
            dout_s[tid] = (o < net.n_out && row0 + r < p.n) ? p.dout[(size_t)(row0 + r) * net.n_out + o] : 0.0f;
        }
    } else {
        for (int idx = tid; idx < kRows * outs; idx += kThreads) {
            const int r = idx / outs, o = idx - r * outs;
            dout_s[r * kWideOut + o] = (row0 + r < p.n) ? p.dout[(size_t)(row0 + r) * outs + o] : 0.0f;
        }
    }
    __syncthreads();
    // ---- dh (thread = hidden column j); the batch reductions dW1 / db0 / db1 happen in the weights pass, in a fixed
    //      order, so that the whole backward is bitwise reproducible (no floating-point atomics anywhere)
    for (int j = tid; j < H; j += kThreads) {
        if (!wide_head) {
            const float w1a = net.W1[j], w1b = net.n_out > 1 ? net.W1b[j] : 0.0f;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const bool live = row0 + r < p.n;
                const float h = live ? p.h1[(size_t)(row0 + r) * H + j] : 0.0f;
                const float d = (h > 0.0f) ? fmaf(dout_s[r * 2 + 1], w1b, dout_s[r * 2] * w1a) : 0.0f;
                dh_s[r * LDH + j] = d;
                if (live) p.dh[(size_t)(row0 + r) * H + j] = d;
            }
        } else {
            float d[kRows];
#pragma unroll
            for (int r = 0; r < kRows; ++r) d[r] = 0.0f;
            for (int o = 0; o < outs; ++o) {                       // fixed order over the outputs
                const float wv = (o < net.hd ? net.W1 : net.W1b)[(size_t)(o < net.hd ? o : o - net.hd) * H + j];
#pragma unroll
                for (int r = 0; r < kRows; ++r) d[r] = fmaf(dout_s[r * kWideOut + o], wv, d[r]);
            }
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const bool live = row0 + r < p.n;
                const float h = live ? p.h1[(size_t)(row0 + r) * H + j] : 0.0f;
                const float v = (h > 0.0f) ? d[r] : 0.0f;
                dh_s[r * LDH + j] = v;
                if (live) p.dh[(size_t)(row0 + r) * H + j] = v;
            }
        }
    }
    __syncthreads();

    // ---- dx0 partials (MFMA): wave w sums over hidden j in [w*H/4, (w+1)*H/4), all EIN columns
    constexpr int NV = EIN / 64;                                   // float4 loads per k-step; 4 interleaved tiles each
    f32x4 acc[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[v][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int li = lane & 15, lg = lane >> 4;
    const int jw = wave * (H / 4);
#pragma unroll 8
    for (int ks = 0; ks < H / 16; ++ks) {
        const int j = jw + ks * 4 + lg;
        const float av = dh_s[li * LDH + j];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 b4 = *reinterpret_cast<const float4*>(&net.W0[(size_t)j * EIN + v * 64 + li * 4]);
            acc[v][0] = mfma4(av, b4.x, acc[v][0]);
            acc[v][1] = mfma4(av, b4.y, acc[v][1]);
            acc[v][2] = mfma4(av, b4.z, acc[v][2]);
            acc[v][3] = mfma4(av, b4.w, acc[v][3]);
        }
    }
    // acc[v][c][i] = partial dx0[row = 4*lg + i][e = 64v + 4*li + c]
    for (int w = 0; w < 4; ++w) {                                  // waves add their partials in a fixed order
        if (wave == w) {
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4* dst = reinterpret_cast<float4*>(&red[(lg * 4 + i) * EIN + v * 64 + li * 4]);
                    float4 o = make_float4(acc[v][0][i], acc[v][1][i], acc[v][2][i], acc[v][3][i]);
                    if (w > 0) { const float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
                    *dst = o;
                }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < kRows * EIN; idx += kThreads) {
        const int r = idx / EIN, e = idx - r * EIN;
        float v = red[idx];
        const bool live = row0 + r < p.n;
        const float x = live ? p.x0[(size_t)(row0 + r) * EIN + e] : 0.0f;
        v = (x > 0.0f) ? v : 0.0f;
        red[idx] = v;                                              // masked dx0 of this tile (for da below)
        if (live) p.dx0[(size_t)(row0 + r) * EIN + e] = v;
    }
    if (p.da) {
        __syncthreads();
        const int eoff = net.cat ? net.E : 0;                      // action embedding columns inside x0
        for (int idx = tid; idx < kRows * net.A; idx += kThreads) {
            const int r = idx / net.A, i = idx - r * net.A;
            if (row0 + r < p.n) {
                float s = 0.0f;
                for (int e = 0; e < net.E; ++e) s = fmaf(red[r * EIN + eoff + e], net.Wa[e * net.A + i], s);
                p.da[(size_t)(row0 + r) * net.A + i] = s;
            }
        }
    }
}

// First-layer gradients from dx0: batch reductions with one owner per output and a fixed summation order.  A workgroup
// owns 16 consecutive outputs and splits the batch 16 ways (256 threads = 16 outputs x 16 slices), so a thread's
// serial chain of dependent loads is n / 16 rows (two groups of 8 at batch 256) -- with 64 outputs x 4 slices it was eight
// groups, ~3 us of exposed latency on a launch that is otherwise the boundary floor.  The 16 slice sums are added in
// order.  Output list idx = q * E + e (e fastest: contiguous dx0 reads); q < wS: state weights / bias, q >= wS: action
// weights / bias.  Columns of x0: [0, E) <- state (+ action when added); [E, 2E) <- action (cat).
// Large first layers (EVOPF: 256 x 101 = 25 856 outputs) keep 64 outputs x 4 slices: there the launch is bound by the
// number of workgroups, not by one workgroup's load chain (928 instead of 232 workgroups cost +10 us, measured).
constexpr int kFlWideFrom = 4096;
__host__ __device__ constexpr int mlp_fl_out(int outputs) { return outputs > kFlWideFrom ? 64 : 16; }
__host__ __device__ constexpr int mlp_fl_blocks(int outputs) { return (outputs + mlp_fl_out(outputs) - 1) / mlp_fl_out(outputs); }

template <int EIN, int OUT>
__device__ __forceinline__ float mlp_bwd_first_layer_impl(const BwdArgs& p, int fl_block) {
    constexpr int SL = kThreads / OUT, CH = 16;                    // batch slices; rows of loads in flight per slice
    __shared__ float fl_partial[SL][OUT];
    const Mlp& net = p.net;
    const int tid = threadIdx.x;
    const int o = tid % OUT, part = tid / OUT;
    const int b_lo = (int)(((long long)p.n * part) / SL), b_hi = (int)(((long long)p.n * (part + 1)) / SL);
    float gmax = 0.0f;
    const int wS = net.S + 1, wA = (net.A > 0 && !p.first_layer_state_only) ? net.A + 1 : 0;   // +1: the bias
    const int idx = fl_block * OUT + o;
    const bool valid = idx < net.E * (wS + wA);
    float acc = 0.0f, cur = 0.0f;
    float* dst = nullptr;
    int e = 0, i = 0, width = 0;
    bool is_a = false;
    if (valid) {
        e = idx % net.E;
        const int q = idx / net.E;
        is_a = q >= wS;
        i = is_a ? q - wS : q;
        width = is_a ? net.A : net.S;
        // (what the gradient holds so far is requested now, not where it is added at the end of the chain)
        dst = (i < width) ? (is_a ? &p.g.Wa[e * net.A + i] : &p.g.Ws[e * net.S + i]) : (is_a ? &p.g.ba[e] : &p.g.bs[e]);
        if (part == 0) cur = *dst;
        const int col = (is_a && net.cat) ? net.E + e : e;
        const float* in = is_a ? p.a : p.s;
        const int stride = is_a ? p.a_stride : p.s_stride;
        const bool is_w = i < width;
        int bb = b_lo;
        for (; bb + CH <= b_hi; bb += CH) {
            float d[CH], x[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                d[u] = p.dx0[(size_t)(bb + u) * EIN + col];
                x[u] = is_w ? in[(size_t)(bb + u) * stride + i] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) acc = fmaf(d[u], x[u], acc);
        }
        for (; bb < b_hi; ++bb) acc = fmaf(p.dx0[(size_t)bb * EIN + col], is_w ? in[(size_t)bb * stride + i] : 1.0f, acc);
    }
    fl_partial[part][o] = acc;
    __syncthreads();
    if (part == 0 && valid) {
        float tot = fl_partial[0][o];
#pragma unroll
        for (int sl = 1; sl < SL; ++sl) tot += fl_partial[sl][o];
        const float nv = cur + tot;
        *dst = nv;
        gmax = fabsf(nv);
    }
    return gmax;
}

// The same gradients of a WIDE first layer (EVOPF: 256 x (57 + 1 | 43 + 1)) on the matrix cores (round 4).  The scalar role above
// is a chain of n / 4 rows of dependent loads per thread in 408 workgroups -- 11 of the 12 us of the critic's weights pass at
// batch 256, the longest of its roles.  Here dWs | dbs (and dWa | dba) are what they are: dx0^T [E x n] times [s | 1] [n x (S + 1)],
// 16 embedding rows x 64 input columns per workgroup like a dW0 tile (the four waves split the batch, their partial tiles meet
// in LDS and are added in a fixed order; one owner per output), E / 16 workgroups per input half; the remaining workgroups of
// the role's grid have nothing to do.  Inputs up to 63 wide; another (fixed) summation order than the scalar role.
template <int EIN>
__device__ __forceinline__ float mlp_bwd_first_layer_mfma(const BwdArgs& p, int fl_block) {
    const Mlp& net = p.net;
    const int halves = (net.A > 0 && !p.first_layer_state_only) ? 2 : 1, eb = net.E / 16;
    if (fl_block >= eb * halves) return 0.0f;
    __shared__ __attribute__((aligned(16))) float fl_tile[4][16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int half = fl_block / eb, et = fl_block - half * eb;
    const bool is_a = half == 1;
    const int width = is_a ? net.A : net.S;
    const float* in = is_a ? p.a : p.s;
    const int stride = is_a ? p.a_stride : p.s_stride;
    const int col = ((is_a && net.cat) ? net.E : 0) + et * 16 + li;           // this lane's column of dx0 (A operand, m = li)
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nk = (p.n + 3) / 4, ks_lo = (nk * wave) / 4, ks_hi = (nk * (wave + 1)) / 4, last = p.n - 1;
    int ks = ks_lo;
    for (; ks + 4 <= ks_hi; ks += 4) {                                          // 4 k-steps of loads in flight before their MFMAs
        float av[4], bv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = (ks + u) * 4 + lg, bc = b < last ? b : last;
            av[u] = p.dx0[(size_t)bc * EIN + col];
            if (b > last) av[u] = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int i = c * 16 + li;
                bv[u][c] = i < width ? in[(size_t)bc * stride + i] : (i == width ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = mfma4(av[u], bv[u][c], acc[c]);
    }
    for (; ks < ks_hi; ++ks) {
        const int b = ks * 4 + lg, bc = b < last ? b : last;
        float av = p.dx0[(size_t)bc * EIN + col];
        if (b > last) av = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = c * 16 + li;
            acc[c] = mfma4(av, i < width ? in[(size_t)bc * stride + i] : (i == width ? 1.0f : 0.0f), acc[c]);
        }
    }
    // acc[c][q] = partial d[e = 16 et + 4 lg + q][i = 16 c + li]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) fl_tile[wave][(lg * 4 + q) * 64 + c * 16 + li] = acc[c][q];
    __syncthreads();
    float gmax = 0.0f;
    const int r = tid >> 4, e = et * 16 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = (tid & 15) * 4 + q;
        if (i > width) continue;
        const int t = r * 64 + i;
        const float tot = ((fl_tile[0][t] + fl_tile[1][t]) + fl_tile[2][t]) + fl_tile[3][t];
        float* dst = i < width ? (is_a ? &p.g.Wa[e * net.A + i] : &p.g.Ws[e * net.S + i]) : (is_a ? &p.g.ba[e] : &p.g.bs[e]);
        const float nv = *dst + tot;
        *dst = nv;
        gmax = fmaxf(gmax, fabsf(nv));
    }
    return gmax;
}

template <int EIN>
__device__ __forceinline__ float mlp_bwd_first_layer(const BwdArgs& p, int fl_block) {
    const Mlp& net = p.net;
    const int outputs = net.E * (net.S + 1 + ((net.A > 0 && !p.first_layer_state_only) ? net.A + 1 : 0));
    if (outputs > kFlWideFrom) {
        if (net.S < 64 && net.A < 64 && (net.E & 15) == 0) return mlp_bwd_first_layer_mfma<EIN>(p, fl_block);
        return mlp_bwd_first_layer_impl<EIN, 64>(p, fl_block);
    }
    return mlp_bwd_first_layer_impl<EIN, 16>(p, fl_block);
}

// ------------------------------------------------------------------------------------------------ backward, weights
// Workgroups [0, H/16 * EIN/256): dW0 tiles (wave = 16 hidden rows x 64 input columns, K = batch); one more
// workgroup accumulates the first-layer gradients dWs / dbs / dWa / dba from dx0, and the last one db0 / dW1 / db1.
// Every output element has exactly one owner and a fixed summation order: the backward pass is bitwise reproducible.
template <int EIN, int H>
__device__ __forceinline__ float mlp_bwd_weights_body(const BwdArgs& p, int vblock = -1) {
    float gmax = 0.0f;                                          // largest |gradient element| this thread wrote
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = vblock < 0 ? (int)blockIdx.x : vblock;        // (the split-K launch numbers its blocks differently)
    constexpr int GEMM_BLOCKS = (H / 16) * (EIN / 64);           // one 16 x 64 tile of dW0 per workgroup
    if (bx < GEMM_BLOCKS) {
        if (!p.param_grads || p.first_layer_state_only) return gmax;
        // the 4 waves split the batch (K) and combine through LDS in a fixed order
        __shared__ __attribute__((aligned(16))) float tile[4][16 * 64];
        const int jt = bx / (EIN / 64), et = bx - jt * (EIN / 64);
        const int li = lane & 15, lg = lane >> 4;
        const int j = jt * 16 + li, e0 = et * 64 + li * 4;
        f32x4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int nk = (p.n + 3) / 4;                              // k-steps of 4 samples
        const int ks_lo = (nk * wave) / 4, ks_hi = (nk * (wave + 1)) / 4;
        const int last = p.n - 1;
        int ks = ks_lo;
        for (; ks + 4 <= ks_hi; ks += 4) {                         // 4 k-steps of loads in flight before their MFMAs
            float av[4];
            float4 bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = (ks + u) * 4 + lg;
                const int bc = b < last ? b : last;
                av[u] = p.dh[(size_t)bc * H + j];
                bv[u] = *reinterpret_cast<const float4*>(&p.x0[(size_t)bc * EIN + e0]);
                if (b > last) av[u] = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = mfma4(av[u], fmaxf(bv[u].x, 0.0f), acc[0]);
                acc[1] = mfma4(av[u], fmaxf(bv[u].y, 0.0f), acc[1]);
                acc[2] = mfma4(av[u], fmaxf(bv[u].z, 0.0f), acc[2]);
                acc[3] = mfma4(av[u], fmaxf(bv[u].w, 0.0f), acc[3]);
            }
        }
        for (; ks < ks_hi; ++ks) {
            const int b = ks * 4 + lg;
            const int bc = b < last ? b : last;
            float av = p.dh[(size_t)bc * H + j];
            const float4 b4 = *reinterpret_cast<const float4*>(&p.x0[(size_t)bc * EIN + e0]);
            if (b > last) av = 0.0f;
            acc[0] = mfma4(av, fmaxf(b4.x, 0.0f), acc[0]);
            acc[1] = mfma4(av, fmaxf(b4.y, 0.0f), acc[1]);
            acc[2] = mfma4(av, fmaxf(b4.z, 0.0f), acc[2]);
            acc[3] = mfma4(av, fmaxf(b4.w, 0.0f), acc[3]);
        }
        // acc[c][i] = partial dW0[j = 16 jt + 4 lg + i][e = 64 et + 4 li + c]
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4*>(&tile[wave][(lg * 4 + i) * 64 + li * 4]) =
                make_float4(acc[0][i], acc[1][i], acc[2][i], acc[3][i]);
        __syncthreads();
        {
            const int r = tid >> 4, c4 = (tid & 15) * 4;           // 256 threads x float4 = the 16 x 64 tile
            const float4 t0 = *reinterpret_cast<const float4*>(&tile[0][r * 64 + c4]);
            const float4 t1 = *reinterpret_cast<const float4*>(&tile[1][r * 64 + c4]);
            const float4 t2 = *reinterpret_cast<const float4*>(&tile[2][r * 64 + c4]);
            const float4 t3 = *reinterpret_cast<const float4*>(&tile[3][r * 64 + c4]);
            float4* dst = reinterpret_cast<float4*>(&p.g.W0[(size_t)(jt * 16 + r) * EIN + et * 64 + c4]);
            float4 cur = *dst;
            cur.x += ((t0.x + t1.x) + t2.x) + t3.x;
            cur.y += ((t0.y + t1.y) + t2.y) + t3.y;
            cur.z += ((t0.z + t1.z) + t2.z) + t3.z;
            cur.w += ((t0.w + t1.w) + t2.w) + t3.w;
            *dst = cur;
            gmax = fmaxf(fmaxf(fabsf(cur.x), fabsf(cur.y)), fmaxf(fabsf(cur.z), fabsf(cur.w)));
        }
        return gmax;
    }
    if (!p.param_grads) return gmax;
    // ---- batch reductions with one owner per output: 64 outputs per workgroup, the batch split over the 4 waves and
    //      combined through LDS in a fixed order (bitwise reproducible)
    __shared__ float partial[4][3][64];
    const int o = tid & 63, part = tid >> 6;
    const int b_lo = (int)(((long long)p.n * part) / 4), b_hi = (int)(((long long)p.n * (part + 1)) / 4);
    const int rb = bx - GEMM_BLOCKS;
    constexpr int HV_BLOCKS = H / 64;
    const int hv_blocks = net.hd > 1 ? H / 16 : HV_BLOCKS;
    if (rb < hv_blocks && net.hd > 1) {
        // multi-output head: db0[j], dW1_k[o][j] = sum_b dout[b][k*hd + o] relu(h1[b][j]), db1_k[o] = sum_b dout[b][k*hd + o].
        // 16 hidden columns per workgroup, the batch split 16 ways and combined through LDS in a fixed order (one owner per
        // output): H/16 workgroups x 16 batch slices keep the exposed load latency to n/16 rows per thread.
        if (p.first_layer_state_only) return gmax;
        __shared__ float wide[16][kWideOut + 1][16];
        static_assert(256 * kWideOut <= 16 * (kWideOut + 1) * 16, "the staged head gradients live in `wide`");
        float* dout_s = &wide[0][0][0];                            // the head gradients of (up to) 256 rows, staged once per pass
        const int outs = net.n_out * net.hd;
        const int jj = tid & 15, slice = tid >> 4;
        const int j = rb * 16 + jj;
        float gb0 = 0.0f, gw[kWideOut];
#pragma unroll
        for (int q = 0; q < kWideOut; ++q) gw[q] = 0.0f;
        // Passes of 256 rows: thread (jj, slice) owns rows [16 slice, 16 slice + 16) of the pass -- for n <= 256 the same rows in
        // the same order as a 16-way split of the batch.  Its 16 rows of dh / h1 are requested together and the pass's head
        // gradients are staged in LDS by the whole workgroup (they were 16 x outs dependent global loads per thread: this role
        // was 21 of the launch's 22 us at batch 256 with 14 outputs).
        for (int base = 0; base < p.n; base += 256) {
            const int rows = p.n - base < 256 ? p.n - base : 256;
            __syncthreads();
            for (int idx = tid; idx < rows * outs; idx += kThreads) dout_s[idx] = p.dout[(size_t)base * outs + idx];
            const int lo = (int)(((long long)rows * slice) / 16), hi = (int)(((long long)rows * (slice + 1)) / 16);
            float d[16], h[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = lo + u < hi ? lo + u : (hi > lo ? hi - 1 : 0);
                d[u] = p.dh[(size_t)(base + r) * H + j];
                h[u] = fmaxf(p.h1[(size_t)(base + r) * H + j], 0.0f);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (lo + u < hi) {
                    gb0 += d[u];
                    const float* drow = dout_s + (lo + u) * outs;
#pragma unroll
                    for (int q = 0; q < kWideOut; ++q)
                        if (q < outs) gw[q] = fmaf(drow[q], h[u], gw[q]);
                }
            }
        }
        __syncthreads();                                           // (`wide` held the staged head gradients until here)
        wide[slice][kWideOut][jj] = gb0;
#pragma unroll
        for (int q = 0; q < kWideOut; ++q) wide[slice][q][jj] = gw[q];
        __syncthreads();
        for (int idx = tid; idx < (outs + 1) * 16; idx += kThreads) {
            const int q = idx >> 4, c = idx & 15;                  // q == outs: the db0 column sums
            const int src = q == outs ? kWideOut : q;
            float tot = 0.0f;
            for (int sl = 0; sl < 16; ++sl) tot += wide[sl][src][c];
            float* dst = q == outs ? &p.g.b0[rb * 16 + c]
                                   : (q < net.hd ? &p.g.W1[(size_t)q * H + rb * 16 + c] : &p.g.W1b[(size_t)(q - net.hd) * H + rb * 16 + c]);
            const float nv = *dst + tot;
            *dst = nv;
            gmax = fmaxf(gmax, fabsf(nv));
        }
        if (rb == 0) {
            // db1: 8 batch slices x 32 outputs, combined in a fixed order
            __syncthreads();
            float* red = &wide[0][0][0];
            const int q = tid & 31, sl8 = tid >> 5;
            const int l8 = (int)(((long long)p.n * sl8) / 8), h8 = (int)(((long long)p.n * (sl8 + 1)) / 8);
            float s0 = 0.0f;
            if (q < outs) {
                int b2 = l8;
                for (; b2 + 32 <= h8; b2 += 32) {                  // 32 rows of loads in flight (a slice of batch 256), summed in order
                    float dv[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) dv[u] = p.dout[(size_t)(b2 + u) * outs + q];
#pragma unroll
                    for (int u = 0; u < 32; ++u) s0 += dv[u];
                }
                for (; b2 < h8; ++b2) s0 += p.dout[(size_t)b2 * outs + q];
            }
            red[sl8 * 32 + q] = s0;
            __syncthreads();
            if (tid < outs) {
                float tot = 0.0f;
                for (int sl = 0; sl < 8; ++sl) tot += red[sl * 32 + tid];
                float* dst = tid < net.hd ? &p.g.b1[tid] : &p.g.b1b[tid - net.hd];
                const float nv = *dst + tot;
                *dst = nv;
                gmax = fmaxf(gmax, fabsf(nv));
            }
        }
        return gmax;
    }
    if (rb < HV_BLOCKS) {
        // hidden-layer vectors: db0[j] = sum_b dh[b][j]; dW1_k[j] = sum_b dout[b][k] relu(h1[b][j]); db1_k = sum_b dout[b][k]
        if (p.first_layer_state_only) return gmax;
        const int j = rb * 64 + o;
        float gb0 = 0.0f, gw1a = 0.0f, gw1b = 0.0f;
        int bb = b_lo;
        for (; bb + 8 <= b_hi; bb += 8) {
            float d[8], h[8], oa[8], ob[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d[u] = p.dh[(size_t)(bb + u) * H + j];
                h[u] = p.h1[(size_t)(bb + u) * H + j];
                oa[u] = p.dout[(size_t)(bb + u) * net.n_out];
                ob[u] = net.n_out > 1 ? p.dout[(size_t)(bb + u) * net.n_out + 1] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gb0 += d[u];
                const float hr = fmaxf(h[u], 0.0f);
                gw1a = fmaf(oa[u], hr, gw1a);
                gw1b = fmaf(ob[u], hr, gw1b);
            }
        }
        for (; bb < b_hi; ++bb) {
            gb0 += p.dh[(size_t)bb * H + j];
            const float hr = fmaxf(p.h1[(size_t)bb * H + j], 0.0f);
            gw1a = fmaf(p.dout[(size_t)bb * net.n_out], hr, gw1a);
            if (net.n_out > 1) gw1b = fmaf(p.dout[(size_t)bb * net.n_out + 1], hr, gw1b);
        }
        partial[part][0][o] = gb0; partial[part][1][o] = gw1a; partial[part][2][o] = gw1b;
        __syncthreads();
        if (part == 0) {
            const float nb0 = p.g.b0[j] + (((partial[0][0][o] + partial[1][0][o]) + partial[2][0][o]) + partial[3][0][o]);
            const float nw1 = p.g.W1[j] + (((partial[0][1][o] + partial[1][1][o]) + partial[2][1][o]) + partial[3][1][o]);
            p.g.b0[j] = nb0;
            p.g.W1[j] = nw1;
            gmax = fmaxf(fabsf(nb0), fabsf(nw1));
            if (net.n_out > 1) {
                const float nw1b = p.g.W1b[j] + (((partial[0][2][o] + partial[1][2][o]) + partial[2][2][o]) + partial[3][2][o]);
                p.g.W1b[j] = nw1b;
                gmax = fmaxf(gmax, fabsf(nw1b));
            }
        }
        if (rb == 0) {
            // db1_k = sum_b dout[b][k]: strided per-thread partials, fixed-pattern wave reduction, 4 wave partials added
            // in order (a single-thread loop over the batch costs ~18 us of serialised load latency)
            __syncthreads();                                   // `partial` is reused below
            float s0 = 0.0f, s1 = 0.0f;
            for (int b2 = tid; b2 < p.n; b2 += kThreads) {
                s0 += p.dout[(size_t)b2 * net.n_out];
                if (net.n_out > 1) s1 += p.dout[(size_t)b2 * net.n_out + 1];
            }
            { float ss[2] = {s0, s1}; rpo_wave_reduce_many(ss, 0u); s0 = ss[0]; s1 = ss[1]; }
            if (o == 0) { partial[part][0][0] = s0; partial[part][1][0] = s1; }
            __syncthreads();
            if (tid == 0) {
                const float nb1 = p.g.b1[0] + (((partial[0][0][0] + partial[1][0][0]) + partial[2][0][0]) + partial[3][0][0]);
                p.g.b1[0] = nb1;
                gmax = fmaxf(gmax, fabsf(nb1));
                if (net.n_out > 1) {
                    const float nb1b = p.g.b1b[0] + (((partial[0][1][0] + partial[1][1][0]) + partial[2][1][0]) + partial[3][1][0]);
                    p.g.b1b[0] = nb1b;
                    gmax = fmaxf(gmax, fabsf(nb1b));
                }
            }
        }
        return gmax;
    }
    return mlp_bwd_first_layer<EIN>(p, rb - hv_blocks);
}

// max over the workgroup of the gradient magnitudes its threads wrote -> one atomic max into one of the gradmax slots
// (order independent: exact)
__device__ __forceinline__ void gradmax_flush(float* gradmax, float v) {
    __shared__ float red[kThreads / 64];
    if (gradmax == nullptr) return;
    v = rpo_wave_max_nonneg(v);                                  // (gradient magnitudes)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, red[w]);
        // slot (workgroup index) % 16, 64 bytes apart: same-line atomics of a wide launch queue up behind each other
        const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
        if (m > 0.0f) rpo_atomic_max_nonneg(gradmax + (b % RPO_GRADMAX_SLOTS) * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS), m);
    }
}
// ------------------------------------------------------------------------------------ weights pass for LARGE batches
// mlp_bwd_weights_body gives every output element ONE owner that walks the whole batch: ~50 workgroups, fine at batch 256
// (latency-bound anyway) and 36 ms at 2^20 rows.  Split-K form: grid (the same blocks, Z batch slices); slice z runs the same
// body on rows [n z / Z, n (z + 1) / Z) (multiples of 4) and accumulates into ITS copy of the network's gradient span inside a
// zeroed scratch buffer; splitk_reduce adds the Z copies in order onto the gradient and takes the inf-norm of what it wrote.
// One owner per element in both launches, fixed orders: bitwise reproducible (not bitwise the Z = 1 result: other association).
struct SplitK {
    float* scratch;        // [Z][span]
    float* lo;             // lowest address of the network's gradient tensors (the span starts here)
    long long span;        // floats from `lo` to the end of the highest tensor
    int Z;
    long long stride;      // floats between two slices' copies: span rounded up to 4 (every slice starts 16-byte aligned)
};

// dW0 tile of 64 hidden rows x 64 input columns over the slice's rows: wave w owns hidden rows [16 w, 16 w + 16) for the WHOLE
// slice (the batch is already split by the grid), so the four waves read the same x0 values (one L1 line each) and a slice's x0
// is fetched by 4 workgroups instead of 16.  One owner per element, k-steps in order.
template <int EIN, int H>
__device__ __forceinline__ void splitk_dw0_tile64(const BwdArgs& p, int blk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int jt = blk / (EIN / 64), et = blk - jt * (EIN / 64);
    const int j = jt * 64 + wave * 16 + li, e0 = et * 64 + li * 4;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nk = (p.n + 3) / 4, last = p.n - 1;
    int ks = 0;
    for (; ks + 16 <= nk; ks += 16) {                            // 16 k-steps of loads in flight before their MFMAs
        float av[16];
        float4 bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = (ks + u) * 4 + lg, bc = b < last ? b : last;
            av[u] = p.dh[(size_t)bc * H + j];
            bv[u] = *reinterpret_cast<const float4*>(&p.x0[(size_t)bc * EIN + e0]);
            if (b > last) av[u] = 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] = mfma4(av[u], fmaxf(bv[u].x, 0.0f), acc[0]);
            acc[1] = mfma4(av[u], fmaxf(bv[u].y, 0.0f), acc[1]);
            acc[2] = mfma4(av[u], fmaxf(bv[u].z, 0.0f), acc[2]);
            acc[3] = mfma4(av[u], fmaxf(bv[u].w, 0.0f), acc[3]);
        }
    }
    for (; ks < nk; ++ks) {
        const int b = ks * 4 + lg, bc = b < last ? b : last;
        float av = p.dh[(size_t)bc * H + j];
        const float4 b4 = *reinterpret_cast<const float4*>(&p.x0[(size_t)bc * EIN + e0]);
        if (b > last) av = 0.0f;
        acc[0] = mfma4(av, fmaxf(b4.x, 0.0f), acc[0]);
        acc[1] = mfma4(av, fmaxf(b4.y, 0.0f), acc[1]);
        acc[2] = mfma4(av, fmaxf(b4.z, 0.0f), acc[2]);
        acc[3] = mfma4(av, fmaxf(b4.w, 0.0f), acc[3]);
    }
    // acc[c][i] = dW0[j = 64 jt + 16 wave + 4 lg + i][e = 64 et + 4 li + c] of this slice
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4* dst = reinterpret_cast<float4*>(&p.g.W0[(size_t)(jt * 64 + wave * 16 + lg * 4 + i) * EIN + e0]);
        float4 cur = *dst;
        cur.x += acc[0][i]; cur.y += acc[1][i]; cur.z += acc[2][i]; cur.w += acc[3][i];
        *dst = cur;
    }
}

// First-layer gradients of the slice: a workgroup owns 16 columns e of x0 and EVERY input of them (state inputs + bias, action
// inputs + bias: up to 16 + 16 outputs per column), so the slice's dx0 is read once instead of once per input.  Thread = (column
// e, 16 row phases); the 16 phase sums are added in order.  Scalar-input networks only (S + 1, A + 1 <= 16).
template <int EIN>
__device__ __forceinline__ void splitk_first_layer_cols(const BwdArgs& p, int eb, bool action_half) {
    __shared__ float part[16][16][17];
    const Mlp& net = p.net;
    const int tid = threadIdx.x, ec = tid & 15, ph = tid >> 4;
    const int width = action_half ? net.A : net.S, e = eb * 16 + ec;
    const int col = (action_half && net.cat) ? net.E + e : e;
    const float* in = action_half ? p.a : p.s;
    const int stride = action_half ? p.a_stride : p.s_stride;
    float acc[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) acc[i] = 0.0f;
    constexpr int U = 8;                                          // rows of loads in flight per thread
    int b = ph;
    for (; b + 16 * (U - 1) < p.n; b += 16 * U) {
        float d[U], x[U][16];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d[u] = p.dx0[(size_t)(b + 16 * u) * EIN + col];
#pragma unroll
            for (int i = 0; i < 16; ++i) x[u][i] = i < width ? in[(size_t)(b + 16 * u) * stride + i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {                             // (rows in order: the same chain as one row at a time)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i < width) acc[i] = fmaf(d[u], x[u][i], acc[i]);
            acc[16] += d[u];                                      // the bias
        }
    }
    for (; b < p.n; b += 16) {
        const float d = p.dx0[(size_t)b * EIN + col];
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < width) acc[i] = fmaf(d, in[(size_t)b * stride + i], acc[i]);
        acc[16] += d;
    }
#pragma unroll
    for (int i = 0; i < 17; ++i) part[ph][ec][i] = acc[i];
    __syncthreads();
    for (int idx = tid; idx < 16 * (width + 1); idx += kThreads) {
        const int c2 = idx / (width + 1), i = idx - c2 * (width + 1), src = i < width ? i : 16;
        float tot = part[0][c2][src];
#pragma unroll
        for (int q = 1; q < 16; ++q) tot += part[q][c2][src];
        const int e2 = eb * 16 + c2;
        float* dst = i < width ? (action_half ? &p.g.Wa[e2 * net.A + i] : &p.g.Ws[e2 * net.S + i])
                               : (action_half ? &p.g.ba[e2] : &p.g.bs[e2]);
        *dst += tot;
    }
}

template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_weights_splitk_kernel(BwdArgs p, SplitK k) {
    const int z = blockIdx.y;
    const long long r0 = (((long long)p.n * z) / k.Z) & ~3LL;
    const long long r1 = z + 1 == k.Z ? (long long)p.n : ((((long long)p.n * (z + 1)) / k.Z) & ~3LL);
    if (r1 <= r0) return;
    BwdArgs q = p;
    q.n = (int)(r1 - r0);
    q.s += r0 * p.s_stride;
    if (q.a) q.a += r0 * p.a_stride;
    q.x0 += r0 * EIN; q.h1 += r0 * H; q.dh += r0 * H; q.dx0 += r0 * EIN;
    q.dout += r0 * (p.net.hd > 1 ? p.net.n_out * p.net.hd : p.net.n_out);
    q.gradmax = nullptr;
    float* base = k.scratch + (long long)z * k.stride;
#define RPO_SK(F) q.g.F = p.g.F ? base + (p.g.F - k.lo) : nullptr
    RPO_SK(Ws); RPO_SK(bs); RPO_SK(Wa); RPO_SK(ba); RPO_SK(W0); RPO_SK(b0); RPO_SK(W1); RPO_SK(b1); RPO_SK(W1b); RPO_SK(b1b);
#undef RPO_SK
    // blocks: [0, G64) dW0 tiles of 64 x 64 | [G64, G64 + HV) hidden-layer vectors (the body's blocks) | first layer
    constexpr int G64 = (H / 64) * (EIN / 64), G16 = (H / 16) * (EIN / 64);
    const int hv = p.net.hd > 1 ? H / 16 : H / 64, bx = blockIdx.x;
    const bool narrow = p.net.S + 1 <= 16 && p.net.A + 1 <= 16 && p.net.hd <= 1;
    if (bx < G64) {
        if (p.param_grads && !p.first_layer_state_only) splitk_dw0_tile64<EIN, H>(q, bx);
        return;
    }
    if (bx < G64 + hv || !narrow) {                              // (wide first layers keep the body's output-owned blocks)
        mlp_bwd_weights_body<EIN, H>(q, bx - G64 + G16);
        return;
    }
    const int fb = bx - G64 - hv, eblocks = p.net.E / 16;
    if (fb < eblocks) splitk_first_layer_cols<EIN>(q, fb, false);
    else if (fb < 2 * eblocks && p.net.A > 0 && !p.first_layer_state_only) splitk_first_layer_cols<EIN>(q, fb - eblocks, true);
}

// (round 5: four threads per element, each adding a quarter of the slices in order, the four partial sums combined as
//  ((q0 + q1) + q2) + q3 -- one thread per element walked 256 dependent loads, 62 us of a 1.4 ms backward.  Still one owner per
//  element and fixed orders.)
// The slices' scratch is zeroed by a KERNEL of this library (RPO_SPLITK_ZERO: 1 default; 0: hipMemsetAsync, the form of rounds
// 3-5).  Round 6 found the large-batch update non-reproducible from run to run under hipGraph replay -- garbage of the size of a
// gradient in the padding floats behind the critic's head bias, NaN once in a few thousand updates -- never with eager launches
// and never when anything was inspected between replays (tools/probe/dbg_padding.py): see DESIGN.md 4.5.
#ifndef RPO_SPLITK_ZERO
#define RPO_SPLITK_ZERO 1
#endif
template <int DUMMY>                                           // (a template: one definition across the translation units that include this)
__global__ __launch_bounds__(RPO_BLOCK) void splitk_zero_kernel(float4* __restrict__ p, long long n4) {
    for (long long i = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * RPO_BLOCK)
        p[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
static inline int splitk_zero(const SplitK& k, hipStream_t stream) {
    const size_t floats = (size_t)k.Z * (size_t)k.stride;        // (stride is a multiple of 4 floats; the scratch is 16-byte aligned)
    if (RPO_SPLITK_ZERO && (reinterpret_cast<uintptr_t>(k.scratch) & 15u) == 0) {
        long long n4 = (long long)(floats / 4), blocks = (n4 + RPO_BLOCK - 1) / RPO_BLOCK;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_zero_kernel<0>, dim3((unsigned)blocks), dim3(RPO_BLOCK), 0, stream, reinterpret_cast<float4*>(k.scratch), n4);
        return hipGetLastError() == hipSuccess ? 0 : RPO_ERR_ARG;
    }
    return hipMemsetAsync(k.scratch, 0, floats * sizeof(float), stream) == hipSuccess ? 0 : RPO_ERR_ARG;
}

template <int DUMMY>
__global__ __launch_bounds__(RPO_BLOCK) void splitk_reduce_kernel(SplitK k, float* gradmax) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    __shared__ float part[4][RPO_BLOCK / 4];
    constexpr int EPB = RPO_BLOCK / 4;                             // elements per block and pass
    const int e_in = threadIdx.x & (EPB - 1), quarter = threadIdx.x / EPB;
    const int zq = (k.Z + 3) / 4, z_lo = quarter * zq, z_hi = z_lo + zq < k.Z ? z_lo + zq : k.Z;
    float gmax = 0.0f;
    for (long long i0 = (long long)blockIdx.x * EPB; i0 < k.span; i0 += (long long)gridDim.x * EPB) {
        const long long i = i0 + e_in;
        float s = 0.0f;
        if (i < k.span)
            for (int z = z_lo; z < z_hi; ++z) s += k.scratch[(long long)z * k.stride + i];    // slices in order
        part[quarter][e_in] = s;
        __syncthreads();
        if (quarter == 0 && i < k.span) {
            const float tot = ((part[0][e_in] + part[1][e_in]) + part[2][e_in]) + part[3][e_in];
            if (tot != 0.0f) {                                     // (positions inside the span that belong to other tensors stay 0)
                const float nv = k.lo[i] + tot;
                k.lo[i] = nv;
                gmax = fmaxf(gmax, fabsf(nv));
            }
        }
        __syncthreads();
    }
    if (gradmax == nullptr) return;
    gmax = rpo_wave_max_nonneg(gmax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < RPO_BLOCK / RPO_WAVE; ++w) m = fmaxf(m, red[w]);
        if (m > 0.0f) rpo_atomic_max_nonneg(gradmax + (blockIdx.x % RPO_GRADMAX_SLOTS) * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS), m);
    }
}

// ---- ONE pass over the saved activations for large batches of the 128 -> 256 scalar-head networks (critic / actor updates
// of the classic-control envs in the large-batch mode): rows pass + every parameter-gradient reduction in one persistent
// kernel.  The two-pass form writes dh and dx0 (1.5 GB at 2^20 rows) and reads h1 / x0 / dh / dx0 back two to four times over;
// here a workgroup walks ITS slice of the batch 16 rows at a time and
//   * keeps its B operands of dx0 = dh W0 in registers for the whole slice (128 floats per thread: the W0 slice of the wave's
//     32 output columns),
//   * accumulates dW0 += dh^T relu(x0) in 128 accumulator registers per thread (wave w owns hidden rows [64 w, 64 w + 64)),
//   * keeps db0 / dW1 (thread = hidden column) and the first-layer gradients (thread = embedding column) in registers,
// and writes them once, into its copy of the gradient span (the scratch of the split-K pass; splitk_reduce adds the copies in
// order).  h1 and x0 are read once, nothing else touches HBM: 1.5 GB instead of ~9 GB per 2^20 rows.  Next tile's h1 / x0 are
// requested while the current tile's 256 MFMAs per wave run.  One owner per element, tiles in order: bitwise reproducible.
// Two groups of four waves share a workgroup (two waves per SIMD, ~200 registers each -- one group holding both the dW0
// accumulators and the W0 slice needs > 512 registers per thread and spilled):
//   W waves  stage the tile's x0 in LDS (requested one tile ahead) and run dW0 += dh^T relu(x0) (128 MFMAs per wave per tile);
//   X waves  request h1 / head gradients / inputs one tile ahead, form dh (thread = hidden column; db0, dW1 along the way),
//            run dx0 = (dh W0) * 1[x0 > 0] (128 MFMAs per wave per tile) and the first-layer sums (thread = (input half, e)).
// The vector work of one group runs under the other's MFMAs.
constexpr int kOnepassThreads = 512;

template <int EIN, int H>
__global__ __launch_bounds__(kOnepassThreads) void mlp_bwd_onepass_kernel(BwdArgs p, SplitK k) {
    static_assert(EIN == 128 && H == 256, "one thread per hidden column, two per embedding column");
    constexpr int LDH = H + 4, LDX = EIN + 4, NT = 256;
    __shared__ __attribute__((aligned(16))) float dh_s[kRows * LDH];
    __shared__ __attribute__((aligned(16))) float x0_s[kRows * LDX];
    __shared__ __attribute__((aligned(16))) float dx_s[kRows * LDX];
    __shared__ __attribute__((aligned(16))) float in_sb[2][kRows * 8], in_ab[2][kRows * 8], dout_s[kRows * 2];   // (inputs by tile parity)
    const Mlp& net = p.net;
    const bool xgrp = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= NT;   // (wave-uniform: a scalar branch, the groups' registers do not interfere)
    const int tid = threadIdx.x & (NT - 1), lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int z = blockIdx.x, Z = gridDim.x;
    const long long tiles = ((long long)p.n + kRows - 1) / kRows;
    const long long t_lo = tiles * z / Z, t_hi = tiles * (z + 1) / Z;
    if (t_hi <= t_lo) return;
    const bool two = net.n_out > 1, has_a = net.A > 0;
    float* base = k.scratch + (long long)z * k.stride;
#define RPO_SK(F) (base + (p.g.F - k.lo))
    if (!xgrp) {
        // ================================================================ W waves: x0 staging + dW0
        f32x4 accw[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) accw[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        float xn[kRows * EIN / NT];
        // first-layer sums: thread (input half, column e) adds dx0[r][e] x its inputs, rows in order -- the PREVIOUS tile's, while
        // the X waves form dh of the current one (both are vector work: on the X waves alone this phase was exposed, 0.4 ms)
        float gfl[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) gfl[i] = 0.0f;
        const int e_own = tid & (EIN - 1);                           // threads [0, 128): state inputs of column e; [128, 256): action inputs
        const bool act_half = tid >= EIN;
        auto first_layer = [&](long long tile) {
            if (act_half && !has_a) return;
            const float4* in = reinterpret_cast<const float4*>(act_half ? in_ab[tile & 1] : in_sb[tile & 1]);
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
                const float d = dx_s[r * LDX + e_own];
                const float4 i0 = in[r * 2], i1 = in[r * 2 + 1];         // (the same address in every lane: broadcast reads)
                gfl[0] = fmaf(d, i0.x, gfl[0]); gfl[1] = fmaf(d, i0.y, gfl[1]); gfl[2] = fmaf(d, i0.z, gfl[2]); gfl[3] = fmaf(d, i0.w, gfl[3]);
                gfl[4] = fmaf(d, i1.x, gfl[4]); gfl[5] = fmaf(d, i1.y, gfl[5]); gfl[6] = fmaf(d, i1.z, gfl[6]); gfl[7] = fmaf(d, i1.w, gfl[7]);
                gfl[8] += d;
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        };
        float tq = 0.0f, tqn1 = 0.0f, tqn2 = 0.0f, tlp = 0.0f, trw = 0.0f, tdn = 0.0f, tdo = 0.0f, tin_s = 0.0f, tin_a = 0.0f;   // small operands
        auto request = [&](long long tile) {
            const long long m0 = tile * kRows;
            if (tid < kRows && p.td.q) {                             // the TD prologue's operands of row m0 + tid
                const TdArgs& t = p.td;
                const long long i = m0 + tid < p.n ? m0 + tid : p.n - 1;
                tq = t.q[i]; tqn1 = t.qn1[i]; tqn2 = t.qn2 ? t.qn2[i] : 0.0f; tlp = t.logp ? t.logp[i] : 0.0f;
                trw = t.reward[(size_t)i * t.reward_stride]; tdn = t.done[(size_t)i * t.done_stride];
            } else if (tid < kRows * 2 && !p.td.q) {
                const int r = tid >> 1, o = tid & 1;
                const long long m = m0 + r < p.n ? m0 + r : p.n - 1;
                tdo = o < net.n_out ? p.dout[(size_t)m * net.n_out + o] : 0.0f;
            } else if (tid >= 64 && tid < 64 + kRows * 8) {
                const int q = tid - 64, r = q >> 3, i = q & 7;
                const long long m = m0 + r < p.n ? m0 + r : p.n - 1;
                tin_s = i < net.S ? p.s[(size_t)m * p.s_stride + i] : 0.0f;
                tin_a = (has_a && i < net.A) ? p.a[(size_t)m * p.a_stride + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kRows * EIN / NT; ++u) {
                const int idx = tid + u * NT, r = idx / EIN, e = idx - r * EIN;
                const long long m = m0 + r < p.n ? m0 + r : p.n - 1;
                xn[u] = p.x0[(size_t)m * EIN + e];
            }
        };
        request(t_lo);
        for (long long tile = t_lo; tile < t_hi; ++tile) {
            const long long m0 = tile * kRows;
#pragma unroll
            for (int u = 0; u < kRows * EIN / NT; ++u) {
                const int idx = tid + u * NT, r = idx / EIN, e = idx - r * EIN;
                x0_s[r * LDX + e] = m0 + r < p.n ? xn[u] : 0.0f;
            }
            // ---- the tile's head gradients (given, or the TD target + Huber prologue of the critic update) and inputs
            if (tid < 64) {                                            // (operands requested one tile ahead: `request`)
                if (p.td.q) {
                    const TdArgs& t = p.td;
                    const long long i = m0 + tid;
                    float dq = 0.0f, hub = 0.0f;
                    if (tid < kRows && i < p.n) {
                        const float qn = rpo_head_dev::td_next_value(tqn1, tqn2, t.qn2 != nullptr, tlp, t.logp != nullptr, t.alpha);
                        const float y = rpo_head_dev::td_target(trw, tdn, t.gamma, qn);
                        dq = rpo_head_dev::td_huber_row(tq, y, 1.0f / (float)p.n, &hub);
                        t.dq_out[i] = dq;
                    }
                    if (tid < kRows) { dout_s[tid * 2] = dq; dout_s[tid * 2 + 1] = 0.0f; }
                    const float sum = rpo_row16_sum_desc_lane0(hub);
                    if (tid == 0) t.loss_partial[tile] = sum;
                } else if (tid < kRows * 2) {
                    dout_s[tid] = m0 + (tid >> 1) < p.n ? tdo : 0.0f;
                }
            } else if (tid < 64 + kRows * 8) {
                const int q = tid - 64, r = q >> 3;
                const bool live = m0 + r < p.n;
                in_sb[tile & 1][q] = live ? tin_s : 0.0f;
                in_ab[tile & 1][q] = live ? tin_a : 0.0f;
            }
        if (tile + 1 < t_hi) request(tile + 1);
            __syncthreads();                                         // B1: x0_s, dout_s, the inputs
            if (tile > t_lo) first_layer(tile - 1);                  // (dx_s of the previous tile: rewritten behind B2)
            __syncthreads();                                         // B2: dh_s
#pragma unroll
            for (int ks = 0; ks < kRows / 4; ++ks) {                 // dW0 += dh^T relu(x0): hidden rows [64 wave, 64 wave + 64)
                float av[4], bv[8];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) av[mt] = dh_s[(ks * 4 + lg) * LDH + wave * 64 + mt * 16 + li];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) bv[nt] = fmaxf(x0_s[(ks * 4 + lg) * LDX + nt * 16 + li], 0.0f);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) accw[mt][nt] = mfma4(av[mt], bv[nt], accw[mt][nt]);
            }
            __syncthreads();                                         // B3: dx_s written, x0_s / dh_s free
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    RPO_SK(W0)[(size_t)(wave * 64 + mt * 16 + lg * 4 + i) * EIN + nt * 16 + li] = accw[mt][nt][i];
        first_layer(t_hi - 1);                                       // (the last tile's dx_s is complete: B3)
        if (!act_half) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < net.S) RPO_SK(Ws)[e_own * net.S + i] = gfl[i];
            RPO_SK(bs)[e_own] = gfl[8];
        } else if (has_a) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < net.A) RPO_SK(Wa)[e_own * net.A + i] = gfl[i];
            RPO_SK(ba)[e_own] = gfl[8];
        }
        return;
    }
    // ==================================================================== X waves: dh, dx0, hidden vectors, first layer
    float w0r[2][H / 4];                                           // B operands of dx0 = dh W0: columns n = 32 wave + 16 t + li, k = 4 ks + lg
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < H / 4; ++ks) w0r[t][ks] = net.W0[(size_t)(ks * 4 + lg) * EIN + wave * 32 + t * 16 + li];
    const float w1a = net.W1[tid], w1b = two ? net.W1b[tid] : 0.0f;
    float gb0 = 0.0f, gw1a = 0.0f, gw1b = 0.0f, gb1a = 0.0f, gb1b = 0.0f;
    float hn[kRows];
    auto request = [&](long long tile) {
        const long long m0 = tile * kRows;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const long long m = m0 + r < p.n ? m0 + r : p.n - 1;
            hn[r] = p.h1[(size_t)m * H + tid];
        }
    };
    request(t_lo);
    for (long long tile = t_lo; tile < t_hi; ++tile) {
        const long long m0 = tile * kRows;
        __syncthreads();                                             // B1
        // ---- dh (thread = hidden column), db0 / dW1 along the way
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const bool live = m0 + r < p.n;
            const float h = live ? hn[r] : 0.0f, da_ = dout_s[r * 2], db_ = dout_s[r * 2 + 1];
            const float d = (h > 0.0f) ? fmaf(db_, w1b, da_ * w1a) : 0.0f;
            dh_s[r * LDH + tid] = d;
            gb0 += d;
            const float hr = fmaxf(h, 0.0f);
            gw1a = fmaf(da_, hr, gw1a);
            gw1b = fmaf(db_, hr, gw1b);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // (bounds the temporaries: the W0 slice must stay in registers)
        }
        if (tid == 0) {
#pragma unroll
            for (int r = 0; r < kRows; ++r) { gb1a += dout_s[r * 2]; gb1b += dout_s[r * 2 + 1]; }
        }
        if (tile + 1 < t_hi) request(tile + 1);                      // (lands under the MFMA phase and the first-layer sums)
        __syncthreads();                                             // B2
        // ---- dx0 = (dh W0) * 1[x0 > 0]: wave w owns columns [32 w, 32 w + 32)
        // (two chains per column tile -- even and odd k-steps, added at the end: with one accumulator per tile every MFMA waited
        //  for the one two before it, 0.85 ms of the launch)
        f32x4 ax[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
        f32x4 ay[2] = {f32x4{0.0f, 0.0f, 0.0f, 0.0f}, f32x4{0.0f, 0.0f, 0.0f, 0.0f}};
#pragma unroll                                                  // (fully: w0r must be indexed statically to stay in registers;
        for (int kc = 0; kc < H / 32; ++kc) {                    //  fenced in eights so the scheduler does not hoist all 64 LDS reads)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int ks = kc * 8 + kk;
                const float av = dh_s[li * LDH + ks * 4 + lg];
                if (ks & 1) { ay[0] = mfma4(av, w0r[0][ks], ay[0]); ay[1] = mfma4(av, w0r[1][ks], ay[1]); }
                else { ax[0] = mfma4(av, w0r[0][ks], ax[0]); ax[1] = mfma4(av, w0r[1][ks], ax[1]); }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = lg * 4 + i, n = wave * 32 + t * 16 + li;
                dx_s[r * LDX + n] = x0_s[r * LDX + n] > 0.0f ? ax[t][i] + ay[t][i] : 0.0f;
            }
        __syncthreads();                                             // B3
        // (the first-layer sums of this tile are the W waves' work behind the next B1)
    }
    // ---- the slice's sums into its copy of the gradient span
    RPO_SK(b0)[tid] = gb0;
    RPO_SK(W1)[tid] = gw1a;
    if (two) RPO_SK(W1b)[tid] = gw1b;
    if (tid == 0) { RPO_SK(b1)[0] = gb1a; if (two) RPO_SK(b1b)[0] = gb1b; }
#undef RPO_SK
}

// Host side: the split-K plan for `args` (Z = 0: not applicable -- small batch, no scratch, or scratch too small)
static inline SplitK splitk_plan(const BwdArgs& a, float* scratch, long long scratch_floats) {
    SplitK k{nullptr, nullptr, 0, 0, 0};
    if (!scratch || a.n < RPO_SPLITK_FROM || !a.param_grads) return k;
    const Mlp& net = a.net;
    const long long ein = net.cat ? 2 * net.E : net.E, heads = net.hd > 1 ? net.hd : 1;
    const float* ptr[10] = {a.g.Ws, a.g.bs, a.g.Wa, a.g.ba, a.g.W0, a.g.b0, a.g.W1, a.g.b1, a.g.W1b, a.g.b1b};
    const long long len[10] = {(long long)net.E * net.S, net.E, (long long)net.E * net.A, net.E, (long long)net.H * ein, net.H,
                               heads * net.H, heads, heads * net.H, heads};
    const float *lo = nullptr, *hi = nullptr;
    for (int i = 0; i < 10; ++i) {
        if (!ptr[i] || len[i] == 0) continue;
        if (!lo || ptr[i] < lo) lo = ptr[i];
        if (!hi || ptr[i] + len[i] > hi) hi = ptr[i] + len[i];
    }
    if (!lo) return k;
    const long long span = hi - lo, stride = (span + 3) & ~3ll;
    long long Z = a.n / 512;                                     // (>= 32 row tiles per slice: two per wave of a streaming workgroup.
    if (Z > 256) Z = 256;                                        //  Until round 6 n / 4096: 4 workgroups at 16 384 rows, 16 at 65 536 --
                                                                 //  the persistent kernels reached one per CU only at 2^20)
    if (Z * stride > scratch_floats) Z = scratch_floats / stride;
    if (Z < 2) return k;
    k.scratch = scratch; k.lo = const_cast<float*>(lo); k.span = span; k.Z = (int)Z; k.stride = stride;
    return k;
}

// The one-pass form applies to the scalar-head 128 -> 256 networks with up to 8 state / action inputs when every parameter
// gradient is wanted and the action-input gradient is not (critic update, actor update).  Applicability is a question of its
// own (onepass_applies) so that "does not apply" can never be mistaken for a failed launch (both used to be -1: a failed
// memset was swallowed, and in the pair form the second network's "not applicable" surfaced as an error after the first
// had already been launched -- ADVICE r03).
template <int EIN, int H>
static inline bool onepass_applies(const BwdArgs& args, const SplitK& k) {
    const Mlp& net = args.net;
    if (EIN != 128 || H != 256 || net.cat || net.hd > 1 || net.S > 8 || net.A > 8 || !args.param_grads || args.first_layer_state_only ||
        args.da || k.Z < 2)
        return false;
    return rpo_tune(RPO_TUNE_BWD_ONEPASS) != 0;
}

// Launches the one-pass backward (caller checked onepass_applies): 0 or an RPO_ERR_* / hipError code.
template <int EIN, int H>
static inline int launch_onepass(const BwdArgs& args, const SplitK& k, hipStream_t stream) {
    if (splitk_zero(k, stream) != 0) return RPO_ERR_ARG;
    hipLaunchKernelGGL((mlp_bwd_onepass_kernel<128, 256>), dim3(k.Z), dim3(kOnepassThreads), 0, stream, args, k);
    RPO_LAUNCH_CHECK();
    long long blocks = (k.span + RPO_BLOCK / 4 - 1) / (RPO_BLOCK / 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((unsigned)blocks), dim3(RPO_BLOCK), 0, stream, k, args.gradmax);
    RPO_LAUNCH_CHECK();
    return 0;
}

template <int EIN, int H>
static inline int launch_weights_splitk(const BwdArgs& args, const SplitK& k, int grid_w, hipStream_t stream) {
    if (splitk_zero(k, stream) != 0) return RPO_ERR_ARG;
    // (grid_w counts the plain pass's blocks: its (H / 16) (EIN / 64) dW0 tiles become (H / 64) (EIN / 64); a narrow first
    // layer becomes E / 16 column blocks per input half)
    const Mlp& net = args.net;
    const int g16 = (H / 16) * (EIN / 64), g64 = (H / 64) * (EIN / 64), hv = net.hd > 1 ? H / 16 : H / 64;
    const bool narrow = net.S + 1 <= 16 && net.A + 1 <= 16 && net.hd <= 1;
    grid_w = narrow ? g64 + hv + (net.E / 16) * (net.A > 0 ? 2 : 1) : grid_w - g16 + g64;
    hipLaunchKernelGGL((mlp_bwd_weights_splitk_kernel<EIN, H>), dim3(grid_w, k.Z), dim3(kThreads), 0, stream, args, k);
    RPO_LAUNCH_CHECK();
    long long blocks = (k.span + RPO_BLOCK / 4 - 1) / (RPO_BLOCK / 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(splitk_reduce_kernel<0>, dim3((unsigned)blocks), dim3(RPO_BLOCK), 0, stream, k, args.gradmax);
    RPO_LAUNCH_CHECK();
    return 0;
}

// The streaming backward (mlp_bwd_stream.h) lives in a translation unit of its own, mlp_bwd_stream.hip, compiled with
// -fno-slp-vectorize (packed f32 vector instructions beside f32 MFMAs: 1 179 -> 1 144 us at 2^20 rows, round 6); mlp.hip
// reaches it through these two (hidden: not symbols of the shared library's ABI).
__attribute__((visibility("hidden"))) bool bwd_stream_applies_x(const BwdArgs& a, const SplitK& k);
__attribute__((visibility("hidden"))) int launch_bwd_stream_x(const BwdArgs& a, const SplitK& k, hipStream_t stream);

}  // namespace rpo_mlp_dev
