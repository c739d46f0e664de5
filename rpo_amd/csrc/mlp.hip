// Small-MLP forward / backward for the RPO actor and critics on MI355X, exact f32 on the matrix cores.
//
// Every network of the reference's agents (rpo/algo/model/{embedding,policy,value}.py with hidden_layer = 1) is
//     x0 = s Ws^T + bs (+ a Wa^T + ba | concatenated)      first layer, K = S (+A) <= ~100      -> VALU
//     h1 = relu(x0) W0^T + b0                              Ein -> H (128 -> 256): all the flops  -> MFMA
//     out = relu(h1) W1^T + b1                             H -> n_out (1 or 2)                   -> wave reductions
// rocBLAS runs these M = 256 .. 4096 GEMMs in 8-36 us each (one or two macro-tiles busy); here one workgroup owns 16
// rows through all three layers (activations stay in LDS / accumulators) and the backward pass is two launches.
//
// v_mfma_f32_16x16x4_f32 is an exact k-ordered fmaf chain.  Operand maps (wave64): A[i = l&15][k = l>>4],
// B[k = l>>4][j = l&15], C/D col = l&15, row = 4*(l>>4) + reg.  Weights keep torch's nn.Linear layout [out][in]:
//   forward   h1[r][j] = sum_k x1[r][k] W0[j][k]   K-contiguous: each lane loads a float4 along k (A from LDS the same
//             way) and feeds 4 consecutive MFMAs -- an MFMA only needs A and B to agree on which k sits in slot l>>4;
//   backward  dx0[r][e] = sum_j dh[r][j] W0[j][e]  N-contiguous: a float4 along e serves 4 interleaved output tiles
//             (tile c, column n <-> e = 4n + c) that share the A operand;
//   weights   dW0[j][e] = sum_b dh[b][j] x1[b][e]  both operands contiguous in their M / N index.
#include "heads_dev.h"
#include "mlp_tile.h"

namespace {

using namespace rpo_mlp_dev;

// ------------------------------------------------------------------------------------------------- forward
// out_mode: 0 identity, 1 tanh-box on output 0 (scale * tanh(o) + base; BoxConstraint, model/utils.py:40-51)
struct FwdArgs {
    Mlp net;
    int n;
    const float* s; int s_stride;
    const float* a; int a_stride;
    float* out;            // [n, n_out]
    float* x0_save;        // [n, Ein] pre-activation of the first layer (NULL: not saved)
    float* h1_save;        // [n, H]   pre-activation of the hidden layer (NULL: not saved)
    int out_mode; float scale, base;
};

template <int EIN, int H, int RT, int INS, int INA>
__global__ __launch_bounds__(kFwdThreads) void mlp_forward_kernel(FwdArgs p) {
    typedef TileLds<EIN, RT, INS, INA> Lds;
    __shared__ Lds lds;
    constexpr int ROWS = kRows * RT;
    const Mlp& net = p.net;
    const int row0 = blockIdx.x * ROWS;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < ROWS * net.S; idx += kFwdThreads) {
        const int r = idx / net.S, i = idx - r * net.S;
        lds.in_s[r * INS + i] = (row0 + r < p.n) ? p.s[(size_t)(row0 + r) * p.s_stride + i] : 0.0f;
    }
    for (int idx = tid; idx < ROWS * net.A; idx += kFwdThreads) {
        const int r = idx / net.A, i = idx - r * net.A;
        lds.in_a[r * INA + i] = (row0 + r < p.n) ? p.a[(size_t)(row0 + r) * p.a_stride + i] : 0.0f;
    }
    mlp_tile_forward<EIN, H, RT, Lds>(net, lds, row0, p.n, p.x0_save, p.h1_save, p.out_mode, p.scale, p.base);
    if (tid < ROWS * net.n_out) {
        const int r = tid / net.n_out, o = tid - r * net.n_out;
        if (row0 + r < p.n) p.out[(size_t)(row0 + r) * net.n_out + o] = lds.out[r * 2 + o];
    }
}

// Multi-output networks (hd > 1, e.g. the 14 basic actions of EVOPF-v0): same tile, MFMA head, raw outputs
// [n, n_out * hd] (head-major); the state-dependent tanh box of such actors is applied by the env's own kernels.
template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void mlp_forward_wide_kernel(FwdArgs p) {
    typedef TileLdsWide<EIN, H> Lds;
    __shared__ Lds lds;
    const Mlp& net = p.net;
    const int row0 = blockIdx.x * kRows;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < kRows * net.S; idx += kFwdThreads) {
        const int r = idx / net.S, i = idx - r * net.S;
        lds.in_s[r * kInS + i] = (row0 + r < p.n) ? p.s[(size_t)(row0 + r) * p.s_stride + i] : 0.0f;
    }
    for (int idx = tid; idx < kRows * net.A; idx += kFwdThreads) {
        const int r = idx / net.A, i = idx - r * net.A;
        lds.in_a[r * kInA + i] = (row0 + r < p.n) ? p.a[(size_t)(row0 + r) * p.a_stride + i] : 0.0f;
    }
    TileWeights<EIN, H, 1> w;
    tile_load_weights<EIN, H, 1>(net, w);
    tile_compute<EIN, H, 1, Lds, true>(net, w, lds, row0, p.n, p.x0_save, p.h1_save, 0, 1.0f, 0.0f);
    const int outs = net.n_out * net.hd;
    if (tid < kRows * outs) {
        const int r = tid / outs, o = tid - r * outs;
        if (row0 + r < p.n) p.out[(size_t)(row0 + r) * outs + o] = lds.outw[r * kWideOut + o];
    }
}

// ------------------------------------------------------------------------------------------------- backward, rows
// Per row tile: dh = (dout W1) * 1[h1 > 0]  -> global (for the weights pass) and LDS; dW1 / db1 / db0 partial sums
// (one atomic per value per workgroup); dx0 = (dh W0) * 1[x0 > 0] -> global; optionally da = dx0_a Wa.
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
    if (!wide_head) {
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

// ------------------------------------------------------------------------------------------------ backward, weights
// Workgroups [0, H/16 * EIN/256): dW0 tiles (wave = 16 hidden rows x 64 input columns, K = batch); one more
// workgroup accumulates the first-layer gradients dWs / dbs / dWa / dba from dx0, and the last one db0 / dW1 / db1.
// Every output element has exactly one owner and a fixed summation order: the backward pass is bitwise reproducible.
template <int EIN, int H>
__device__ __forceinline__ float mlp_bwd_weights_body(const BwdArgs& p) {
    float gmax = 0.0f;                                          // largest |gradient element| this thread wrote
    const Mlp& net = p.net;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int GEMM_BLOCKS = (H / 16) * (EIN / 64);           // one 16 x 64 tile of dW0 per workgroup
    if ((int)blockIdx.x < GEMM_BLOCKS) {
        if (!p.param_grads || p.first_layer_state_only) return gmax;
        // the 4 waves split the batch (K) and combine through LDS in a fixed order
        __shared__ __attribute__((aligned(16))) float tile[4][16 * 64];
        const int jt = blockIdx.x / (EIN / 64), et = blockIdx.x - jt * (EIN / 64);
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
    const int rb = (int)blockIdx.x - GEMM_BLOCKS;
    constexpr int HV_BLOCKS = H / 64;
    const int hv_blocks = net.hd > 1 ? H / 16 : HV_BLOCKS;
    if (rb < hv_blocks && net.hd > 1) {
        // multi-output head: db0[j], dW1_k[o][j] = sum_b dout[b][k*hd + o] relu(h1[b][j]), db1_k[o] = sum_b dout[b][k*hd + o].
        // 16 hidden columns per workgroup, the batch split 16 ways and combined through LDS in a fixed order (one owner per
        // output): H/16 workgroups x 16 batch slices keep the exposed load latency to n/16 rows per thread.
        if (p.first_layer_state_only) return gmax;
        __shared__ float wide[16][kWideOut + 1][16];
        const int outs = net.n_out * net.hd;
        const int jj = tid & 15, slice = tid >> 4;
        const int j = rb * 16 + jj;
        const int lo = (int)(((long long)p.n * slice) / 16), hi = (int)(((long long)p.n * (slice + 1)) / 16);
        float gb0 = 0.0f, gw[kWideOut];
#pragma unroll
        for (int q = 0; q < kWideOut; ++q) gw[q] = 0.0f;
        int bb = lo;
        for (; bb + 4 <= hi; bb += 4) {
            float d[4], h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                d[u] = p.dh[(size_t)(bb + u) * H + j];
                h[u] = fmaxf(p.h1[(size_t)(bb + u) * H + j], 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gb0 += d[u];
                const float* drow = p.dout + (size_t)(bb + u) * outs;
#pragma unroll
                for (int q = 0; q < kWideOut; ++q)
                    if (q < outs) gw[q] = fmaf(drow[q], h[u], gw[q]);
            }
        }
        for (; bb < hi; ++bb) {
            gb0 += p.dh[(size_t)bb * H + j];
            const float hr = fmaxf(p.h1[(size_t)bb * H + j], 0.0f);
#pragma unroll
            for (int q = 0; q < kWideOut; ++q)
                if (q < outs) gw[q] = fmaf(p.dout[(size_t)bb * outs + q], hr, gw[q]);
        }
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
            if (q < outs)
                for (int b2 = l8; b2 < h8; ++b2) s0 += p.dout[(size_t)b2 * outs + q];
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
            s0 = rpo_wave_sum(s0);
            s1 = rpo_wave_sum(s1);
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
    // first layer: output list idx = q * E + e (e fastest: coalesced dx0 reads); q < wS: state weights / bias,
    // q >= wS: action weights / bias.  Columns of x0: [0, E) <- state (+ action when added); [E, 2E) <- action (cat).
    const int wS = net.S + 1, wA = (net.A > 0 && !p.first_layer_state_only) ? net.A + 1 : 0;   // +1: the bias
    const int idx = (rb - hv_blocks) * 64 + o;
    const bool valid = idx < net.E * (wS + wA);
    float acc = 0.0f;
    int e = 0, i = 0, width = 0;
    bool is_a = false;
    if (valid) {
        e = idx % net.E;
        const int q = idx / net.E;
        is_a = q >= wS;
        i = is_a ? q - wS : q;
        width = is_a ? net.A : net.S;
        const int col = (is_a && net.cat) ? net.E + e : e;
        const float* in = is_a ? p.a : p.s;
        const int stride = is_a ? p.a_stride : p.s_stride;
        const bool is_w = i < width;
        int bb = b_lo;
        for (; bb + 8 <= b_hi; bb += 8) {
            float d[8], x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                d[u] = p.dx0[(size_t)(bb + u) * EIN + col];
                x[u] = is_w ? in[(size_t)(bb + u) * stride + i] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(d[u], x[u], acc);
        }
        for (; bb < b_hi; ++bb) acc = fmaf(p.dx0[(size_t)bb * EIN + col], is_w ? in[(size_t)bb * stride + i] : 1.0f, acc);
    }
    partial[part][0][o] = acc;
    __syncthreads();
    if (part == 0 && valid) {
        const float tot = ((partial[0][0][o] + partial[1][0][o]) + partial[2][0][o]) + partial[3][0][o];
        float* dst = (i < width) ? (is_a ? &p.g.Wa[e * net.A + i] : &p.g.Ws[e * net.S + i]) : (is_a ? &p.g.ba[e] : &p.g.bs[e]);
        const float nv = *dst + tot;
        *dst = nv;
        gmax = fabsf(nv);
    }
    return gmax;
}

template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_rows_kernel(BwdArgs p) { mlp_bwd_rows_body<EIN, H>(p); }
// max over the workgroup of the gradient magnitudes its threads wrote -> one atomic max (order independent: exact)
__device__ __forceinline__ void gradmax_flush(float* gradmax, float v) {
    __shared__ float red[kThreads / 64];
    if (gradmax == nullptr) return;
    v = rpo_wave_max(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, red[w]);
        if (m > 0.0f) rpo_atomic_max_nonneg(gradmax, m);
    }
}
template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_weights_kernel(BwdArgs p) {
    gradmax_flush(p.gradmax, mlp_bwd_weights_body<EIN, H>(p));
}

// Two networks of the same shape in one launch (blockIdx.y picks the network): SAC's twin critics.  Both backward
// passes are latency-bound on a handful of workgroups, so the pair costs what one costs.
struct BwdArgs2 {
    BwdArgs net[2];
};
template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_rows_kernel2(BwdArgs2 p) { mlp_bwd_rows_body<EIN, H>(p.net[blockIdx.y]); }
template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_weights_kernel2(BwdArgs2 p) {
    gradmax_flush(p.net[blockIdx.y].gradmax, mlp_bwd_weights_body<EIN, H>(p.net[blockIdx.y]));
}

// ------------------------------------------------------------------------------------------------- policy heads
// DDPG head backward (model/policy.py:30-31 + agent/ddpg_pa.py:108-110): ap = clip(scale*tanh(o)+base + eps_t*noise);
// dout = dap * 1[lo <= ap_noisy <= hi] * scale * (1 - tanh(o)^2), tanh(o) recovered from the stored deterministic ap.
__global__ __launch_bounds__(RPO_BLOCK) void tanh_box_bwd_kernel(int n, const float* __restrict__ dap,
                                                                 const float* __restrict__ ap_det,
                                                                 const float* __restrict__ noise, float eps_start,
                                                                 float eps_end, float eps_decay,
                                                                 const long long* __restrict__ ctrl, float lo, float hi,
                                                                 float scale, float base, float* __restrict__ dout) {
    const float t = ctrl ? (float)ctrl[RPO_CTRL_T] : 0.0f;
    const float eps_t = fmaxf(eps_end, eps_start - eps_decay * t);
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float a = ap_det[i];
        const float x = noise ? a + eps_t * noise[i] : a;
        const float y = (a - base) / scale;
        const bool pass = !noise || (x >= lo && x <= hi);
        dout[i] = pass ? dap[i] * scale * (1.0f - y * y) : 0.0f;
    }
}

// Squashed-Gaussian head (GaussianSharedPolicy.forward, model/policy.py:53-66, + the clip of PDSAC_PA.take_action,
// agent/sac_pa.py:111): raw = (mean, log-std head output); eps = the standard-normal draw of rsample.
using rpo_head_dev::kLogSigMin;
using rpo_head_dev::kLogSigMax;

__global__ __launch_bounds__(RPO_BLOCK) void gauss_head_kernel(int n, const float* __restrict__ raw,
                                                               const float* __restrict__ eps, float scale, float base,
                                                               float lo, float hi, int deterministic,
                                                               float* __restrict__ ap_out, float* __restrict__ logp_out) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 r = reinterpret_cast<const float2*>(raw)[i];
        float lp;
        ap_out[i] = rpo_head_dev::gauss_head_row(r.x, r.y, eps[i], scale, base, lo, hi, deterministic, &lp);
        if (logp_out) logp_out[i] = lp;
    }
}

// d(loss)/d(raw) given d/d(ap) per row and a uniform d/d(log_prob) (SAC actor loss: alpha / B, rpo_sac.py:331).
__global__ __launch_bounds__(RPO_BLOCK) void gauss_head_bwd_kernel(int n, const float* __restrict__ raw,
                                                                   const float* __restrict__ eps,
                                                                   const float* __restrict__ dap, float dlogp,
                                                                   float scale, float base, float lo, float hi,
                                                                   float* __restrict__ draw) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 r = reinterpret_cast<const float2*>(raw)[i];
        const float lsr = r.y - 3.0f;
        const float ls = fminf(fmaxf(lsr, kLogSigMin), kLogSigMax);
        const float sd = expf(ls), e = eps[i];
        const float y = tanhf(r.x + e * sd);
        const float omy = 1.0f - y * y;
        const float a = scale * y + base;
        const float g_ap = (a >= lo && a <= hi) ? dap[i] * scale * omy : 0.0f;
        const float g_lp = dlogp * (2.0f * scale * y * omy) / (scale * omy + 1e-6f);
        const float gx = g_ap + g_lp;
        const float dls = gx * e * sd - dlogp;
        reinterpret_cast<float2*>(draw)[i] = make_float2(gx, (lsr >= kLogSigMin && lsr <= kLogSigMax) ? dls : 0.0f);
    }
}

int check_net(const Mlp& m) {
    if (m.S <= 0 || m.S > 64 || m.A < 0 || m.A > 48 || m.n_out < 1 || m.n_out > 2) return RPO_ERR_ARG;
    if (m.hd < 0 || m.hd > 16) return RPO_ERR_ARG;
    if (!m.Ws || !m.bs || !m.W0 || !m.b0 || !m.W1 || !m.b1 || (m.A > 0 && (!m.Wa || !m.ba))) return RPO_ERR_NULL;
    if (m.n_out > 1 && (!m.W1b || !m.b1b)) return RPO_ERR_NULL;
    if (m.cat && m.A == 0) return RPO_ERR_ARG;
    return 0;
}

// supported (Ein, H): (128,256) cart / pendulum scripts; (256,256); (512,256) concatenating critic with E = 256
#define RPO_MLP_DISPATCH(EIN_, H_, KERNEL, GRID, ARGS)                                                    \
    if (ein == EIN_ && net.H == H_) {                                                                       \
        hipLaunchKernelGGL((KERNEL<EIN_, H_>), dim3(GRID), dim3(kFwdThreads), 0, (hipStream_t)stream, ARGS); \
        RPO_LAUNCH_CHECK();                                                                                 \
        return 0;                                                                                           \
    }

}  // namespace

extern "C" {

int rpo_mlp_supported(int E, int H, int cat) {
    const int ein = cat ? 2 * E : E;
    return (H == 256 && (ein == 128 || ein == 256 || ein == 512)) ? 1 : 0;   // instantiated (Ein, H) pairs
}

int rpo_mlp_forward(const rpo_mlp* net_host, int n, const float* s, int s_stride, const float* a, int a_stride,
                    float* out, float* x0_save, float* h1_save, int out_mode, float scale, float base, void* stream) {
    if (!net_host) return RPO_ERR_NULL;
    Mlp net{net_host->Ws, net_host->bs, net_host->Wa, net_host->ba, net_host->W0, net_host->b0, net_host->W1,
            net_host->b1, net_host->W1b, net_host->b1b, net_host->S, net_host->A, net_host->E, net_host->H,
            net_host->n_out, net_host->cat, net_host->head_dim};
    if (int e = check_net(net)) return e;
    if (n <= 0 || s_stride < net.S || (net.A > 0 && a_stride < net.A)) return RPO_ERR_ARG;
    if (!s || !out || (net.A > 0 && !a)) return RPO_ERR_NULL;
    const int ein = net.cat ? 2 * net.E : net.E;
    FwdArgs args{net, n, s, s_stride, a, a_stride, out, x0_save, h1_save, out_mode, scale, base};
    if (net.hd > 1) {
        if (out_mode != 0) return RPO_ERR_ARG;
#define RPO_MLP_FWD_WIDE(EIN_, H_)                                                                                    \
        if (ein == EIN_ && net.H == H_) {                                                                             \
            hipLaunchKernelGGL((mlp_forward_wide_kernel<EIN_, H_>), dim3((n + kRows - 1) / kRows), dim3(kFwdThreads), \
                               0, (hipStream_t)stream, args);                                                         \
            RPO_LAUNCH_CHECK();                                                                                       \
            return 0;                                                                                                 \
        }
        RPO_MLP_FWD_WIDE(128, 256)
        RPO_MLP_FWD_WIDE(256, 256)
        return RPO_ERR_ARG;
    }
    // 64 rows per workgroup once that still fills the chip (and the inputs fit the narrow LDS tiles); else 16
    const bool wide = n >= 64 * 192 && net.S <= 8 && net.A <= 8 && ein == 128;   // (LDS: 64 x 132 floats of x1)
#define RPO_MLP_FWD(EIN_, H_)                                                                                          \
    if (ein == EIN_ && net.H == H_) {                                                                                  \
        if (wide) {                                                                                                    \
            hipLaunchKernelGGL((mlp_forward_kernel<128, H_, 4, 8, 8>), dim3((n + 63) / 64),                            \
                               dim3(kFwdThreads), 0, (hipStream_t)stream, args);                                       \
        } else {                                                                                                       \
            hipLaunchKernelGGL((mlp_forward_kernel<EIN_, H_, 1, kInS, kInA>), dim3((n + kRows - 1) / kRows),           \
                               dim3(kFwdThreads), 0, (hipStream_t)stream, args);                                       \
        }                                                                                                              \
        RPO_LAUNCH_CHECK();                                                                                            \
        return 0;                                                                                                      \
    }
    RPO_MLP_FWD(128, 256)
    RPO_MLP_FWD(256, 256)
    RPO_MLP_FWD(512, 256)
    return RPO_ERR_ARG;
}

static int make_bwd_args(BwdArgs& args, const rpo_mlp* net_host, const rpo_mlp_grad* grad_host, int n, const float* s,
                         int s_stride, const float* a, int a_stride, const float* x0, const float* h1, const float* dout,
                         float* dh, float* dx0, float* da, int param_grads, int first_layer_state_only, float* gradmax) {
    if (!net_host) return RPO_ERR_NULL;
    Mlp net{net_host->Ws, net_host->bs, net_host->Wa, net_host->ba, net_host->W0, net_host->b0, net_host->W1,
            net_host->b1, net_host->W1b, net_host->b1b, net_host->S, net_host->A, net_host->E, net_host->H,
            net_host->n_out, net_host->cat, net_host->head_dim};
    if (int e = check_net(net)) return e;
    if (n <= 0 || s_stride < net.S || (net.A > 0 && a_stride < net.A)) return RPO_ERR_ARG;
    if (!s || !x0 || !h1 || !dout || !dh || !dx0 || (net.A > 0 && !a)) return RPO_ERR_NULL;
    MlpGrad g{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (param_grads) {
        if (!grad_host) return RPO_ERR_NULL;
        g = MlpGrad{grad_host->Ws, grad_host->bs, grad_host->Wa, grad_host->ba, grad_host->W0, grad_host->b0,
                    grad_host->W1, grad_host->b1, grad_host->W1b, grad_host->b1b};
        if (!g.Ws || !g.bs) return RPO_ERR_NULL;
        if (!first_layer_state_only && (!g.W0 || !g.b0 || !g.W1 || !g.b1 || (net.A > 0 && (!g.Wa || !g.ba))))
            return RPO_ERR_NULL;
        if (!first_layer_state_only && net.n_out > 1 && (!g.W1b || !g.b1b)) return RPO_ERR_NULL;
    }
    args = BwdArgs{net, g, n, s, s_stride, a, a_stride, x0, h1, dout, dh, dx0, da, param_grads, first_layer_state_only,
                   gradmax};
    return 0;
}

static int bwd_weights_grid(const Mlp& net, int first_layer_state_only) {
    const int ein = net.cat ? 2 * net.E : net.E;
    const int fl_outputs = net.E * (net.S + 1 + ((net.A > 0 && !first_layer_state_only) ? net.A + 1 : 0));
    return (net.H / 16) * (ein / 64) + (net.hd > 1 ? net.H / 16 : net.H / 64) + (fl_outputs + 63) / 64;
}

int rpo_mlp_backward(const rpo_mlp* net_host, const rpo_mlp_grad* grad_host, int n, const float* s, int s_stride,
                     const float* a, int a_stride, const float* x0, const float* h1, const float* dout, float* dh,
                     float* dx0, float* da, int param_grads, int first_layer_state_only, float* gradmax, void* stream) {
    BwdArgs args;
    if (int e = make_bwd_args(args, net_host, grad_host, n, s, s_stride, a, a_stride, x0, h1, dout, dh, dx0, da,
                              param_grads, first_layer_state_only, gradmax))
        return e;
    const Mlp& net = args.net;
    const int ein = net.cat ? 2 * net.E : net.E;
    const int grid_rows = (n + kRows - 1) / kRows;
    const int grid_w = bwd_weights_grid(net, first_layer_state_only);
#define RPO_MLP_BWD(EIN_, H_)                                                                                       \
    if (ein == EIN_ && net.H == H_) {                                                                               \
        hipLaunchKernelGGL((mlp_bwd_rows_kernel<EIN_, H_>), dim3(grid_rows), dim3(kThreads), 0, (hipStream_t)stream, \
                           args);                                                                                   \
        RPO_LAUNCH_CHECK();                                                                                         \
        if (param_grads) {                                                                                          \
            hipLaunchKernelGGL((mlp_bwd_weights_kernel<EIN_, H_>), dim3(grid_w), dim3(kThreads), 0,                 \
                               (hipStream_t)stream, args);                                                          \
            RPO_LAUNCH_CHECK();                                                                                     \
        }                                                                                                           \
        return 0;                                                                                                   \
    }
    RPO_MLP_BWD(128, 256)
    RPO_MLP_BWD(256, 256)
    RPO_MLP_BWD(512, 256)
    return RPO_ERR_ARG;
}

int rpo_mlp_backward_pair(const rpo_mlp* net1_host, const rpo_mlp_grad* grad1_host, const rpo_mlp* net2_host,
                          const rpo_mlp_grad* grad2_host, int n, const float* s, int s_stride, const float* a,
                          int a_stride, const float* x0_1, const float* h1_1, const float* dout_1, float* dh_1,
                          float* dx0_1, float* da_1, const float* x0_2, const float* h1_2, const float* dout_2, float* dh_2,
                          float* dx0_2, float* da_2, int param_grads, int first_layer_state_only, float* gradmax,
                          void* stream) {
    BwdArgs2 args;
    if (int e = make_bwd_args(args.net[0], net1_host, grad1_host, n, s, s_stride, a, a_stride, x0_1, h1_1, dout_1, dh_1,
                              dx0_1, da_1, param_grads, first_layer_state_only, gradmax))
        return e;
    if (int e = make_bwd_args(args.net[1], net2_host, grad2_host, n, s, s_stride, a, a_stride, x0_2, h1_2, dout_2, dh_2,
                              dx0_2, da_2, param_grads, first_layer_state_only, gradmax))
        return e;
    const Mlp &n1 = args.net[0].net, &n2 = args.net[1].net;
    if (n1.S != n2.S || n1.A != n2.A || n1.E != n2.E || n1.H != n2.H || n1.n_out != n2.n_out || n1.cat != n2.cat ||
        n1.hd != n2.hd)
        return RPO_ERR_ARG;
    const int ein = n1.cat ? 2 * n1.E : n1.E;
    const int grid_rows = (n + kRows - 1) / kRows;
    const int grid_w = bwd_weights_grid(n1, first_layer_state_only);
#define RPO_MLP_BWD2(EIN_, H_)                                                                                         \
    if (ein == EIN_ && n1.H == H_) {                                                                                   \
        hipLaunchKernelGGL((mlp_bwd_rows_kernel2<EIN_, H_>), dim3(grid_rows, 2), dim3(kThreads), 0, (hipStream_t)stream, \
                           args);                                                                                      \
        RPO_LAUNCH_CHECK();                                                                                            \
        if (param_grads) {                                                                                             \
            hipLaunchKernelGGL((mlp_bwd_weights_kernel2<EIN_, H_>), dim3(grid_w, 2), dim3(kThreads), 0,                \
                               (hipStream_t)stream, args);                                                             \
            RPO_LAUNCH_CHECK();                                                                                        \
        }                                                                                                              \
        return 0;                                                                                                      \
    }
    RPO_MLP_BWD2(128, 256)
    RPO_MLP_BWD2(256, 256)
    RPO_MLP_BWD2(512, 256)
    return RPO_ERR_ARG;
}

int rpo_tanh_box_bwd(int n, const float* dap, const float* ap_det, const float* noise, float eps_start, float eps_end,
                     float eps_decay, const long long* ctrl, float lo, float hi, float scale, float base, float* dout,
                     void* stream) {
    if (n <= 0 || scale == 0.0f) return RPO_ERR_ARG;
    if (!dap || !ap_det || !dout) return RPO_ERR_NULL;
    hipLaunchKernelGGL(tanh_box_bwd_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, dap,
                       ap_det, noise, eps_start, eps_end, eps_decay, ctrl, lo, hi, scale, base, dout);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_gauss_head(int n, const float* raw, const float* eps, float scale, float base, float lo, float hi,
                   int deterministic, float* ap_out, float* logp_out, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!raw || !eps || !ap_out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(gauss_head_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, raw, eps,
                       scale, base, lo, hi, deterministic, ap_out, logp_out);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_gauss_head_bwd(int n, const float* raw, const float* eps, const float* dap, float dlogp, float scale,
                       float base, float lo, float hi, float* draw, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!raw || !eps || !dap || !draw) return RPO_ERR_NULL;
    hipLaunchKernelGGL(gauss_head_bwd_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, raw,
                       eps, dap, dlogp, scale, base, lo, hi, draw);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
