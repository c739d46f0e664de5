// Device-side building block shared by the MLP kernels (mlp.hip) and the fused pipelines (fused.hip): one workgroup of
// 512 threads pushes RT tiles of 16 rows through first layer -> hidden layer (f32 MFMA) -> head.  See mlp.hip for the
// operand-layout notes.
//
// A workgroup has to pull the whole hidden-layer matrix W0 (128 KB at 128 -> 256) through its CU, which takes ~2.7 us
// at the ~47 GB/s one CU draws from L2 (measured).  RT = 1 (16 rows) maximises the number of workgroups for the
// batch-256 update kernels; RT = 4 (64 rows) amortises that stream over 4x the MFMA work and is used when there are
// enough rows to fill the chip anyway (vectorised rollouts, large batches).
#pragma once
#include "common.h"

namespace rpo_mlp_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;          // rows per MFMA tile
constexpr int kThreads = 256;      // backward kernels: 4 waves
constexpr int kFwdWaves = 8;       // forward: 512 threads, every wave owns H / 8 hidden columns
constexpr int kFwdThreads = kFwdWaves * 64;
constexpr int kInS = 64, kInA = 48;   // row strides of the LDS input tiles (S <= 64, A <= 48)

struct Mlp {
    const float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;   // W1b / b1b: second head (SAC log-std), n_out = 2
    int S, A, E, H, n_out, cat;
    int hd;   // outputs per head (0 / 1: scalar heads, reduced with wave shuffles; > 1: the head is a 16 x 16 MFMA tile)
    int l1_vec;   // 1: keep the first layer on the vector ALU (rpo_tuning(RPO_TUNE_L1_MFMA, 0): the A/B form of the tests)
};
struct MlpGrad {
    float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS working set of one forward over RT * 16 rows
template <int EIN, int RT = 1, int INS = kInS, int INA = kInA>
struct TileLds {
    static constexpr int ROWS = kRows * RT;
    static constexpr int kS = INS, kA = INA;                     // row strides of in_s / in_a (>= S, >= A)
    __attribute__((aligned(16))) float x1[ROWS * (EIN + 4)];    // relu(first layer), padded rows (ds_read_b128)
    float in_s[ROWS * INS];                                      // state inputs of the rows
    float in_a[ROWS * INA];                                      // action inputs of the rows
    float part[kFwdWaves * ROWS * 2];                            // per-wave head partials
    float out[ROWS * 2];                                         // outputs (after the epilogue)
};

constexpr int kWideOut = 32;        // outputs per row of a multi-output network (2 heads x <= 16)

// ... plus what a multi-output head (hd > 1: 14 basic actions of EVOPF-v0) needs: relu(h1) of the tile as an MFMA operand
template <int EIN, int H>
struct TileLdsWide : TileLds<EIN, 1, kInS, kInA> {
    __attribute__((aligned(16))) float hr[kRows * (H + 4)];
    float outw[kRows * kWideOut];
};

// Everything a thread keeps in registers for one network: its first-layer column, its wave's slice of W0 and the head
// vectors.  Loaded once (tile_load_weights) and reused for any number of row tiles (tile_compute): a persistent
// workgroup streams the 128 KB hidden-layer matrix once instead of once per tile.
template <int EIN, int H, int RT>
struct TileWeights {
    static constexpr int NT = H / (16 * kFwdWaves);              // 16-column tiles per wave
    static constexpr int ITS = EIN / 16;                         // k-groups of 16
    static constexpr int PRE_MAX = RT > 1 ? 4 : 16;           // (64-row form: 120 registers = TWO workgroups per CU; the other k-groups stream from L2)           // (64-row form: 128 registers = two workgroups per CU; the last k-groups stream from L2)
    static constexpr int PRE = ITS < PRE_MAX ? ITS : PRE_MAX;    // k-groups kept in registers
    static constexpr int TPW = EIN / (16 * kFwdWaves);           // first-layer column tiles per wave (matrix-core form)
    float bias;
    float ws0[8], wa0[8];                                        // first chunk of the column's first-layer weights
    float wk[TPW][3];                                            // matrix-core form: A operands of the three k-steps, per tile
    float4 wpre[PRE][NT];
    float b0v[NT], w1av[NT], w1bv[NT];
    float b1v;
};

// The first layer on the matrix cores (round 5; the streaming kernels of mlp_stream.h had it first): x0^T[e][row] =
// Wcat[e][u] in^T[u][row] with the k slots u = (bs x 1 | ba x 1 | state inputs | zero pad || action inputs | zero pad) as three
// 16 x 16 x 4 steps per 16 columns.  The biases enter as the FIRST terms -- (0 + bs) + ba is the vector form's bias = bs + ba,
// exactly -- so the MFMA is the k-ordered fmaf chain the vector form was (bias, state inputs in order, action inputs in order;
// zero pads are exact no-ops) and the bits do not change; the vector form's scalar LDS reads and run-time bounds on 16 + 16
// unrolled steps become a dozen instructions.  Every operand is ONE predicated load and nothing is computed from loaded values
// where the weights are requested (nsplit_dev.h NsWeights has the measurements behind that rule).  For 128-wide networks
// without concatenation, S <= 6, A <= 4: every classic-control network
// (the wider row-tile kernels are a cold path -- EVOPF's networks run layer by layer, mlp_gemm.h -- and keep the vector form).
template <int EIN>
__device__ __forceinline__ bool tile_l1_mfma(const Mlp& net) {
    return EIN == 128 && !net.cat && net.S <= 6 && net.A <= 4 && net.S > 0 && !net.l1_vec;
}

#ifndef RPO_TILE_SKIP
#define RPO_TILE_SKIP 0            // timing-only builds (tools/probe/build_stream_variants.sh KIND=rollout TILE=N): 1 no weight loads,
#endif                             // 2 no first layer, 4 no MFMA loop, 8 no head reduction

template <int EIN, int H, int RT>
__device__ __forceinline__ void tile_load_weights(const Mlp& net, TileWeights<EIN, H, RT>& w) {
    constexpr int ROWS = kRows * RT;
    constexpr int NT = TileWeights<EIN, H, RT>::NT, PRE = TileWeights<EIN, H, RT>::PRE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    if (RPO_TILE_SKIP & 1) {
        w.bias = 0.01f * (float)li;
#pragma unroll
        for (int u = 0; u < 8; ++u) { w.ws0[u] = 0.1f; w.wa0[u] = 0.1f; }
#pragma unroll
        for (int it = 0; it < PRE; ++it)
#pragma unroll
            for (int c = 0; c < NT; ++c) w.wpre[it][c] = make_float4(0.01f, -0.01f, 0.02f, 0.01f * (float)lg);
#pragma unroll
        for (int c = 0; c < NT; ++c) { w.b0v[c] = 0.1f; w.w1av[c] = 0.05f; w.w1bv[c] = 0.0f; }
        w.b1v = 0.0f;
        return;
    }
    // ---- layer-1 operands of this thread first: they are what layer 1 waits for (returns are in order)
    if (tile_l1_mfma<EIN>(net)) {
        constexpr int TPW = TileWeights<EIN, H, RT>::TPW;
        const bool add_a = net.A > 0;
        w.bias = 0.0f;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int e = (wave + t * kFwdWaves) * 16 + li;      // this lane's row of the A operand; k slots 0 / 1: the biases
            const float* p0 = lg == 0 ? net.bs + e : (lg == 1 ? net.ba + e : net.Ws + e * net.S + (lg - 2));
            const bool live0 = lg == 0 || (lg == 1 ? add_a : lg - 2 < net.S);
            w.wk[t][0] = live0 ? *p0 : 0.0f;
            w.wk[t][1] = 2 + lg < net.S ? net.Ws[e * net.S + 2 + lg] : 0.0f;
            w.wk[t][2] = (add_a && lg < net.A) ? net.Wa[e * net.A + lg] : 0.0f;
        }
    } else {
        const int e_col = tid % EIN;
        const bool act_part = net.cat && e_col >= net.E;        // concatenating critic: columns [E, 2E) embed the action
        const int er = act_part ? e_col - net.E : e_col;
        const bool use_s = !act_part, use_a = net.A > 0 && (act_part || !net.cat);
        w.bias = act_part ? net.ba[er] : (net.bs[er] + ((net.A > 0 && !net.cat) ? net.ba[er] : 0.0f));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w.ws0[u] = (use_s && u < net.S) ? net.Ws[er * net.S + u] : 0.0f;
            w.wa0[u] = (use_a && u < net.A) ? net.Wa[er * net.A + u] : 0.0f;
        }
    }
    // ---- then the wave's slice of W0: it streams in underneath layer 1 and feeds the MFMA loop k-group by k-group
    const int j0 = wave * (H / kFwdWaves);
#pragma unroll
    for (int it = 0; it < PRE; ++it)
#pragma unroll
        for (int c = 0; c < NT; ++c)
            w.wpre[it][c] = *reinterpret_cast<const float4*>(&net.W0[(size_t)(j0 + c * 16 + li) * EIN + it * 16 + lg * 4]);
    // ---- what the epilogue needs is requested last
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        w.b0v[c] = net.b0[j0 + c * 16 + li];
        w.w1av[c] = net.hd > 1 ? 0.0f : net.W1[j0 + c * 16 + li];
        w.w1bv[c] = (net.n_out > 1 && net.hd <= 1) ? net.W1b[j0 + c * 16 + li] : 0.0f;
    }
    w.b1v = (net.hd <= 1 && tid < ROWS * net.n_out) ? (((tid % net.n_out) == 0) ? net.b1[0] : net.b1b[0]) : 0.0f;
}

// Forward of the RT * 16 rows whose inputs sit in lds.in_s / lds.in_a (the caller wrote them; this function
// synchronises before reading).  Outputs land in lds.out[r * 2 + o] and are visible to every thread on return.
// out_mode 1: BoxConstraint's tanh map on output 0.  x0_save / h1_save (global, may be NULL) get the pre-activations
// of rows row0 + r < n.
template <int EIN, int H, int RT, class LDS, bool WIDE = false>
__device__ __forceinline__ void tile_compute(const Mlp& net, const TileWeights<EIN, H, RT>& w, LDS& lds, int row0, int n,
                                             float* x0_save, float* h1_save, int out_mode, float scale, float base) {
    constexpr int kInS = LDS::kS, kInA = LDS::kA;               // (shadow the default strides)
    constexpr int ROWS = kRows * RT;
    constexpr int LDX = EIN + 4;
    constexpr int GROUPS = kFwdThreads / EIN;                   // thread groups that split the rows in layer 1
    constexpr int RPT = ROWS / GROUPS;                          // rows per thread in layer 1
    static_assert(kFwdThreads % EIN == 0 && ROWS % GROUPS == 0, "layer-1 thread mapping");
    constexpr int NT = TileWeights<EIN, H, RT>::NT, ITS = TileWeights<EIN, H, RT>::ITS, PRE = TileWeights<EIN, H, RT>::PRE;
    float* x1 = lds.x1;
    const float* in_s = lds.in_s;
    const float* in_a = lds.in_a;
    float* part = lds.part;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int e_col = tid % EIN, r_lo = (tid / EIN) * RPT;
    const bool act_part = net.cat && e_col >= net.E;
    const int er = act_part ? e_col - net.E : e_col;
    const bool use_s = !act_part, use_a = net.A > 0 && (act_part || !net.cat);
    const int j0 = wave * (H / kFwdWaves);
    float acc1[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc1[r] = w.bias;
    __syncthreads();
    const bool l1_mfma = tile_l1_mfma<EIN>(net);
    if (l1_mfma) {
        // ---- layer 1 (MFMA, see tile_l1_mfma): wave w owns column tiles w, w + 8, ... for every row tile
        constexpr int TPW = TileWeights<EIN, H, RT>::TPW;
        const bool x0_vec = (reinterpret_cast<uintptr_t>(x0_save) & 15u) == 0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rt * kRows + li;
            const float b0 = lg < 2 ? 1.0f : (lg - 2 < net.S ? in_s[row * kInS + lg - 2] : 0.0f);
            const float b1 = 2 + lg < net.S ? in_s[row * kInS + 2 + lg] : 0.0f;
            const float b2 = (net.A > 0 && lg < net.A) ? in_a[row * kInA + lg] : 0.0f;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                f32x4 c = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (!(RPO_TILE_SKIP & 2)) {
                    c = mfma4(w.wk[t][0], b0, c);
                    c = mfma4(w.wk[t][1], b1, c);
                    c = mfma4(w.wk[t][2], b2, c);
                }
                // c[i] = x0[row][e0 + i], e0 = 16 (w + 8 t) + 4 lg
                const int e0 = (wave + t * kFwdWaves) * 16 + 4 * lg;
                if (x0_save && row0 + row < n) {
                    float* dst = x0_save + (size_t)(row0 + row) * EIN + e0;
                    if (x0_vec) *reinterpret_cast<f32x4*>(dst) = c;
                    else { dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; dst[3] = c[3]; }
                }
                *reinterpret_cast<f32x4*>(&x1[row * LDX + e0]) =
                    f32x4{fmaxf(c[0], 0.0f), fmaxf(c[1], 0.0f), fmaxf(c[2], 0.0f), fmaxf(c[3], 0.0f)};
            }
        }
    }

    // ---- layer 1 (VALU): x0[r][e]
    if (!l1_mfma && use_s && !(RPO_TILE_SKIP & 2)) {
        for (int i0 = 0; i0 < net.S; i0 += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = (i0 == 0) ? w.ws0[u] : ((i0 + u < net.S) ? net.Ws[er * net.S + i0 + u] : 0.0f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < net.S) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) acc1[r] = fmaf(in_s[(r_lo + r) * kInS + i0 + u], wv[u], acc1[r]);
                }
            }
        }
    }
    if (!l1_mfma && use_a && !(RPO_TILE_SKIP & 2)) {
        for (int i0 = 0; i0 < net.A; i0 += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = (i0 == 0) ? w.wa0[u] : ((i0 + u < net.A) ? net.Wa[er * net.A + i0 + u] : 0.0f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (i0 + u < net.A) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) acc1[r] = fmaf(in_a[(r_lo + r) * kInA + i0 + u], wv[u], acc1[r]);
                }
            }
        }
    }
    if (!l1_mfma) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (x0_save && row0 + r_lo + r < n) x0_save[(size_t)(row0 + r_lo + r) * EIN + e_col] = acc1[r];
            x1[(r_lo + r) * LDX + e_col] = fmaxf(acc1[r], 0.0f);
        }
    }
    __syncthreads();

    // ---- layer 2 (MFMA): wave w owns hidden columns [w*H/8, (w+1)*H/8) = NT tiles of 16, for all RT row tiles
    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[rt][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int it = 0; it < ((RPO_TILE_SKIP & 4) ? 1 : ITS); ++it) {
        float4 a4[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
            a4[rt] = *reinterpret_cast<const float4*>(&x1[(rt * kRows + li) * LDX + it * 16 + lg * 4]);
        float4 b4[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
            b4[c] = (it < PRE) ? w.wpre[it < PRE ? it : 0][c]
                               : *reinterpret_cast<const float4*>(&net.W0[(size_t)(j0 + c * 16 + li) * EIN + it * 16 + lg * 4]);
        // consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[rt][c] = mfma4(a4[rt].x, b4[c].x, acc[rt][c]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[rt][c] = mfma4(a4[rt].y, b4[c].y, acc[rt][c]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[rt][c] = mfma4(a4[rt].z, b4[c].z, acc[rt][c]);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[rt][c] = mfma4(a4[rt].w, b4[c].w, acc[rt][c]);
    }
    // acc[rt][c][i] = h1[row = 16 rt + 4 lg + i][col = j0 + 16c + li] (before bias)
    if constexpr (WIDE) {
        // ---- multi-output head: relu(h1) of the tile becomes the A operand of one more MFMA tile per head
        static_assert(RT == 1, "multi-output heads run on 16-row tiles");
        constexpr int LDH = H + 4;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int col = j0 + c * 16 + li;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h = acc[0][c][i] + w.b0v[c];
                const int row = row0 + lg * 4 + i;
                if (h1_save && row < n) h1_save[(size_t)row * H + col] = h;
                lds.hr[(lg * 4 + i) * LDH + col] = fmaxf(h, 0.0f);
            }
        }
        __syncthreads();
        if (wave < net.n_out) {                                    // wave k computes head k: out[r][o] = hr[r][:] . W1_k[o][:]
            const float* W = wave == 0 ? net.W1 : net.W1b;
            const float* bvec = wave == 0 ? net.b1 : net.b1b;
            const bool live = li < net.hd;
            f32x4 o4 = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int it = 0; it < H / 16; ++it) {
                const float4 a4 = *reinterpret_cast<const float4*>(&lds.hr[li * LDH + it * 16 + lg * 4]);
                float4 b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (live) b4 = *reinterpret_cast<const float4*>(&W[(size_t)li * H + it * 16 + lg * 4]);
                o4 = mfma4(a4.x, b4.x, o4);
                o4 = mfma4(a4.y, b4.y, o4);
                o4 = mfma4(a4.z, b4.z, o4);
                o4 = mfma4(a4.w, b4.w, o4);
            }
            if (live) {
                const float bias = bvec[li];
#pragma unroll
                for (int i = 0; i < 4; ++i) lds.outw[(lg * 4 + i) * kWideOut + wave * net.hd + li] = o4[i] + bias;
            }
        }
        __syncthreads();
        return;
    }
    // ---- head: per-lane partial dot products, reduced over the 16 lanes that share lg, then over the waves
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float po[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int col = j0 + c * 16 + li;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h = acc[rt][c][i] + w.b0v[c];
                const int row = row0 + rt * kRows + lg * 4 + i;
                if (h1_save && row < n) h1_save[(size_t)row * H + col] = h;
                const float hr = fmaxf(h, 0.0f);
                po[0][i] = fmaf(hr, w.w1av[c], po[0][i]);
                po[1][i] = fmaf(hr, w.w1bv[c], po[1][i]);
            }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = po[o][i];
                if (!(RPO_TILE_SKIP & 8)) v = rpo_row16_sum_lane0(v);   // (same association as the xor butterfly, at li == 0)
                if (li == 0) part[(wave * ROWS + rt * kRows + lg * 4 + i) * 2 + o] = v;
            }
    }
    __syncthreads();
    if (tid < ROWS * net.n_out) {
        const int r = tid / net.n_out, o = tid - r * net.n_out;
        float v = w.b1v;
        for (int ww = 0; ww < kFwdWaves; ++ww) v += part[(ww * ROWS + r) * 2 + o];   // fixed order
        if (out_mode == 1 && o == 0) v = scale * tanhf(v) + base;
        lds.out[r * 2 + o] = v;
    }
    __syncthreads();
}

// load + compute for one tile
template <int EIN, int H, int RT = 1, class LDS = TileLds<EIN, RT>>
__device__ __forceinline__ void mlp_tile_forward(const Mlp& net, LDS& lds, int row0, int n, float* x0_save,
                                                 float* h1_save, int out_mode, float scale, float base) {
    TileWeights<EIN, H, RT> w;
    tile_load_weights<EIN, H, RT>(net, w);
    tile_compute<EIN, H, RT, LDS>(net, w, lds, row0, n, x0_save, h1_save, out_mode, scale, base);
}

}  // namespace rpo_mlp_dev
