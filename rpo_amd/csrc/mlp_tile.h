// Device-side building block shared by the MLP kernels (mlp.hip) and the fused pipelines (fused.hip): one workgroup of
// 512 threads pushes a tile of 16 rows through first layer -> hidden layer (f32 MFMA) -> head.  See mlp.hip for the
// operand-layout notes.
#pragma once
#include "common.h"

namespace rpo_mlp_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 16;          // rows (samples) per workgroup
constexpr int kThreads = 256;      // backward kernels: 4 waves
constexpr int kFwdWaves = 8;       // forward: 512 threads, every wave owns H / 8 hidden columns
constexpr int kFwdThreads = kFwdWaves * 64;
constexpr int kInS = 64, kInA = 48;   // row strides of the LDS input tiles (S <= 64, A <= 48)

struct Mlp {
    const float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;   // W1b / b1b: second head (SAC log-std), n_out = 2
    int S, A, E, H, n_out, cat;
};
struct MlpGrad {
    float *Ws, *bs, *Wa, *ba, *W0, *b0, *W1, *b1, *W1b, *b1b;
};

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS working set of one tile forward
template <int EIN>
struct TileLds {
    __attribute__((aligned(16))) float x1[kRows * (EIN + 4)];   // relu(first layer), padded rows (ds_read_b128)
    float in_s[kRows * kInS];                                    // state inputs of the tile
    float in_a[kRows * kInA];                                    // action inputs of the tile
    float part[kFwdWaves * kRows * 2];                           // per-wave head partials
    float out[kRows * 2];                                        // outputs of the tile (after the epilogue)
};

// Forward of the 16 rows whose inputs sit in lds.in_s / lds.in_a (the caller wrote them; this function synchronises
// before reading).  Outputs land in lds.out[r * 2 + o] and are visible to every thread on return.
// out_mode 1: BoxConstraint's tanh map on output 0.  x0_save / h1_save (global, may be NULL) get the pre-activations
// of rows row0 + r < n.
template <int EIN, int H>
__device__ __forceinline__ void mlp_tile_forward(const Mlp& net, TileLds<EIN>& lds, int row0, int n, float* x0_save,
                                                 float* h1_save, int out_mode, float scale, float base) {
    constexpr int LDX = EIN + 4;
    float* x1 = lds.x1;
    const float* in_s = lds.in_s;
    const float* in_a = lds.in_a;
    float* part = lds.part;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    // first-layer weights of this thread's embedding column, fetched in chunks of 8 before they are needed
    const int e_col = tid;                                      // EIN <= 512 == kFwdThreads: one column per thread
    const bool has_col = e_col < EIN;
    const bool act_part = has_col && net.cat && e_col >= net.E; // concatenating critic: columns [E, 2E) embed the action
    const int er = act_part ? e_col - net.E : e_col;
    float acc1[kRows];
    float ws0[8], wa0[8];                                       // first chunk of this column's first-layer weights
    {
        const float bias = !has_col ? 0.0f
                                    : (act_part ? net.ba[er] : (net.bs[er] + ((net.A > 0 && !net.cat) ? net.ba[er] : 0.0f)));
#pragma unroll
        for (int r = 0; r < kRows; ++r) acc1[r] = bias;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ws0[u] = (has_col && !act_part && u < net.S) ? net.Ws[er * net.S + u] : 0.0f;
            wa0[u] = (has_col && net.A > 0 && (act_part || !net.cat) && u < net.A) ? net.Wa[er * net.A + u] : 0.0f;
        }
    }
    // ---- then the wave's slice of W0 (128 KB per workgroup, ~2.7 us at the ~47 GB/s one CU pulls from L2): requested
    //      AFTER the few loads layer 1 waits for (returns are in order), it streams in underneath layer 1 and feeds
    //      the MFMA loop k-group by k-group
    constexpr int NT = H / (16 * kFwdWaves);                    // 16-column tiles per wave
    constexpr int ITS = EIN / 16;                               // k-groups of 16
    constexpr int PRE = ITS < 16 ? ITS : 16;                    // k-groups kept in registers up front
    const int j0 = wave * (H / kFwdWaves);
    float4 wpre[PRE][NT];
#pragma unroll
    for (int it = 0; it < PRE; ++it)
#pragma unroll
        for (int c = 0; c < NT; ++c)
            wpre[it][c] = *reinterpret_cast<const float4*>(&net.W0[(size_t)(j0 + c * 16 + li) * EIN + it * 16 + lg * 4]);

    // what the epilogue needs is requested last
    float b0v[NT], w1av[NT], w1bv[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        b0v[c] = net.b0[j0 + c * 16 + li];
        w1av[c] = net.W1[j0 + c * 16 + li];
        w1bv[c] = net.n_out > 1 ? net.W1b[j0 + c * 16 + li] : 0.0f;
    }
    const float b1v = (tid < kRows * net.n_out) ? (((tid % net.n_out) == 0) ? net.b1[0] : net.b1b[0]) : 0.0f;

    __syncthreads();

    // ---- layer 1 (VALU): x0[r][e]
    if (has_col) {
        if (!act_part) {
            for (int i0 = 0; i0 < net.S; i0 += 8) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = (i0 == 0) ? ws0[u] : ((i0 + u < net.S) ? net.Ws[er * net.S + i0 + u] : 0.0f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < net.S) {
#pragma unroll
                        for (int r = 0; r < kRows; ++r) acc1[r] = fmaf(in_s[r * kInS + i0 + u], w[u], acc1[r]);
                    }
                }
            }
        }
        if (net.A > 0 && (act_part || !net.cat)) {
            for (int i0 = 0; i0 < net.A; i0 += 8) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = (i0 == 0) ? wa0[u] : ((i0 + u < net.A) ? net.Wa[er * net.A + i0 + u] : 0.0f);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < net.A) {
#pragma unroll
                        for (int r = 0; r < kRows; ++r) acc1[r] = fmaf(in_a[r * kInA + i0 + u], w[u], acc1[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            if (x0_save && row0 + r < n) x0_save[(size_t)(row0 + r) * EIN + e_col] = acc1[r];
            x1[r * LDX + e_col] = fmaxf(acc1[r], 0.0f);
        }
    }
    __syncthreads();

    // ---- layer 2 (MFMA): wave w owns hidden columns [w*H/8, (w+1)*H/8) = NT tiles of 16
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const float4 a4 = *reinterpret_cast<const float4*>(&x1[li * LDX + it * 16 + lg * 4]);
        float4 b4[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c)
            b4[c] = (it < PRE) ? wpre[it < PRE ? it : 0][c]
                               : *reinterpret_cast<const float4*>(&net.W0[(size_t)(j0 + c * 16 + li) * EIN + it * 16 + lg * 4]);
        // consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = mfma4(a4.x, b4[c].x, acc[c]);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = mfma4(a4.y, b4[c].y, acc[c]);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = mfma4(a4.z, b4[c].z, acc[c]);
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[c] = mfma4(a4.w, b4[c].w, acc[c]);
    }
    // acc[c][i] = h1[row = 4*lg + i][col = j0 + 16c + li] (before bias)
    float po[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int col = j0 + c * 16 + li;
        const float b0 = b0v[c];
        const float w1a = w1av[c], w1b = w1bv[c];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h = acc[c][i] + b0;
            const int row = row0 + lg * 4 + i;
            if (h1_save && row < n) h1_save[(size_t)row * H + col] = h;
            const float hr = fmaxf(h, 0.0f);
            po[0][i] = fmaf(hr, w1a, po[0][i]);
            po[1][i] = fmaf(hr, w1b, po[1][i]);
        }
    }
    // ---- head: reduce over the 16 lanes that share lg, then over the waves (fixed order)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = po[o][i];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (li == 0) part[(wave * kRows + lg * 4 + i) * 2 + o] = v;
        }
    __syncthreads();
    if (tid < kRows * net.n_out) {
        const int r = tid / net.n_out, o = tid - r * net.n_out;
        float v = b1v;
        for (int w = 0; w < kFwdWaves; ++w) v += part[(w * kRows + r) * 2 + o];
        if (out_mode == 1 && o == 0) v = scale * tanhf(v) + base;
        lds.out[r * 2 + o] = v;
    }
    __syncthreads();
}

}  // namespace rpo_mlp_dev
