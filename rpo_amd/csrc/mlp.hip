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
#include <stdlib.h>

#include "heads_dev.h"
#include "mlp_bwd.h"
#include "mlp_gemm.h"
#include "mlp_tile.h"
#include "mlp_stream.h"

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
__device__ __forceinline__ void mlp_forward_body(const FwdArgs& p) {
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

// (HIP: the second bound is waves per SIMD -- four = two workgroups per CU; at 146 registers the 64-row form ran ONE -- every phase of a tile exposed, 0.48 of the MFMA peak)
template <int EIN, int H, int RT, int INS, int INA>
__global__ __launch_bounds__(kFwdThreads, 4) void mlp_forward_kernel(FwdArgs p) { mlp_forward_body<EIN, H, RT, INS, INA>(p); }

// Up to four same-shaped networks on their own inputs in one launch (gridDim.y = network): Q_targ(s', a') and Q(s, a) of a
// critic update (and SAC's twins) are independent of each other.
struct FwdArgs4 {
    FwdArgs net[4];
};
template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void mlp_forward_multi_kernel(FwdArgs4 p) {
    mlp_forward_body<EIN, H, 1, kInS, kInA>(p.net[blockIdx.y]);
}
template <int H>
__global__ __launch_bounds__(kFwdThreads, 4) void mlp_forward_multi64_kernel(FwdArgs4 p) {     // 64 rows per workgroup (large n)
    mlp_forward_body<128, H, 4, 8, 8>(p.net[blockIdx.y]);
}

// Large n: weights stationary in LDS, rows stream through independent waves (mlp_stream.h); gridDim.y = network.
template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void mlp_forward_stream_kernel(FwdArgs4 p) {
    mlp_forward_stream_body<256, NW>(p);
}

static int stream_cus() { return rpo_cu_count(); }

// the streaming forward applies: large n, every network of the launch 128 -> 256 with scalar heads and <= 11 inputs
static bool stream_applies(const FwdArgs4& a, int count, int n) {
    if (!rpo_tune(RPO_TUNE_FWD_STREAM) || n < 64 * 192) return false;
    for (int k = 0; k < count; ++k) {
        if (!stream_shape_ok(a.net[k].net)) return false;
        // (the pre-activations leave as 16-byte stores: buffers that are not 16-byte aligned keep the row-tile kernels)
        if ((reinterpret_cast<uintptr_t>(a.net[k].x0_save) | reinterpret_cast<uintptr_t>(a.net[k].h1_save)) & 15u) return false;
    }
    return true;
}

static int launch_stream(const FwdArgs4& a, int count, int n, hipStream_t stream) {
    const int nw = rpo_tune(RPO_TUNE_FWD_STREAM_WAVES) == 12 ? 12 : 16;
    const int tiles = (n + kRows - 1) / kRows;
    int gx = stream_cus() / count;                              // one persistent workgroup per CU (LDS: 150 KB each)
    if (gx < 1) gx = 1;
    if (gx > (tiles + nw - 1) / nw) gx = (tiles + nw - 1) / nw;
    if (nw == 16) hipLaunchKernelGGL((mlp_forward_stream_kernel<16>), dim3(gx, count), dim3(16 * 64), 0, stream, a);
    else hipLaunchKernelGGL((mlp_forward_stream_kernel<12>), dim3(gx, count), dim3(12 * 64), 0, stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
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

template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void mlp_bwd_rows_kernel(BwdArgs p) { mlp_bwd_rows_body<EIN, H>(p); }
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
        dout[i] = rpo_head_dev::tanh_box_bwd_row(dap[i], ap_det[i], noise ? noise[i] : 0.0f, noise != nullptr, eps_t, lo, hi,
                                                 scale, base);
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
        reinterpret_cast<float2*>(draw)[i] = rpo_head_dev::gauss_head_bwd_row(r.x, r.y, eps[i], dap[i], dlogp, scale, base, lo, hi);
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
            net_host->n_out, net_host->cat, net_host->head_dim, rpo_tune(RPO_TUNE_L1_MFMA) ? 0 : 1};
    if (int e = check_net(net)) return e;
    if (n <= 0 || s_stride < net.S || (net.A > 0 && a_stride < net.A)) return RPO_ERR_ARG;
    if (!s || !out || (net.A > 0 && !a)) return RPO_ERR_NULL;
    const int ein = net.cat ? 2 * net.E : net.E;
    FwdArgs args{net, n, s, s_stride, a, a_stride, out, x0_save, h1_save, out_mode, scale, base};
    if (gemm_path_ok(net, n, x0_save, h1_save, out_mode)) {     // wide networks: one small GEMM launch per layer (mlp_gemm.h)
        const GemmFwd f{net, n, s, s_stride, a, a_stride, out, x0_save, h1_save};
        return gemm_forward(&f, 1, (hipStream_t)stream);
    }
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
    {
        FwdArgs4 one{};
        one.net[0] = args;
        if (stream_applies(one, 1, n)) return launch_stream(one, 1, n, (hipStream_t)stream);
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
                         float* dh, float* dx0, float* da, int param_grads, int first_layer_state_only, float* gradmax,
                         const rpo_td* td) {
    if (!net_host) return RPO_ERR_NULL;
    Mlp net{net_host->Ws, net_host->bs, net_host->Wa, net_host->ba, net_host->W0, net_host->b0, net_host->W1,
            net_host->b1, net_host->W1b, net_host->b1b, net_host->S, net_host->A, net_host->E, net_host->H,
            net_host->n_out, net_host->cat, net_host->head_dim};
    if (int e = check_net(net)) return e;
    if (n <= 0 || s_stride < net.S || (net.A > 0 && a_stride < net.A)) return RPO_ERR_ARG;
    TdArgs t{};
    if (td) {                                                   // TD / Huber prologue: dout is produced, not read
        if (net.n_out != 1 || net.hd > 1) return RPO_ERR_ARG;
        if (!td->q || !td->qn1 || !td->reward || !td->done || !td->dq_out || !td->loss_partial) return RPO_ERR_NULL;
        if (td->reward_stride <= 0 || td->done_stride <= 0) return RPO_ERR_ARG;
        t = TdArgs{td->q, td->qn1, td->qn2, td->logp, td->reward, td->reward_stride, td->done, td->done_stride, td->alpha,
                   td->gamma, td->dq_out, td->loss_partial};
        dout = td->dq_out;
    }
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
                   gradmax, t};
    return 0;
}

static int bwd_weights_grid(const Mlp& net, int first_layer_state_only) {
    const int ein = net.cat ? 2 * net.E : net.E;
    const int fl_outputs = net.E * (net.S + 1 + ((net.A > 0 && !first_layer_state_only) ? net.A + 1 : 0));
    return (net.H / 16) * (ein / 64) + (net.hd > 1 ? net.H / 16 : net.H / 64) + mlp_fl_blocks(fl_outputs);
}

int rpo_mlp_forward_multi(int count, const rpo_mlp* const* nets, int n, const float* const* s, const int* s_stride,
                          const float* const* a, const int* a_stride, float* const* out, float* const* x0_save,
                          float* const* h1_save, void* stream) {
    if (count < 1 || count > 4 || n <= 0) return RPO_ERR_ARG;
    if (!nets || !s || !s_stride || !a || !a_stride || !out || !x0_save || !h1_save) return RPO_ERR_NULL;
    FwdArgs4 args;
    for (int k = 0; k < count; ++k) {
        const rpo_mlp* h = nets[k];
        if (!h) return RPO_ERR_NULL;
        Mlp net{h->Ws, h->bs, h->Wa, h->ba, h->W0, h->b0, h->W1, h->b1, h->W1b, h->b1b, h->S, h->A, h->E, h->H,
                h->n_out, h->cat, h->head_dim, rpo_tune(RPO_TUNE_L1_MFMA) ? 0 : 1};
        if (int e = check_net(net)) return e;
        if (net.hd > 1 || s_stride[k] < net.S || (net.A > 0 && a_stride[k] < net.A)) return RPO_ERR_ARG;
        if (!s[k] || !out[k] || (net.A > 0 && !a[k])) return RPO_ERR_NULL;
        if (k > 0) {
            const Mlp& f = args.net[0].net;
            if (net.S != f.S || net.A != f.A || net.E != f.E || net.H != f.H || net.n_out != f.n_out || net.cat != f.cat)
                return RPO_ERR_ARG;
        }
        args.net[k] = FwdArgs{net, n, s[k], s_stride[k], a[k], a_stride[k], out[k], x0_save[k], h1_save[k], 0, 1.0f, 0.0f};
    }
    const Mlp& net = args.net[0].net;
    const int ein = net.cat ? 2 * net.E : net.E;
    {
        bool saved = true;
        for (int k = 0; k < count; ++k) saved = saved && x0_save[k] && h1_save[k];
        if (saved && gemm_fits(net, count) && gemm_path_ok(net, n, x0_save[0], h1_save[0], 0)) {
            GemmFwd f[4];
            for (int k = 0; k < count; ++k)
                f[k] = GemmFwd{args.net[k].net, n, s[k], s_stride[k], a[k], a_stride[k], out[k], x0_save[k], h1_save[k]};
            return gemm_forward(f, count, (hipStream_t)stream);
        }
    }
    if (stream_applies(args, count, n)) return launch_stream(args, count, n, (hipStream_t)stream);
    if (n >= 64 * 192 && net.S <= 8 && net.A <= 8 && ein == 128 && net.H == 256) {   // as rpo_mlp_forward: 64-row tiles
        hipLaunchKernelGGL((mlp_forward_multi64_kernel<256>), dim3((n + 63) / 64, count), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, args);
        RPO_LAUNCH_CHECK();
        return 0;
    }
#define RPO_MLP_FWD_MULTI(EIN_, H_)                                                                                  \
    if (ein == EIN_ && net.H == H_) {                                                                                \
        hipLaunchKernelGGL((mlp_forward_multi_kernel<EIN_, H_>), dim3((n + kRows - 1) / kRows, count),               \
                           dim3(kFwdThreads), 0, (hipStream_t)stream, args);                                         \
        RPO_LAUNCH_CHECK();                                                                                          \
        return 0;                                                                                                    \
    }
    RPO_MLP_FWD_MULTI(128, 256)
    RPO_MLP_FWD_MULTI(256, 256)
    RPO_MLP_FWD_MULTI(512, 256)
    return RPO_ERR_ARG;
}

int rpo_mlp_backward(const rpo_mlp* net_host, const rpo_mlp_grad* grad_host, int n, const float* s, int s_stride,
                     const float* a, int a_stride, const float* x0, const float* h1, const float* dout, float* dh,
                     float* dx0, float* da, int param_grads, int first_layer_state_only, float* gradmax,
                     const rpo_td* td, void* stream) {
    BwdArgs args;
    if (int e = make_bwd_args(args, net_host, grad_host, n, s, s_stride, a, a_stride, x0, h1, dout, dh, dx0, da,
                              param_grads, first_layer_state_only, gradmax, td))
        return e;
    const Mlp& net = args.net;
    const int ein = net.cat ? 2 * net.E : net.E;
    const int grid_rows = (n + kRows - 1) / kRows;
    const int grid_w = bwd_weights_grid(net, first_layer_state_only);
    const bool gemm_rows = gemm_path_ok(net, n, x0, h1, 0);       // wide networks: dh | dx0 | da as GEMM launches (mlp_gemm.h)
    if (gemm_rows)
        if (int e = gemm_backward_rows(&args, 1, (hipStream_t)stream)) return e;
    if (ein == 128 && net.H == 256 && !gemm_rows) {               // large batches: the two streaming launches (mlp_bwd_stream.h)
        SplitK sk{nullptr, nullptr, 0, 0};
        if (param_grads) sk = splitk_plan(args, grad_host->splitk_scratch, grad_host->splitk_floats);
        if (bwd_stream_applies_x(args, sk)) return launch_bwd_stream_x(args, sk, (hipStream_t)stream);
    }
    if (param_grads && ein == 128 && net.H == 256) {              // ... or rows + weights in ONE pass over the activations (round 3)
        const SplitK sk = splitk_plan(args, grad_host->splitk_scratch, grad_host->splitk_floats);
        if (sk.Z > 0 && onepass_applies<128, 256>(args, sk)) return launch_onepass<128, 256>(args, sk, (hipStream_t)stream);
    }
#define RPO_MLP_BWD(EIN_, H_)                                                                                       \
    if (ein == EIN_ && net.H == H_) {                                                                               \
        if (!gemm_rows)                                                                                             \
            hipLaunchKernelGGL((mlp_bwd_rows_kernel<EIN_, H_>), dim3(grid_rows), dim3(kThreads), 0, (hipStream_t)stream, \
                               args);                                                                               \
        RPO_LAUNCH_CHECK();                                                                                         \
        if (param_grads) {                                                                                          \
            const SplitK sk = splitk_plan(args, grad_host->splitk_scratch, grad_host->splitk_floats);                 \
            if (sk.Z > 0) return launch_weights_splitk<EIN_, H_>(args, sk, grid_w, (hipStream_t)stream);            \
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
                          const rpo_td* td1, const rpo_td* td2, void* stream) {
    BwdArgs2 args;
    if (int e = make_bwd_args(args.net[0], net1_host, grad1_host, n, s, s_stride, a, a_stride, x0_1, h1_1, dout_1, dh_1,
                              dx0_1, da_1, param_grads, first_layer_state_only, gradmax, td1))
        return e;
    if (int e = make_bwd_args(args.net[1], net2_host, grad2_host, n, s, s_stride, a, a_stride, x0_2, h1_2, dout_2, dh_2,
                              dx0_2, da_2, param_grads, first_layer_state_only, gradmax, td2))
        return e;
    const Mlp &n1 = args.net[0].net, &n2 = args.net[1].net;
    if (n1.S != n2.S || n1.A != n2.A || n1.E != n2.E || n1.H != n2.H || n1.n_out != n2.n_out || n1.cat != n2.cat ||
        n1.hd != n2.hd)
        return RPO_ERR_ARG;
    const int ein = n1.cat ? 2 * n1.E : n1.E;
    const int grid_rows = (n + kRows - 1) / kRows;
    const int grid_w = bwd_weights_grid(n1, first_layer_state_only);
    const bool gemm_rows = gemm_path_ok(n1, n, x0_1, h1_1, 0) && x0_2 && h1_2 && gemm_fits(n1, 2);
    if (gemm_rows)
        if (int e = gemm_backward_rows(args.net, 2, (hipStream_t)stream)) return e;
    if (ein == 128 && n1.H == 256 && !gemm_rows) {                // (see rpo_mlp_backward: both networks or neither)
        SplitK k1{nullptr, nullptr, 0, 0}, k2{nullptr, nullptr, 0, 0};
        const bool own = !param_grads || grad1_host->splitk_scratch != grad2_host->splitk_scratch;
        if (param_grads) {
            k1 = splitk_plan(args.net[0], grad1_host->splitk_scratch, grad1_host->splitk_floats);
            k2 = splitk_plan(args.net[1], grad2_host->splitk_scratch, grad2_host->splitk_floats);
        }
        if (own && bwd_stream_applies_x(args.net[0], k1) && bwd_stream_applies_x(args.net[1], k2)) {
            if (int r = launch_bwd_stream_x(args.net[0], k1, (hipStream_t)stream)) return r;
            return launch_bwd_stream_x(args.net[1], k2, (hipStream_t)stream);
        }
    }
    if (param_grads && ein == 128 && n1.H == 256 && grad1_host->splitk_scratch != grad2_host->splitk_scratch) {   // (see rpo_mlp_backward)
        const SplitK k1 = splitk_plan(args.net[0], grad1_host->splitk_scratch, grad1_host->splitk_floats);
        const SplitK k2 = splitk_plan(args.net[1], grad2_host->splitk_scratch, grad2_host->splitk_floats);
        // (both networks or neither: nothing is launched before both are known to qualify)
        if (k1.Z > 0 && k2.Z > 0 && onepass_applies<128, 256>(args.net[0], k1) && onepass_applies<128, 256>(args.net[1], k2)) {
            if (int r = launch_onepass<128, 256>(args.net[0], k1, (hipStream_t)stream)) return r;
            return launch_onepass<128, 256>(args.net[1], k2, (hipStream_t)stream);
        }
    }
#define RPO_MLP_BWD2(EIN_, H_)                                                                                         \
    if (ein == EIN_ && n1.H == H_) {                                                                                   \
        if (!gemm_rows)                                                                                                \
            hipLaunchKernelGGL((mlp_bwd_rows_kernel2<EIN_, H_>), dim3(grid_rows, 2), dim3(kThreads), 0, (hipStream_t)stream, \
                               args);                                                                                  \
        RPO_LAUNCH_CHECK();                                                                                            \
        if (param_grads) {                                                                                             \
            /* large batches: one split-K weights pass per network (each half of the scratch buffer... its own) */     \
            const SplitK k1 = splitk_plan(args.net[0], grad1_host->splitk_scratch, grad1_host->splitk_floats);         \
            const SplitK k2 = splitk_plan(args.net[1], grad2_host->splitk_scratch, grad2_host->splitk_floats);         \
            if (k1.Z > 0 && k2.Z > 0 && grad1_host->splitk_scratch != grad2_host->splitk_scratch) {                    \
                if (int e = launch_weights_splitk<EIN_, H_>(args.net[0], k1, grid_w, (hipStream_t)stream)) return e;   \
                return launch_weights_splitk<EIN_, H_>(args.net[1], k2, grid_w, (hipStream_t)stream);                  \
            }                                                                                                          \
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
