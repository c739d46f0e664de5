#include <stdlib.h>
// Update-step kernels of RPODDPG.train / RPOSAC.train on MI355X: fused TD-target + Huber (forward + backward),
// inf-norm of a flat gradient buffer, fused clip + Adam (+DualAdam clamp, + Polyak), Polyak alone, Philox test hook.
// All are single-pass streaming kernels with wave64 shuffle reductions.
#include "common.h"
#include "heads_dev.h"

namespace {

// ------------------------------------------------------------------------------------------- TD target + Huber
__global__ __launch_bounds__(RPO_BLOCK) void td_huber_kernel(
    int n, const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ qn1,
    const float* __restrict__ qn2, const float* __restrict__ logp, float alpha, const float* __restrict__ reward,
    int reward_stride, const float* __restrict__ done, int done_stride, float gamma, float* __restrict__ loss_out,
    float* __restrict__ grad_q1, float* __restrict__ grad_q2, float* __restrict__ target_out) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const float inv_n = 1.0f / (float)n;
    float acc = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float qn = rpo_head_dev::td_next_value(qn1[i], qn2 ? qn2[i] : 0.0f, qn2 != nullptr,           // rpo_sac.py:346-347
                                                     logp ? logp[i] : 0.0f, logp != nullptr, alpha);
        const float y = rpo_head_dev::td_target(reward[(size_t)i * reward_stride], done[(size_t)i * done_stride], gamma,
                                                qn);                                                            // rpo_ddpg.py:332
        if (target_out) target_out[i] = y;
        float h1 = 0.0f, h2 = 0.0f;
        const float g1 = rpo_head_dev::td_huber_row(q1[i], y, inv_n, &h1);
        if (grad_q1) grad_q1[i] = g1;
        if (q2) {
            const float g2 = rpo_head_dev::td_huber_row(q2[i], y, inv_n, &h2);
            if (grad_q2) grad_q2[i] = g2;
        }
        acc += h1 + h2;
    }
    const float r = rpo_wave_sum(acc);
    if ((threadIdx.x & (RPO_WAVE - 1)) == 0) red[threadIdx.x / RPO_WAVE] = r;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out) {
        float s = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) s += red[w];
        atomicAdd(loss_out, s);
    }
}

// ------------------------------------------------------------------------------------------------- inf-norm
__global__ __launch_bounds__(RPO_BLOCK) void absmax_kernel(long long n, const float* __restrict__ x,
                                                           float* __restrict__ max_out, int slotted) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    float m = 0.0f;
    const long long n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (long long i = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; i < n4; i += (long long)gridDim.x * RPO_BLOCK) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RPO_BLOCK)
        m = fmaxf(m, fabsf(x[i]));
    m = rpo_wave_max(m);
    if ((threadIdx.x & (RPO_WAVE - 1)) == 0) red[threadIdx.x / RPO_WAVE] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RPO_BLOCK / RPO_WAVE; ++w) m = fmaxf(m, red[w]);
        m = fmaxf(m, red[0]);
        // slotted: a gradmax buffer (RPO_GRADMAX_LEN) -- the workgroups spread over its 16 slots instead of queueing on one word
        float* dst = slotted ? max_out + (blockIdx.x % RPO_GRADMAX_SLOTS) * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS) : max_out;
        if (m > 0.0f) rpo_atomic_max_nonneg(dst, m);
    }
}

// -------------------------------------------------------------------------------------- clip + Adam + Polyak
struct AdamArgs {
    long long n;
    float* param;
    float* grad;
    float* m;
    float* v;
    int* step_dev;
    float lr, beta1, beta2, eps, weight_decay;
    int maximize;
    float clip_thres;
    float* gradmax;
    int reset_gradmax;
    int zero_grad;
    int clamp_min0;
    float* target;
    float tau;
    long long* arrive;   // scratch word for the "last workgroup" epilogue (step counter / gradmax reset)
    float* target2;      // second Polyak target for elements [0, n2)
    long long n2;
    int polyak_only;
    long long* clock;    // NULL, or a device counter advanced by one when the launch has finished (update clock)
    int prepared;        // 1: the launch before this one advanced step_dev[0] / cached the corrections (rpo_adam_prepare
                         //    semantics, see rpo_hip.h): no bookkeeping here, hence no last-workgroup detection at all
};

__device__ __forceinline__ void adam_body(const AdamArgs& p) {
    if (p.polyak_only) {
        for (long long i = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += (long long)gridDim.x * RPO_BLOCK)
            p.target[i] = p.target[i] * (1.0f - p.tau) + p.param[i] * p.tau;
        return;
    }
    // The element's operands are requested FIRST: the step counter / cached corrections / gradmax slots below are each a
    // memory round trip of their own (written by the launch before, usually by another XCD), and with the loads behind them
    // the launch was five dependent round trips long (4.8 us in the chain for 138 KB); now they all fly together.
    const long long i0 = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x, stride = (long long)gridDim.x * RPO_BLOCK;
    float g0 = 0.0f, w0 = 0.0f, m0 = 0.0f, v0 = 0.0f, t0 = 0.0f, t20 = 0.0f;
    if (i0 < p.n) {
        g0 = p.grad[i0]; w0 = p.param[i0]; m0 = p.m[i0]; v0 = p.v[i0];
        if (p.target) t0 = p.target[i0];
        if (p.target2 && i0 < p.n2) t20 = p.target2[i0];
    }
    asm volatile("" ::: "memory");                          // keeps the requests above this line (the compiler sinks them otherwise)
    float gm = 0.0f;                                        // the norm is the maximum over the slots (RPO_GRADMAX_SLOTS)
    if (p.clip_thres > 0.0f) {
#pragma unroll
        for (int j = 0; j < RPO_GRADMAX_SLOTS; ++j) gm = fmaxf(gm, p.gradmax[j * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS)]);
    }
    // torch.optim.Adam (single-tensor path): bias corrections in double like torch's Python floats.  The two double-
    // precision pow() calls are ~1 us of every thread's critical path, so the workgroup that finishes a step last leaves
    // the corrections of the NEXT step behind the arrival word (step_dev + 4: {1 - beta1^t, sqrt(1 - beta2^t)} as doubles,
    // 0.0 = not cached yet): same functions, same arguments, same bits -- computed once instead of 34 000 times.
    const int step = p.prepared ? p.step_dev[0] : p.step_dev[0] + 1;
    double* cache = reinterpret_cast<double*>(p.step_dev + 4);
    double bc1 = cache[0], bc2s = cache[1];
    if (bc1 == 0.0 || p.step_dev[1] != step) {                  // step_dev[1]: the step the cached corrections belong to
        bc1 = 1.0 - pow((double)p.beta1, (double)step);
        bc2s = sqrt(1.0 - pow((double)p.beta2, (double)step));
    }
    const float step_size = (float)((double)p.lr / bc1);
    const float bc2_sqrt = (float)bc2s;
    // clip_grad_norm_(..., norm_type=inf): clip_coef clamped to 1, always applied
    const float coef = p.clip_thres > 0.0f ? fminf(p.clip_thres / (gm + 1e-6f), 1.0f) : 1.0f;
    const float omb1 = 1.0f - p.beta1, omb2 = 1.0f - p.beta2;
    for (long long i = i0; i < p.n; i += stride) {
        if (i != i0) {
            g0 = p.grad[i]; w0 = p.param[i]; m0 = p.m[i]; v0 = p.v[i];
            if (p.target) t0 = p.target[i];
            if (p.target2 && i < p.n2) t20 = p.target2[i];
        }
        float g = g0 * coef;
        if (p.zero_grad) p.grad[i] = 0.0f;            // the gradient is consumed: the next backward accumulates from zero
        else if (p.clip_thres > 0.0f) p.grad[i] = g;  // clip_grad_norm_ scales the gradients in place
        if (p.maximize) g = -g;
        float w = w0;
        if (p.weight_decay != 0.0f) g += p.weight_decay * w;
        float m = m0, v = v0;
        m = m + (g - m) * omb1;                       // exp_avg.lerp_(grad, 1 - beta1)
        v = v * p.beta2 + omb2 * g * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(v) / bc2_sqrt + p.eps;
        w = w - step_size * (m / denom);              // param.addcdiv_(exp_avg, denom, value=-step_size)
        if (p.clamp_min0) w = fmaxf(w, 0.0f);         // DualAdam, model/dual.py:41-43
        p.param[i] = w; p.m[i] = m; p.v[i] = v;
        if (p.target) p.target[i] = t0 * (1.0f - p.tau) + w * p.tau;   // soft_update, ddpg_pa.py:77-86
        if (p.target2 && i < p.n2) p.target2[i] = t20 * (1.0f - p.tau) + w * p.tau;
    }
    if (p.prepared) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        // last workgroup of the launch: two-level arrival (sub-counter b % 16, then the top word) once the grid is wide
        // enough for same-word atomics to queue up; every counter is back at 0 when the launch ends
        const unsigned n = gridDim.x, sub = blockIdx.x % 16u;
        unsigned long long* top = reinterpret_cast<unsigned long long*>(p.arrive);
        bool last;
        if (n <= 16u) {
            last = atomicAdd(top, 1ull) == (unsigned long long)n - 1ull;
        } else {
            unsigned long long* sc = reinterpret_cast<unsigned long long*>(p.step_dev + 32 + 32 * sub);
            const unsigned long long in_sub = (n - sub + 15u) / 16u;
            last = false;
            if (atomicAdd(sc, 1ull) == in_sub - 1ull) {
                *sc = 0;
                last = atomicAdd(top, 1ull) == 15ull;
            }
        }
        if (last) {
            *top = 0;
            p.step_dev[0] = step;
            cache[0] = 1.0 - pow((double)p.beta1, (double)(step + 1));
            cache[1] = sqrt(1.0 - pow((double)p.beta2, (double)(step + 1)));
            p.step_dev[1] = step + 1;
            if (p.reset_gradmax && p.gradmax)
                for (int j = 0; j < RPO_GRADMAX_SLOTS; ++j) p.gradmax[j * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS)] = 0.0f;
            if (p.clock) p.clock[0] += 1;
        }
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void adam_kernel(AdamArgs p) { adam_body(p); }

struct AdamArgs4 {
    AdamArgs seg[4];
};
__global__ __launch_bounds__(RPO_BLOCK) void adam_multi_kernel(AdamArgs4 p) { adam_body(p.seg[blockIdx.y]); }

__global__ __launch_bounds__(RPO_BLOCK) void min_q_bwd_kernel(int n, const float* __restrict__ q1, const float* __restrict__ q2,
                                                              float scale, float* __restrict__ dq1, float* __restrict__ dq2) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float a = q1[i], b = q2[i];
        const float w = (a < b ? 1.0f : 0.0f) + 0.5f * (a == b ? 1.0f : 0.0f);
        dq1[i] = w * scale;
        dq2[i] = (1.0f - w) * scale;
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void polyak_kernel(long long n, const float* __restrict__ param,
                                                           float* __restrict__ target, float tau) {
    for (long long i = (long long)blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * RPO_BLOCK)
        target[i] = target[i] * (1.0f - tau) + param[i] * tau;
}

__global__ __launch_bounds__(RPO_BLOCK) void philox_fill_kernel(int n, uint32_t* __restrict__ out, uint64_t seed,
                                                                uint32_t id_base, uint32_t index, uint32_t tag) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const rpo_u4 r = rpo_philox(seed, id_base + (uint32_t)i, index, tag);
        reinterpret_cast<uint4*>(out)[i] = make_uint4(r.x, r.y, r.z, r.w);
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void philox_normal_kernel(int n, float* __restrict__ out, uint64_t seed,
                                                                  uint32_t id_base, uint32_t salt, uint32_t tag,
                                                                  const long long* __restrict__ ctrl) {
    const uint32_t t = ctrl ? (uint32_t)ctrl[RPO_CTRL_T] : 0u;
    const uint32_t sub = ctrl ? (uint32_t)ctrl[RPO_CTRL_UPDATES] : 0u;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const rpo_u4 r = rpo_philox(seed, id_base + (uint32_t)i, t + salt, tag, sub);
        out[i] = rpo_normal(r.x, r.y);
    }
}

}  // namespace

// defaults of the kernel-variant switches (RPO_TUNE_* order)
int g_rpo_tune[RPO_TUNE_COUNT] = {/* FWD_STREAM */ 1, /* FWD_STREAM_WAVES */ 16, /* BWD_ONEPASS */ 1, /* GEMM_KSPLIT */ 1,
                                  /* MLP_GEMM */ 1, /* ROLLOUT_WIDE: 2 = by size */ 2, /* BWD_STREAM */ 1,
                                  /* L1_MFMA */ 1, /* EVOPF_PLACE */ 0};

extern "C" {

int rpo_philox_normal(int n, float* out, unsigned long long seed, unsigned id_base, unsigned salt,
                      unsigned stream_tag, const long long* ctrl, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(philox_normal_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, out,
                       (uint64_t)seed, (uint32_t)id_base, (uint32_t)salt, (uint32_t)stream_tag, ctrl);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_abi_version(void) { return RPO_ABI_VERSION; }

int rpo_tuning(int key, int value) {
    if (key < 0 || key >= RPO_TUNE_COUNT) return RPO_ERR_ARG;
    const int old = g_rpo_tune[key];
    if (value >= 0) g_rpo_tune[key] = value;
    return old;
}

int rpo_philox_fill(int n, unsigned* out, unsigned long long seed, unsigned id_base, unsigned index,
                    unsigned stream_tag, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(philox_fill_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, out,
                       (uint64_t)seed, (uint32_t)id_base, (uint32_t)index, (uint32_t)stream_tag);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_td_huber(int n, const float* q1, const float* q2, const float* qn1, const float* qn2, const float* logp,
                 float alpha, const float* reward, int reward_stride, const float* done, int done_stride, float gamma,
                 float* loss_out, float* grad_q1, float* grad_q2, float* target_out, void* stream) {
    if (n <= 0 || reward_stride <= 0 || done_stride <= 0) return RPO_ERR_ARG;
    if (!q1 || !qn1 || !reward || !done) return RPO_ERR_NULL;
    hipLaunchKernelGGL(td_huber_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, q1, q2, qn1,
                       qn2, logp, alpha, reward, reward_stride, done, done_stride, gamma, loss_out, grad_q1, grad_q2,
                       target_out);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_absmax(long long n, const float* x, float* max_out, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!x || !max_out) return RPO_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return RPO_ERR_ARG;
    hipLaunchKernelGGL(absmax_kernel, dim3(rpo_grid_for(n / 4 + 1)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, x,
                       max_out, 0);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_absmax_slots(long long n, const float* x, float* gradmax, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!x || !gradmax) return RPO_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return RPO_ERR_ARG;
    hipLaunchKernelGGL(absmax_kernel, dim3(rpo_grid_for(n / 4 + 1)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, x,
                       gradmax, 1);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_adam_step(long long n, float* param, float* grad, float* exp_avg, float* exp_avg_sq, int* step_dev,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int maximize, float clip_thres,
                  float* gradmax, int reset_gradmax, int zero_grad, int clamp_min0, float* target, float tau,
                  long long* clock, int prepared, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !step_dev) return RPO_ERR_NULL;
    if (clip_thres > 0.0f && !gradmax) return RPO_ERR_NULL;
    // the arrival word lives right behind the step counter: step_dev must point at int32[RPO_ADAM_STATE_LEN] = {step, pad,
    // arrive (8 B), cached bias corrections of the next step (2 doubles), ..., 16 sub-counters from word 32 on}
    AdamArgs a{n, param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1, beta2, eps, weight_decay, maximize,
               clip_thres, gradmax, reset_gradmax, zero_grad, clamp_min0, target, tau,
               reinterpret_cast<long long*>(step_dev + 2), nullptr, 0, 0, clock, prepared ? 1 : 0};
    hipLaunchKernelGGL(adam_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_adam_step_multi(int count, const rpo_adam_seg* segs, long long* clock, void* stream) {
    if (count < 1 || count > 4) return RPO_ERR_ARG;
    if (!segs) return RPO_ERR_NULL;
    AdamArgs4 a;
    long long n_max = 0;
    for (int k = 0; k < count; ++k) {
        const rpo_adam_seg& g = segs[k];
        if (g.n <= 0 || g.n2 < 0 || g.n2 > g.n) return RPO_ERR_ARG;
        if (!g.param) return RPO_ERR_NULL;
        if (g.polyak_only) {
            if (!g.target || (k == 0 && clock)) return RPO_ERR_NULL;     // the clock rides on slice 0's arrival counter
        } else {
            if (!g.grad || !g.exp_avg || !g.exp_avg_sq || !g.step_dev) return RPO_ERR_NULL;
            if (g.clip_thres > 0.0f && !g.gradmax) return RPO_ERR_NULL;
        }
        a.seg[k] = AdamArgs{g.n, g.param, g.grad, g.exp_avg, g.exp_avg_sq, g.step_dev, g.lr, g.beta1, g.beta2, g.eps,
                            g.weight_decay, g.maximize, g.clip_thres, g.gradmax, g.reset_gradmax, g.zero_grad, g.clamp_min0,
                            g.target, g.tau, g.polyak_only ? nullptr : reinterpret_cast<long long*>(g.step_dev + 2),
                            g.target2, g.n2, g.polyak_only, (k == 0 && !g.polyak_only) ? clock : nullptr, g.prepared ? 1 : 0};
        n_max = g.n > n_max ? g.n : n_max;
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3(rpo_grid_for(n_max), count), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_min_q_bwd(int n, const float* q1, const float* q2, float scale, float* dq1, float* dq2, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!q1 || !q2 || !dq1 || !dq2) return RPO_ERR_NULL;
    hipLaunchKernelGGL(min_q_bwd_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, q1, q2, scale, dq1, dq2);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_polyak(long long n, const float* param, float* target, float tau, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!param || !target) return RPO_ERR_NULL;
    hipLaunchKernelGGL(polyak_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, param, target,
                       tau);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
