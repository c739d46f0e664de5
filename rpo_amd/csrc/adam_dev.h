// clip_grad_norm_(inf) + torch.optim.Adam (+ DualAdam clamp, + Polyak) for ONE element of a flat slice: the arithmetic of
// rpo_adam_step / rpo_adam_step_multi (train_ops.hip), shared with the update launches that step the gradients they have
// just written (nsplit.hip, "folded" optimiser step) -- one definition, the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace rpo_adam_dev {

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

struct AdamCoef {
    float step_size, bc2_sqrt, coef, omb1, omb2;
};

// Step-wide scalars.  gm: the inf-norm of the slice's gradient (ignored when clip_thres == 0).
__device__ __forceinline__ AdamCoef adam_coefs_at(const AdamArgs& p, float gm, int step, int cached_step, double bc1, double bc2s) {
    // torch.optim.Adam (single-tensor path): bias corrections in double like torch's Python floats.  The two double-
    // precision pow() calls are ~1 us of every thread's critical path, so the workgroup that finishes a step last leaves
    // the corrections of the NEXT step behind the arrival word (step_dev + 4: {1 - beta1^t, sqrt(1 - beta2^t)} as doubles,
    // 0.0 = not cached yet): same functions, same arguments, same bits -- computed once instead of 34 000 times.
    if (bc1 == 0.0 || cached_step != step) {                    // step_dev[1]: the step the cached corrections belong to
        bc1 = 1.0 - pow((double)p.beta1, (double)step);
        bc2s = sqrt(1.0 - pow((double)p.beta2, (double)step));
    }
    AdamCoef k;
    k.step_size = (float)((double)p.lr / bc1);
    k.bc2_sqrt = (float)bc2s;
    // clip_grad_norm_(..., norm_type=inf): clip_coef clamped to 1, always applied
    k.coef = p.clip_thres > 0.0f ? fminf(p.clip_thres / (gm + 1e-6f), 1.0f) : 1.0f;
    k.omb1 = 1.0f - p.beta1;
    k.omb2 = 1.0f - p.beta2;
    return k;
}

__device__ __forceinline__ AdamCoef adam_coefs(const AdamArgs& p, float gm) {
    const int step = p.prepared ? p.step_dev[0] : p.step_dev[0] + 1;
    const double* cache = reinterpret_cast<const double*>(p.step_dev + 4);
    return adam_coefs_at(p, gm, step, p.step_dev[1], cache[0], cache[1]);
}

// Element i with its operands already loaded (g0 gradient, w0 parameter, m0 / v0 moments, t0 / t20 Polyak targets).
__device__ __forceinline__ void adam_elem(const AdamArgs& p, const AdamCoef& k, long long i, float g0, float w0, float m0,
                                          float v0, float t0, float t20) {
    float g = g0 * k.coef;
    if (p.zero_grad) p.grad[i] = 0.0f;            // the gradient is consumed: the next backward accumulates from zero
    else if (p.clip_thres > 0.0f) p.grad[i] = g;  // clip_grad_norm_ scales the gradients in place
    if (p.maximize) g = -g;
    float w = w0;
    if (p.weight_decay != 0.0f) g += p.weight_decay * w;
    float m = m0, v = v0;
    m = m + (g - m) * k.omb1;                     // exp_avg.lerp_(grad, 1 - beta1)
    v = v * p.beta2 + k.omb2 * g * g;             // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / k.bc2_sqrt + p.eps;
    w = w - k.step_size * (m / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (p.clamp_min0) w = fmaxf(w, 0.0f);         // DualAdam, model/dual.py:41-43
    p.param[i] = w; p.m[i] = m; p.v[i] = v;
    if (p.target) p.target[i] = t0 * (1.0f - p.tau) + w * p.tau;   // soft_update, ddpg_pa.py:77-86
    if (p.target2 && i < p.n2) p.target2[i] = t20 * (1.0f - p.tau) + w * p.tau;
}

}  // namespace rpo_adam_dev
