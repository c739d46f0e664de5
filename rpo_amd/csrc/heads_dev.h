// Policy-head arithmetic shared by the head kernels (mlp.hip) and the fused rollout pipelines (fused.hip).
#pragma once
#include "common.h"

namespace rpo_head_dev {

constexpr float kLogSigMin = -23.0f, kLogSigMax = -2.0f, kHalfLog2Pi = 0.9189385332046727f;   // model/policy.py:6-7

// Squashed-Gaussian head (GaussianSharedPolicy.forward, model/policy.py:53-66, + the clip of PDSAC_PA.take_action,
// agent/sac_pa.py:111) for one row: raw = (mean, log-std head output), e = the N(0,1) draw of rsample.
__device__ __forceinline__ float gauss_head_row(float raw_mean, float raw_ls, float e, float scale, float base, float lo,
                                                float hi, int deterministic, float* logp) {
    RPO_FP_STRICT
    const float ls = rpo_clamp(raw_ls - 3.0f, kLogSigMin, kLogSigMax);
    const float x = raw_mean + e * expf(ls);
    const float y = tanhf(x);
    if (logp) *logp = -0.5f * e * e - ls - kHalfLog2Pi - logf(scale * (1.0f - y * y) + 1e-6f);
    const float a = deterministic ? scale * tanhf(raw_mean) + base : scale * y + base;
    return rpo_clamp(a, lo, hi);                                  // (NaN-propagating like torch.clamp, agent/sac_pa.py:111)
}

// Backward of gauss_head_row w.r.t. (mean head, log-std head) given d loss / d ap and the coefficient of log pi in the
// loss (SAC actor loss: alpha / B); autograd of model/policy.py:53-66 with the clip of agent/sac_pa.py:111.
__device__ __forceinline__ float2 gauss_head_bwd_row(float raw_mean, float raw_ls, float e, float dap, float dlogp, float scale,
                                                     float base, float lo, float hi) {
    RPO_FP_STRICT
    const float lsr = raw_ls - 3.0f;
    const float ls = fminf(fmaxf(lsr, kLogSigMin), kLogSigMax);
    const float sd = expf(ls);
    const float y = tanhf(raw_mean + e * sd);
    const float omy = 1.0f - y * y;
    const float a = scale * y + base;
    const float g_ap = (a >= lo && a <= hi) ? dap * scale * omy : 0.0f;
    const float g_lp = dlogp * (2.0f * scale * y * omy) / (scale * omy + 1e-6f);
    const float gx = g_ap + g_lp;
    const float dls = gx * e * sd - dlogp;
    return make_float2(gx, (lsr >= kLogSigMin && lsr <= kLogSigMax) ? dls : 0.0f);
}

// Backward of ap = clip(scale * tanh(o) + base + eps_t * noise, lo, hi) w.r.t. o (model/policy.py:30-31,
// agent/ddpg_pa.py:108-110), expressed through ap_det = scale * tanh(o) + base; has_noise = 0: no noise, no clip.
__device__ __forceinline__ float tanh_box_bwd_row(float dap, float ap_det, float noise, int has_noise, float eps_t, float lo,
                                                  float hi, float scale, float base) {
    RPO_FP_STRICT
    const float x = has_noise ? ap_det + eps_t * noise : ap_det;
    const float y = (ap_det - base) / scale;
    const bool pass = !has_noise || (x >= lo && x <= hi);
    return pass ? dap * scale * (1.0f - y * y) : 0.0f;
}

// TD target and Huber loss of one sample (rpo_ddpg.py:331-335, rpo_sac.py:346-353): qn = Q_targ(s', a') for RPODDPG,
// min(Q1_targ, Q2_targ) - alpha log pi(a'|s') for RPOSAC (has_q2 / has_logp); returns dLoss/dQ = clamp(q - y, -1, 1) / n
// and the sample's share of the mean smooth-L1 loss.  Shared by rpo_td_huber and the prologue of the backward kernels.
__device__ __forceinline__ float td_next_value(float qn1, float qn2, int has_q2, float logp, int has_logp, float alpha) {
    RPO_FP_STRICT
    float qn = has_q2 ? fminf(qn1, qn2) : qn1;
    if (has_logp) qn = qn - alpha * logp;
    return qn;
}
__device__ __forceinline__ float td_target(float reward, float done, float gamma, float qn) {
    RPO_FP_STRICT
    return reward + gamma * (1.0f - done) * qn;
}
__device__ __forceinline__ float td_huber_row(float q, float y, float inv_n, float* hub) {
    RPO_FP_STRICT
    const float d = q - y, ad = fabsf(d);
    *hub = (ad < 1.0f ? 0.5f * d * d : ad - 0.5f) * inv_n;
    return fminf(fmaxf(d, -1.0f), 1.0f) * inv_n;
}

}  // namespace rpo_head_dev
