// Device-side pieces of CartSafe-v0 shared by cartsafe.hip and the fused pipelines (fused.hip).
#pragma once
#include "common.h"

namespace rpo_cart_dev {

struct CartConsts {
    float C[2], C_p, C_o_inv, b;
    float G[12], d[6], G_r[6], d_r[6];
    int partial;
};

static inline int load_consts(CartConsts& c, const float* h, int partial) {
    if (h == nullptr) return RPO_ERR_NULL;
    if (partial != 0 && partial != 1) return RPO_ERR_ARG;
    int k = 0;
    c.C[0] = h[k++]; c.C[1] = h[k++];
    c.C_p = h[k++]; c.C_o_inv = h[k++]; c.b = h[k++];
    for (int i = 0; i < 12; ++i) c.G[i] = h[k++];
    for (int i = 0; i < 6; ++i) c.d[i] = h[k++];
    for (int i = 0; i < 6; ++i) c.G_r[i] = h[k++];
    for (int i = 0; i < 6; ++i) c.d_r[i] = h[k++];
    c.partial = partial;
    return 0;
}

// physical constants, cartpole.py:75-91
constexpr float kGravity = 9.8f, kMassPole = 0.1f, kTotalMass = 1.1f, kLength = 0.5f;
constexpr float kPoleMassLength = 0.05f, kTau = 0.02f, kMuC = 0.1f, kMuP = 0.01f;
constexpr float kCosD0 = 0.5000000000000001f, kCosD1 = 0.8660254037844387f;   // cos(pi/3), cos(-pi/6)
constexpr float kSinD0 = 0.8660254037844386f, kSinD1 = -0.5f;                 // sin(pi/3), sin(-pi/6)
constexpr float kThetaThreshold = 0.20943951023931953f, kXThreshold = 2.4f;   // 12 deg, cartpole.py:88-89
constexpr float kActMax = 10.0f, kResetLo = -0.05f, kResetSpan = 0.1f;

__device__ __forceinline__ void eq_ineq(const CartConsts& c, float a0, float a1, float& h, float (&g)[6]) {
    RPO_FP_STRICT
    h = c.b - (a0 * c.C[0] + a1 * c.C[1]);                       // eq_resid, cartpole.py:375-376
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = (a0 * c.G[2 * i] + a1 * c.G[2 * i + 1]) - c.d[i];   // ineq_resid :378-379
}

// One row of the Lagrangian term nu . relu(g(a)) (Dual.forward on ineq_dist, rpo_ddpg.py:312-319): its value, the
// distances (d/d nu) and d/d action.  Shared by rpo_cartsafe_lagrangian and the fused actor-update pipeline.
__device__ __forceinline__ void lagrangian_row(const CartConsts& c, float a0, float a1, const float (&nu)[6], float& loss,
                                               float (&dist)[6], float& g0, float& g1) {
    RPO_FP_STRICT
    float h, g[6];
    eq_ineq(c, a0, a1, h, g);
    loss = 0.0f; g0 = 0.0f; g1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        dist[j] = fmaxf(g[j], 0.0f);                              // ineq_dist, cartpole.py:385-387
        loss += nu[j] * dist[j];                                  // Dual.forward, dual.py:63-65
        if (g[j] > 0.0f) { g0 += nu[j] * c.G[2 * j]; g1 += nu[j] * c.G[2 * j + 1]; }
    }
}

// autograd through complete_partial (cartpole.py:369-373): d/d ap of the loss given d/d action
__device__ __forceinline__ float complete_bwd_row(const CartConsts& c, float g0, float g1) {
    RPO_FP_STRICT
    const float k = -(c.C_p * c.C_o_inv);
    return c.partial == 0 ? g0 + k * g1 : g1 + k * g0;
}

__device__ __forceinline__ void reset_state(float (&s)[6], uint64_t seed, uint32_t env_id, uint32_t episode) {
    RPO_FP_STRICT
    const rpo_u4 r0 = rpo_philox(seed, env_id, episode, RPO_STREAM_RESET);
    const rpo_u4 r1 = rpo_philox(seed, env_id, episode, RPO_STREAM_RESET + 0x100u);
    s[0] = kResetLo + rpo_u01(r0.x) * kResetSpan;
    s[1] = kResetLo + rpo_u01(r0.y) * kResetSpan;
    s[2] = kResetLo + rpo_u01(r0.z) * kResetSpan;
    s[3] = kResetLo + rpo_u01(r0.w) * kResetSpan;
    s[4] = kResetLo + rpo_u01(r1.x) * kResetSpan;
    s[5] = kResetLo + rpo_u01(r1.y) * kResetSpan;
}

__device__ __forceinline__ void load_state(const float* __restrict__ p, float (&s)[6]) {
    const float2* q = reinterpret_cast<const float2*>(p);
    const float2 a = q[0], b = q[1], c = q[2];
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y; s[4] = c.x; s[5] = c.y;
}
__device__ __forceinline__ void store_state(float* __restrict__ p, const float (&s)[6]) {
    float2* q = reinterpret_cast<float2*>(p);
    q[0] = make_float2(s[0], s[1]); q[1] = make_float2(s[2], s[3]); q[2] = make_float2(s[4], s[5]);
}

// -------------------------------------------------------------------------------------------------------- step
struct StepArgs {
    int n;
    float* state;
    const float* action;
    int* ep_len;
    float* ep_ret;
    unsigned* ep_count;
    float* rows;
    long long cap_steps;
    float* stats;
    int stats_cap;
    long long* ctrl;
    int max_episode_steps;
    int auto_reset;
    float viol_thresh;
    uint64_t seed;
    uint32_t env_id_base;
    int tiles_ok;        // state / rows pointers are 16-byte aligned: the coalesced tile path may be used
};

constexpr int kStepStats = 10;   // 0..7 sums (+terminated), 8..9 maxima -- see flush order below

// One lane's step: dynamics, violations, TimeLimit, statistics, auto-reset.  `row` receives the 6 float4 chunks of the
// transition row (ReplayBuffer.add, buffer.py:22-29), `ns` the state the lane continues from.
__device__ __forceinline__ void cart_lane(const StepArgs& p, const CartConsts& c, int i, const float (&s)[6], float2 a,
                                          const RpoEpisode& ep, float (&ns)[6], float4 (&row)[6], float (&st)[kStepStats]) {
    RPO_FP_STRICT
    // violations of the PRE-step state and UN-clipped action (cartpole.py:229)
    float h, g[6];
    eq_ineq(c, a.x, a.y, h, g);
    float max_ineq = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) { g[k] = fmaxf(g[k], 0.0f); max_ineq = fmaxf(max_ineq, g[k]); }
    const float max_eq = fabsf(h);

    // dynamics, cartpole.py:170-197
    const float f0 = fminf(fmaxf(a.x, -kActMax), kActMax), f1 = fminf(fmaxf(a.y, -kActMax), kActMax);
    const float force = f0 * kCosD0 + f1 * kCosD1;
    const float force_y = f0 * kSinD0 + f1 * kSinD1;
    const float x = s[0], x_dot = s[1], theta = s[3], theta_dot = s[4], thetaacc_prev = s[5];
    float sn, cs;
    sincosf(theta, &sn, &cs);
    const float td2 = theta_dot * theta_dot;
    const float n_c = force_y + kTotalMass * kGravity - kPoleMassLength * (thetaacc_prev * sn + td2 * cs);
    const float prod = n_c * x_dot;
    const float sign = (prod > 0.0f) ? 1.0f : ((prod < 0.0f) ? -1.0f : prod);   // np.sign (0 -> 0, nan -> nan)
    const float temp = (force + kPoleMassLength * td2 * (sn + kMuC * sign * cs)) / kTotalMass + kMuC * kGravity * sign;
    const float thetaacc = (kGravity * sn - cs * temp - kMuP * theta_dot / kPoleMassLength) /
                           (kLength * (4.0f / 3.0f - kMassPole * cs * (cs - kMuC * kGravity * sign) / kTotalMass));
    const float xacc = (force + kPoleMassLength * (td2 * sn - thetaacc * cs) - kMuC * n_c * sign) / kTotalMass;
    ns[0] = x + kTau * x_dot;
    ns[1] = x_dot + kTau * xacc;
    ns[2] = xacc;
    ns[3] = theta + kTau * theta_dot;
    ns[4] = theta_dot + kTau * thetaacc;
    ns[5] = thetaacc;
    const bool terminated = ns[0] < -kXThreshold || ns[0] > kXThreshold || ns[3] < -kThetaThreshold ||
                            ns[3] > kThetaThreshold;                                  // cartpole.py:208-213
    // (a NaN action is what the reference's assert stops on, cartpole.py:170-174; a non-finite next state is a diverged lane)
    rpo_flag_nonfinite(p.ctrl, a.x != a.x || a.y != a.y || !__builtin_isfinite(((ns[0] + ns[1]) + (ns[2] + ns[3])) + (ns[4] + ns[5])));
    const int len = ep.len + 1;
    const bool done = terminated || len >= p.max_episode_steps;                        // gym TimeLimit
    const float reward = 1.0f;                                                         // cartpole.py:215-221
    const float ret = ep.ret + reward;

    row[0] = make_float4(s[0], s[1], s[2], s[3]);
    row[1] = make_float4(s[4], s[5], a.x, a.y);
    row[2] = make_float4(ns[0], ns[1], ns[2], ns[3]);
    row[3] = make_float4(ns[4], ns[5], reward, done ? 1.0f : 0.0f);
    row[4] = make_float4(h, g[0], g[1], g[2]);
    row[5] = make_float4(g[3], g[4], g[5], 0.0f);

    st[0] += reward;
    st[4] += max_ineq;
    st[5] += max_eq;
    st[6] += (fmaxf(max_ineq, max_eq) > p.viol_thresh) ? 1.0f : 0.0f;
    st[8] = fmaxf(st[8], max_ineq);
    st[9] = fmaxf(st[9], max_eq);
    if (done) {
        st[1] += 1.0f;
        st[2] += ret;
        st[3] += (float)len;
        st[7] += terminated ? 1.0f : 0.0f;
    }
    if (done && p.auto_reset) {   // env.reset() after a done, rpo_ddpg.py:142
        const unsigned episode = ep.count + 1u;
        p.ep_count[i] = episode;
        reset_state(ns, p.seed, p.env_id_base + (uint32_t)i, episode);
        p.ep_len[i] = 0;
        p.ep_ret[i] = 0.0f;
    } else {
        p.ep_len[i] = len;
        p.ep_ret[i] = ret;
    }
}

// ------------------------------------------------------------------------------- explore + complete + project
struct ActArgs {
    int n;
    const float* ap_raw;
    const float* noise;
    float* action;
    int* iters;
    int noise_mode;
    float eps_start, eps_end, eps_decay, box_lo, box_hi;
    int max_steps;
    float corr_lr, corr_eps, corr_momentum;
    uint64_t seed;
    uint32_t env_id_base;
    const long long* ctrl;
    float* stats;
    int stats_cap;
};

__device__ __forceinline__ float reduced_grad(const CartConsts& c, float ap) {
    RPO_FP_STRICT
    // ineq_partial_grad, cartpole.py:396-403: sign-based (sub)gradient of the reduced inequalities
    float grad = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) grad += ((ap * c.G_r[i] - c.d_r[i]) > 0.0f) ? c.G_r[i] : 0.0f;
    return grad;
}

// Exploration (ddpg_pa.py:108-110 / model/utils.py:53-62) + complete_partial (cartpole.py:369-373) + grad_steps
// (rpo_ddpg.py:266-286, corr_mode 0, per-lane stop test) for one lane.  Returns the full action and the iteration count.
__device__ __forceinline__ float2 cart_explore_project(const ActArgs& p, const CartConsts& c, int i, float ap_in, float eps_t,
                                                       long long t, int& iters) {
    RPO_FP_STRICT
    float ap = (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : ap_in;
    if (p.noise_mode == RPO_NOISE_EXPLICIT) {
        ap = rpo_clamp(ap + eps_t * p.noise[i], p.box_lo, p.box_hi);                  // ddpg_pa.py:108-110
    } else if (p.noise_mode == RPO_NOISE_PHILOX) {
        const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
        ap = rpo_clamp(ap + eps_t * rpo_normal(r.x, r.y), p.box_lo, p.box_hi);
    } else if (p.noise_mode == RPO_NOISE_UNIFORM) {                                      // model/utils.py:53-62
        const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
        const float scale = (p.box_hi - p.box_lo) * 0.5f;
        ap = scale * (2.0f * rpo_u01(r.x) - 1.0f) + (p.box_lo + scale);
    } else if (p.noise_mode == RPO_NOISE_CLIP_ONLY) {
        ap = rpo_clamp(ap, p.box_lo, p.box_hi);
    }
    // complete_partial, cartpole.py:369-373
    float ao = (c.b - ap * c.C_p) * c.C_o_inv;
    // grad_steps, rpo_ddpg.py:266-286 (corr_mode 0), per-lane stop test
    float old_p = 0.0f, old_o = 0.0f;
    int k = 0;
    for (; k < p.max_steps; ++k) {
        const float a0 = c.partial == 0 ? ap : ao, a1 = c.partial == 0 ? ao : ap;
        float h, g[6];
        eq_ineq(c, a0, a1, h, g);
        float mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) mx = fmaxf(mx, g[j]);
        if (k > 0 && !(fabsf(h) > p.corr_eps || mx > p.corr_eps)) break;
        const float gp = reduced_grad(c, ap);
        const float go = -(gp * c.C_p) * c.C_o_inv;                                      // cartpole.py:407
        const float sp = p.corr_lr * gp + p.corr_momentum * old_p;
        const float so = p.corr_lr * go + p.corr_momentum * old_o;
        ap -= sp; ao -= so;
        old_p = sp; old_o = so;
    }
    iters = k;
    return c.partial == 0 ? make_float2(ap, ao) : make_float2(ao, ap);
}

}  // namespace rpo_cart_dev
