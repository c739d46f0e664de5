// SpringPendulum-v0 on MI355X: vectorised env step (+TimeLimit, violations, replay scatter, auto-reset),
// exploration + equation solver + GRG projection with a state-dependent equality, and the constraint pieces of the
// actor loss.  Reference semantics: rpo/env/classic_control/pendulum.py (cited per function in include/rpo_hip.h).
// Every function here is a pure function of (obs, action): the reference's cached per-call tensors
// (pendulum.py:284-296) are not replicated, and ineq_partial_grad is evaluated row-wise.
#include "common.h"

namespace {

// pendulum.py:15-29
constexpr float kMaxSpeed = 8.0f, kMaxTorque = 6.0f, kMaxSum = 32.0f, kDt = 0.05f, kG = 10.0f, kM = 0.5f;
constexpr float kK = 1.0f, kL0 = 1.0f, kMDt = 10.0f;   // m / dt
constexpr float kPi = 3.14159265358979323846f, kTwoPi = 6.28318530717958647692f;
constexpr float kThetaLim = 0.26179938779914943654f;   // pi / 12
// reset box, pendulum.py:131-132 (float32 arrays in the reference)
__device__ constexpr float kResetLo[4] = {-0.26179938779914943654f, -1.0f, 0.95f, -0.05f};
__device__ constexpr float kResetHi[4] = {0.26179938779914943654f, 1.0f, 1.05f, 0.05f};

struct Eq {   // set_eq, pendulum.py:264-288
    float C_p, C_o, C_o_inv, b;
};

__device__ __forceinline__ Eq set_eq(float cos_t, float sin_t, float thdot, float l, float ldot) {
    Eq e;
    e.C_p = sin_t;   // fx is the basic action (pendulum.py:48,275-278)
    e.C_o = cos_t;
    e.C_o_inv = 1.0f / cos_t;
    e.b = -kMDt * ldot - (l * kM * thdot * thdot - kK * (l - kL0) - kM * kG * cos_t);   // :287
    return e;
}

__device__ __forceinline__ void reset_internal(float (&s)[4], uint64_t seed, uint32_t env_id, uint32_t episode) {
    const rpo_u4 r = rpo_philox(seed, env_id, episode, RPO_STREAM_RESET);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = kResetLo[k] + rpo_u01(w[k]) * (kResetHi[k] - kResetLo[k]);
}

__device__ __forceinline__ void store_obs(float* __restrict__ o, float cs, float sn, float thdot, float l, float ldot) {
    o[0] = cs; o[1] = sn; o[2] = thdot; o[3] = l; o[4] = ldot;
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_reset_kernel(int n, float* __restrict__ internal,
                                                                   float* __restrict__ obs, int* __restrict__ ep_len,
                                                                   float* __restrict__ ep_ret,
                                                                   const unsigned* __restrict__ ep_count,
                                                                   uint64_t seed, uint32_t env_id_base) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        float s[4];
        reset_internal(s, seed, env_id_base + (uint32_t)i, ep_count ? ep_count[i] : 0u);
        reinterpret_cast<float4*>(internal)[i] = make_float4(s[0], s[1], s[2], s[3]);
        float sn, cs;
        sincosf(s[0], &sn, &cs);
        if (obs) store_obs(obs + (size_t)i * 5, cs, sn, s[1], s[2], s[3]);
        ep_len[i] = 0;
        ep_ret[i] = 0.0f;
    }
}

struct StepArgs {
    int n;
    float* internal;
    float* obs;
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
};

constexpr int kStepStats = 10;

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_step_kernel(StepArgs p) {
    __shared__ float red[(RPO_BLOCK / RPO_WAVE) * kStepStats];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const long long ring_base = p.rows ? (t % p.cap_steps) * (long long)p.n : 0;
    float st[kStepStats];
#pragma unroll
    for (int k = 0; k < kStepStats; ++k) st[k] = 0.0f;

    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        const float4 s = reinterpret_cast<const float4*>(p.internal)[i];
        const float th = s.x, thdot = s.y, l = s.z, ldot = s.w;
        const float2 a = reinterpret_cast<const float2*>(p.action)[i];
        float sn, cs;
        sincosf(th, &sn, &cs);

        // violations of the PRE-step observation and UN-clipped action (pendulum.py:128,313-321)
        const Eq e = set_eq(cs, sn, thdot, l, ldot);
        const float h = e.b - (a.x * e.C_p + a.y * e.C_o);                       // eq_resid :298-300
        const float gi = fmaxf(a.x * a.x + a.y * a.y - kMaxSum, 0.0f);           // ineq_dist :302-311
        const float max_eq = fabsf(h);

        // dynamics, pendulum.py:85-124
        const float fx = fminf(fmaxf(a.x, -kMaxTorque), kMaxTorque), fy = fminf(fmaxf(a.y, -kMaxTorque), kMaxTorque);
        const float fth = -fy * sn + fx * cs;
        const float fl = fy * cs + fx * sn;
        float an = fmodf(th + kPi, kTwoPi);                                      // angle_normalize :367-368
        if (an < 0.0f) an += kTwoPi;
        const float costs = fabsf(an - kPi);
        const float thacc = (fth - kM * (kG * sn + 2.0f * ldot * thdot)) / (l * kM);
        const float lacc = (fl - kM * kG * cs + kM * l * thdot * thdot - kK * (l - kL0)) / kM;
        float nthdot = thdot + thacc * kDt;
        const float nldot = ldot + lacc * kDt;
        const float nth = th + nthdot * kDt;                                     // semi-implicit in theta :119
        const float nl = l + ldot * kDt;                                         // explicit in l :120
        nthdot = fminf(fmaxf(nthdot, -kMaxSpeed), kMaxSpeed);
        const bool terminated = nl <= 0.5f || nl >= 1.5f || nth >= kThetaLim || nth <= -kThetaLim;   // :124
        const float reward = 1.0f / (100.0f * costs + 1.0f);
        const int len = p.ep_len[i] + 1;
        const bool done = terminated || len >= p.max_episode_steps;
        const float ret = p.ep_ret[i] + reward;
        float nsn, ncs;
        sincosf(nth, &nsn, &ncs);

        if (p.rows) {
            float4* row = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_PEND_ROW);
            row[0] = make_float4(cs, sn, thdot, l);
            row[1] = make_float4(ldot, a.x, a.y, ncs);
            row[2] = make_float4(nsn, nthdot, nl, nldot);
            row[3] = make_float4(reward, done ? 1.0f : 0.0f, h, gi);
        }
        st[0] += reward;
        st[4] += gi;
        st[5] += max_eq;
        st[6] += (fmaxf(gi, max_eq) > p.viol_thresh) ? 1.0f : 0.0f;
        st[8] = fmaxf(st[8], gi);
        st[9] = fmaxf(st[9], max_eq);
        if (done) {
            st[1] += 1.0f;
            st[2] += ret;
            st[3] += (float)len;
            st[7] += terminated ? 1.0f : 0.0f;
        }
        float ns[4] = {nth, nthdot, nl, nldot};
        if (done && p.auto_reset) {
            const unsigned ep = p.ep_count[i] + 1u;
            p.ep_count[i] = ep;
            reset_internal(ns, p.seed, p.env_id_base + (uint32_t)i, ep);
            sincosf(ns[0], &nsn, &ncs);
            p.ep_len[i] = 0;
            p.ep_ret[i] = 0.0f;
        } else {
            p.ep_len[i] = len;
            p.ep_ret[i] = ret;
        }
        reinterpret_cast<float4*>(p.internal)[i] = make_float4(ns[0], ns[1], ns[2], ns[3]);
        if (p.obs) store_obs(p.obs + (size_t)i * 5, ncs, nsn, ns[1], ns[2], ns[3]);
    }

    if (p.stats) {
        const int slot[kStepStats] = {RPO_STAT_REWARD_SUM, RPO_STAT_EPISODES, RPO_STAT_RETURN_SUM, RPO_STAT_LENGTH_SUM,
                                      RPO_STAT_MAX_INEQ_SUM, RPO_STAT_MAX_EQ_SUM, RPO_STAT_VIOL_COUNT, RPO_STAT_TERMINATED,
                                      RPO_STAT_MAX_INEQ_MAX, RPO_STAT_MAX_EQ_MAX};
        rpo_stats_flush<kStepStats>(st, 8, slot, rpo_stats_row(p.stats, p.stats_cap, t), red);
    }
    rpo_step_epilogue(p.ctrl, t, p.stats, p.stats_cap);
}

// ------------------------------------------------------------------------------- explore + complete + project
struct ActArgs {
    int n;
    const float* obs;
    int obs_stride;
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

// ineq_partial_grad for one row, pendulum.py:331-343 (B = 1 semantics)
__device__ __forceinline__ void ipg_row(const Eq& e, float ax, float ay, float& gx, float& gy) {
    const float Gx = 2.0f * ax, Gy = 2.0f * ay;                                  // set_ineq :296
    const float dgp = Gx - Gy * (e.C_o_inv * e.C_p);                             // :334-335
    const float bgp = kMaxSum - (e.b * e.C_o_inv) * Gy;                          // :336
    const float bm = ax * dgp - bgp;                                             // :337
    gx = (bm > 0.0f) ? dgp : 0.0f;                                               // :339
    gy = -(gx * e.C_p) * e.C_o_inv;                                              // :342
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_act_project_kernel(ActArgs p) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);
    float iters_sum = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        const float* o = p.obs + (size_t)i * p.obs_stride;
        const Eq e = set_eq(o[0], o[1], o[2], o[3], o[4]);
        float ax = (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : p.ap_raw[i];
        if (p.noise_mode == RPO_NOISE_EXPLICIT) {
            ax = fminf(fmaxf(ax + eps_t * p.noise[i], p.box_lo), p.box_hi);
        } else if (p.noise_mode == RPO_NOISE_PHILOX) {
            const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
            ax = fminf(fmaxf(ax + eps_t * rpo_normal(r.x, r.y), p.box_lo), p.box_hi);
        } else if (p.noise_mode == RPO_NOISE_UNIFORM) {
            const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
            const float scale = (p.box_hi - p.box_lo) * 0.5f;
            ax = scale * (2.0f * rpo_u01(r.x) - 1.0f) + (p.box_lo + scale);
        } else if (p.noise_mode == RPO_NOISE_CLIP_ONLY) {
            ax = fminf(fmaxf(ax, p.box_lo), p.box_hi);
        }
        float ay = (e.b - ax * e.C_p) * e.C_o_inv;                               // complete_partial :256-262
        float old_x = 0.0f, old_y = 0.0f;
        int k = 0;
        for (; k < p.max_steps; ++k) {                                           // grad_steps, rpo_ddpg.py:266-286
            const float h = e.b - (ax * e.C_p + ay * e.C_o);
            const float g = ax * ax + ay * ay - kMaxSum;
            if (k > 0 && !(fabsf(h) > p.corr_eps || g > p.corr_eps)) break;
            float gx, gy;
            ipg_row(e, ax, ay, gx, gy);
            const float sx = p.corr_lr * gx + p.corr_momentum * old_x;
            const float sy = p.corr_lr * gy + p.corr_momentum * old_y;
            ax -= sx; ay -= sy;
            old_x = sx; old_y = sy;
        }
        reinterpret_cast<float2*>(p.action)[i] = make_float2(ax, ay);
        if (p.iters) p.iters[i] = k;
        iters_sum += (float)k;
    }
    if (p.stats) {
        const float r = rpo_wave_sum(iters_sum);
        if ((threadIdx.x & (RPO_WAVE - 1)) == 0) red[threadIdx.x / RPO_WAVE] = r;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.0f;
            for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) s += red[w];
            if (s != 0.0f) atomicAdd(rpo_stats_row(p.stats, p.stats_cap, t) + RPO_STAT_PROJ_ITERS, s);
        }
    }
}

// ------------------------------------------------------------------ the reference's literal BATCHED projection
// RPODDPG.grad_steps on a batch (rpo_ddpg.py:266-286) calls SpringPendulumEnv.ineq_partial_grad with B > 1, where
// `action[:, p] @ diff_grad_partial.T` is [B,1] @ [1,B] (pendulum.py:337): sample i's step sums over EVERY sample j,
//   grad_i = sum_j 1[a_x,i * dgp_j - bgp_i > 0] * dgp_j      (:337-339),
// and the loop's stop test is one max over the whole batch (rpo_ddpg.py:271-272).  This kernel reproduces exactly that
//  for the update step's target projection; one workgroup owns the batch, dgp lives in LDS.
__global__ __launch_bounds__(1024) void pendulum_project_batchref_kernel(
    int n, const float* __restrict__ obs, int obs_stride, const float* __restrict__ ap, float* __restrict__ action,
    int* __restrict__ iters_out, int max_steps, float corr_lr, float corr_eps, float corr_momentum) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // dgp[n] | stop flag
    float* dgp_s = lds;
    int* flag = reinterpret_cast<int*>(lds + n);
    const int i = threadIdx.x;                                     // one sample per thread, n <= blockDim.x <= 1024
    const bool live = i < n;
    Eq e = {0.0f, 1.0f, 1.0f, 0.0f};
    float ax = 0.0f, ay = 0.0f, ox = 0.0f, oy = 0.0f;
    if (live) {
        const float* o = obs + (size_t)i * obs_stride;
        e = set_eq(o[0], o[1], o[2], o[3], o[4]);
        ax = ap[i];
        ay = (e.b - ax * e.C_p) * e.C_o_inv;                       // complete_partial :256-262
    }
    int k = 0;
    for (; k < max_steps; ++k) {
        if (threadIdx.x == 0) *flag = 0;
        __syncthreads();
        if (live) {
            const float h = e.b - (ax * e.C_p + ay * e.C_o);
            const float g = ax * ax + ay * ay - kMaxSum;
            if (fabsf(h) > corr_eps || g > corr_eps) atomicOr(flag, 1);
            dgp_s[i] = 2.0f * ax - 2.0f * ay * (e.C_o_inv * e.C_p);                  // :334-335
        }
        __syncthreads();
        if (k > 0 && *flag == 0) break;                            // batch-global stop test, rpo_ddpg.py:271-272
        if (live) {
            const float bgp = kMaxSum - (e.b * e.C_o_inv) * (2.0f * ay);             // :336
            float grad = 0.0f;
            for (int j = 0; j < n; ++j) {                          // [B,1] @ [1,B] coupling, :337-339
                const float d = dgp_s[j];
                grad += (ax * d - bgp > 0.0f) ? d : 0.0f;
            }
            const float gy = -(grad * e.C_p) * e.C_o_inv;                            // :342
            const float sx = corr_lr * grad + corr_momentum * ox;
            const float sy = corr_lr * gy + corr_momentum * oy;
            ax -= sx; ay -= sy;
            ox = sx; oy = sy;
        }
        __syncthreads();
    }
    if (live) reinterpret_cast<float2*>(action)[i] = make_float2(ax, ay);
    if (threadIdx.x == 0 && iters_out) *iters_out = k;
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_complete_bwd_kernel(int n, const float* __restrict__ obs,
                                                                          int obs_stride,
                                                                          const float* __restrict__ ga,
                                                                          float* __restrict__ gap) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float* o = obs + (size_t)i * obs_stride;
        const float2 g = reinterpret_cast<const float2*>(ga)[i];
        gap[i] = g.x - g.y * (o[1] * (1.0f / o[0]));     // d a_y / d a_x = -C_p * C_o_inv
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_resid_kernel(int n, const float* __restrict__ obs, int obs_stride,
                                                                   const float* __restrict__ action,
                                                                   float* __restrict__ eq, float* __restrict__ ineq) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        if (eq) {
            const float* o = obs + (size_t)i * obs_stride;
            const Eq e = set_eq(o[0], o[1], o[2], o[3], o[4]);
            eq[i] = e.b - (a.x * e.C_p + a.y * e.C_o);
        }
        if (ineq) ineq[i] = a.x * a.x + a.y * a.y - kMaxSum;
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_ipg_kernel(int n, const float* __restrict__ obs, int obs_stride,
                                                                 const float* __restrict__ action,
                                                                 float* __restrict__ step) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float* o = obs + (size_t)i * obs_stride;
        const Eq e = set_eq(o[0], o[1], o[2], o[3], o[4]);
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        float gx, gy;
        ipg_row(e, a.x, a.y, gx, gy);
        reinterpret_cast<float2*>(step)[i] = make_float2(gx, gy);
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_lagrangian_kernel(int n, const float* __restrict__ action,
                                                                        const float* __restrict__ nu, float scale,
                                                                        float* __restrict__ loss_out,
                                                                        float* __restrict__ grad_action,
                                                                        float* __restrict__ grad_nu) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const float nu0 = nu[0];
    float acc = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        const float g = a.x * a.x + a.y * a.y - kMaxSum;
        acc += fmaxf(g, 0.0f);
        if (grad_action) {
            const float k = (g > 0.0f) ? 2.0f * scale * nu0 : 0.0f;
            reinterpret_cast<float2*>(grad_action)[i] = make_float2(k * a.x, k * a.y);
        }
    }
    const float r = rpo_wave_sum(acc);
    if ((threadIdx.x & (RPO_WAVE - 1)) == 0) red[threadIdx.x / RPO_WAVE] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) s += red[w];
        s *= scale;
        if (loss_out) atomicAdd(loss_out, nu0 * s);
        if (grad_nu) atomicAdd(grad_nu, s);
    }
}

}  // namespace

extern "C" {

int rpo_pendulum_reset(int n_envs, float* internal, float* obs, int* ep_len, float* ep_ret, const unsigned* ep_count,
                       unsigned long long seed, unsigned env_id_base, void* stream) {
    if (n_envs <= 0) return RPO_ERR_ARG;
    if (!internal || !ep_len || !ep_ret) return RPO_ERR_NULL;
    hipLaunchKernelGGL(pendulum_reset_kernel, dim3(rpo_grid_for(n_envs)), dim3(RPO_BLOCK), 0, (hipStream_t)stream,
                       n_envs, internal, obs, ep_len, ep_ret, ep_count, (uint64_t)seed, (uint32_t)env_id_base);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_step(int n_envs, float* internal, float* obs, const float* action, int* ep_len, float* ep_ret,
                      unsigned* ep_count, float* rows, long long cap_steps, float* stats, int stats_cap,
                      long long* ctrl, int max_episode_steps, int auto_reset, float viol_thresh,
                      unsigned long long seed, unsigned env_id_base, void* stream) {
    if (n_envs <= 0 || max_episode_steps <= 0) return RPO_ERR_ARG;
    if (!internal || !action || !ep_len || !ep_ret || !ep_count) return RPO_ERR_NULL;
    if (rows && cap_steps <= 0) return RPO_ERR_ARG;
    if (stats && stats_cap <= 0) return RPO_ERR_ARG;
    StepArgs a{n_envs, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, stats_cap, ctrl,
               max_episode_steps, auto_reset, viol_thresh, (uint64_t)seed, (uint32_t)env_id_base};
    hipLaunchKernelGGL(pendulum_step_kernel, dim3(rpo_grid_for(n_envs)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_act_project(int n, const float* obs, int obs_stride, const float* ap_raw, const float* noise,
                             float* action, int* iters, int noise_mode, float eps_start, float eps_end,
                             float eps_decay, float box_lo, float box_hi, int max_steps, float corr_lr,
                             float corr_eps, float corr_momentum, unsigned long long seed, unsigned env_id_base,
                             const long long* ctrl, float* stats, int stats_cap, void* stream) {
    if (n <= 0 || max_steps < 0 || obs_stride < RPO_PEND_OBS_DIM || noise_mode < RPO_NOISE_NONE ||
        noise_mode > RPO_NOISE_CLIP_ONLY)
        return RPO_ERR_ARG;
    if (!obs || !action || (noise_mode != RPO_NOISE_UNIFORM && !ap_raw)) return RPO_ERR_NULL;
    if (noise_mode == RPO_NOISE_EXPLICIT && !noise) return RPO_ERR_NULL;
    if (stats && stats_cap <= 0) return RPO_ERR_ARG;
    ActArgs a{n, obs, obs_stride, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo,
              box_hi, max_steps, corr_lr, corr_eps, corr_momentum, (uint64_t)seed, (uint32_t)env_id_base, ctrl, stats,
              stats_cap};
    hipLaunchKernelGGL(pendulum_act_project_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_project_batchref(int n, const float* obs, int obs_stride, const float* ap, float* action,
                                  int* iters_out, int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                                  void* stream) {
    if (n <= 0 || n > 1024 || max_steps < 0 || obs_stride < RPO_PEND_OBS_DIM) return RPO_ERR_ARG;
    if (!obs || !ap || !action) return RPO_ERR_NULL;
    const size_t lds = ((size_t)n + 4) * sizeof(float);
    const int threads = (n + RPO_WAVE - 1) / RPO_WAVE * RPO_WAVE;
    hipLaunchKernelGGL(pendulum_project_batchref_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, n, obs,
                       obs_stride, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_complete_bwd(int n, const float* obs, int obs_stride, const float* grad_action, float* grad_ap,
                              void* stream) {
    if (n <= 0 || obs_stride < RPO_PEND_OBS_DIM) return RPO_ERR_ARG;
    if (!obs || !grad_action || !grad_ap) return RPO_ERR_NULL;
    hipLaunchKernelGGL(pendulum_complete_bwd_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       obs, obs_stride, grad_action, grad_ap);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_resid(int n, const float* obs, int obs_stride, const float* action, float* eq_resid,
                       float* ineq_resid, void* stream) {
    if (n <= 0 || obs_stride < RPO_PEND_OBS_DIM) return RPO_ERR_ARG;
    if (!action || (eq_resid && !obs)) return RPO_ERR_NULL;
    hipLaunchKernelGGL(pendulum_resid_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, obs,
                       obs_stride, action, eq_resid, ineq_resid);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_ineq_partial_grad(int n, const float* obs, int obs_stride, const float* action, float* step,
                                   void* stream) {
    if (n <= 0 || obs_stride < RPO_PEND_OBS_DIM) return RPO_ERR_ARG;
    if (!obs || !action || !step) return RPO_ERR_NULL;
    hipLaunchKernelGGL(pendulum_ipg_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, obs,
                       obs_stride, action, step);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_lagrangian(int n, const float* action, const float* nu, float scale, float* loss_out,
                            float* grad_action, float* grad_nu, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!action || !nu) return RPO_ERR_NULL;
    hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       action, nu, scale, loss_out, grad_action, grad_nu);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
