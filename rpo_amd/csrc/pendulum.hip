// SpringPendulum-v0 on MI355X: vectorised env step (+TimeLimit, violations, replay scatter, auto-reset),
// exploration + equation solver + GRG projection with a state-dependent equality, and the constraint pieces of the
// actor loss.  Reference semantics: rpo/env/classic_control/pendulum.py (cited per function in include/rpo_hip.h).
// Every function here is a pure function of (obs, action): the reference's cached per-call tensors
// (pendulum.py:284-296) are not replicated, and ineq_partial_grad is evaluated row-wise.
#include "pendulum_dev.h"

namespace {

using namespace rpo_pend_dev;

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

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_step_kernel(StepArgs p) {
    __shared__ float red[(RPO_BLOCK / RPO_WAVE) * kStepStats];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const long long ring_base = p.rows ? (t % p.cap_steps) * (long long)p.n : 0;
    float st[kStepStats];
#pragma unroll
    for (int k = 0; k < kStepStats; ++k) st[k] = 0.0f;

    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        const float4 s = reinterpret_cast<const float4*>(p.internal)[i];
        const float2 a = reinterpret_cast<const float2*>(p.action)[i];
        float ns[4], ncs, nsn;
        float4 row[4];
        pend_lane(p, i, s, a, rpo_load_episode(p.ep_len, p.ep_ret, p.ep_count, i), ns, ncs, nsn, row, st);
        if (p.rows) {
            float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_PEND_RING);
#pragma unroll
            for (int q = 0; q < 4; ++q) gr[q] = row[q];
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
__global__ __launch_bounds__(RPO_BLOCK) void pendulum_act_project_kernel(ActArgs p) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);
    float iters_sum = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        int k;
        const float2 act = pend_explore_project(p, p.obs + (size_t)i * p.obs_stride, i,
                                                (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : p.ap_raw[i], eps_t, t, k);
        reinterpret_cast<float2*>(p.action)[i] = act;
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
template <int LPS>
__global__ __launch_bounds__(1024) void pendulum_project_batchref_kernel(
    int n, const float* __restrict__ obs, int obs_stride, const float* __restrict__ ap, float* __restrict__ action,
    int* __restrict__ iters_out, int max_steps, float corr_lr, float corr_eps, float corr_momentum) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // dgp[n] | stop flag
    const int i = threadIdx.x / LPS;                               // LPS lanes per sample, n * LPS <= blockDim.x <= 1024
    project_batchref_body<LPS>(n, i < n ? obs + (size_t)i * obs_stride : nullptr, i < n ? ap[i] : 0.0f, action, iters_out,
                               max_steps, corr_lr, corr_eps, corr_momentum, lds);
}

// n <= 256: the register-tiled form (project_batchref_wide), always a full 1024-thread workgroup
__global__ __launch_bounds__(1024) void pendulum_project_batchref_wide_kernel(
    int n, const float* __restrict__ obs, int obs_stride, const float* __restrict__ ap, float* __restrict__ action,
    int* __restrict__ iters_out, int max_steps, float corr_lr, float corr_eps, float corr_momentum) {
    __shared__ __attribute__((aligned(16))) float lds[kWideLds];
    project_batchref_wide(n, obs, obs_stride, (int)threadIdx.x < n ? ap[threadIdx.x] : 0.0f, action, iters_out, max_steps,
                          corr_lr, corr_eps, corr_momentum, lds);
}

__global__ __launch_bounds__(RPO_BLOCK) void pendulum_complete_bwd_kernel(int n, const float* __restrict__ obs,
                                                                          int obs_stride,
                                                                          const float* __restrict__ ga,
                                                                          float* __restrict__ gap) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float* o = obs + (size_t)i * obs_stride;
        const float2 g = reinterpret_cast<const float2*>(ga)[i];
        gap[i] = complete_bwd_row(o, g.x, g.y);
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
                                                                        float* __restrict__ grad_nu,
                                                                        float* __restrict__ partials_out) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const float nu0 = nu[0];
    float acc = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        float dist, g0, g1;
        lagrangian_row(a.x, a.y, nu0, scale, dist, g0, g1);
        acc += dist;
        if (grad_action) reinterpret_cast<float2*>(grad_action)[i] = make_float2(g0, g1);
    }
    const float r = rpo_wave_sum(acc);
    if ((threadIdx.x & (RPO_WAVE - 1)) == 0) red[threadIdx.x / RPO_WAVE] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) s += red[w];
        s *= scale;
        // (a launch that adds into loss_out / grad_nu is always ONE workgroup; wider batches leave per-workgroup sums in
        //  partials_out: see rpo_cartsafe_lagrangian)
        if (partials_out) { partials_out[(size_t)blockIdx.x * 8] = nu0 * s; partials_out[(size_t)blockIdx.x * 8 + 1] = s; }
        else {
            if (loss_out) atomicAdd(loss_out, nu0 * s);
            if (grad_nu) atomicAdd(grad_nu, s);
        }
    }
}

// (== lagrangian_reduce_kernel of cartsafe.hip: G partial vectors [G][8] summed in a fixed order by one workgroup)
__global__ __launch_bounds__(RPO_BLOCK) void pend_lagrangian_reduce_kernel(int G, int K, const float* __restrict__ partials,
                                                                           float* __restrict__ loss_out, float* __restrict__ grad_nu) {
    __shared__ float red[RPO_BLOCK];
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        for (int b = threadIdx.x; b < G; b += RPO_BLOCK) acc += partials[(size_t)b * 8 + k];
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int off = RPO_BLOCK / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float* dst = k == 0 ? loss_out : (grad_nu ? grad_nu + k - 1 : nullptr);
            if (dst) *dst += red[0];
        }
        __syncthreads();
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
    if (n <= 256) {                                                // four samples x 16 values per thread (project_batchref_wide)
        hipLaunchKernelGGL(pendulum_project_batchref_wide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, obs,
                           obs_stride, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum);
    } else {
        const int threads = (n + RPO_WAVE - 1) / RPO_WAVE * RPO_WAVE;
        hipLaunchKernelGGL(pendulum_project_batchref_kernel<1>, dim3(1), dim3(threads), lds, (hipStream_t)stream, n, obs,
                           obs_stride, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum);
    }
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
    if (n > RPO_BLOCK && (loss_out || grad_nu)) {               // (deterministic sums: see rpo_cartsafe_lagrangian)
        const int G = rpo_grid_for(n);
        if (grad_action && (long long)8 * G <= (long long)2 * n) {
            hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, (float*)nullptr, (float*)nullptr, grad_action);
            RPO_LAUNCH_CHECK();
            hipLaunchKernelGGL(pend_lagrangian_reduce_kernel, dim3(1), dim3(RPO_BLOCK), 0, (hipStream_t)stream, G, 2,
                               (const float*)grad_action, loss_out, grad_nu);
            RPO_LAUNCH_CHECK();
            hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, grad_action, (float*)nullptr, (float*)nullptr);
            RPO_LAUNCH_CHECK();
            return 0;
        }
        if (grad_action) {
            hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, grad_action, (float*)nullptr, (float*)nullptr);
            RPO_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(1), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                           loss_out, (float*)nullptr, grad_nu, (float*)nullptr);
        RPO_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(pendulum_lagrangian_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       action, nu, scale, loss_out, grad_action, grad_nu, (float*)nullptr);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
