// CartSafe-v0 on MI355X: vectorised env step (+TimeLimit, violations, replay scatter, auto-reset), exploration +
// equation solver + GRG projection, and the constraint pieces of the actor loss.  One lane per env / sample.
// Reference semantics: rpo/env/classic_control/cartpole.py, rpo/algo/rpo_ddpg.py (cited per function in
// include/rpo_hip.h).  All kernels are HBM-bound streaming kernels (O(100) flops per ~150 B).
#include "cartsafe_dev.h"

namespace {

using namespace rpo_cart_dev;

// ------------------------------------------------------------------------------------------------------- reset
__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_reset_kernel(int n, float* __restrict__ state,
                                                                   int* __restrict__ ep_len,
                                                                   float* __restrict__ ep_ret,
                                                                   const unsigned* __restrict__ ep_count,
                                                                   uint64_t seed, uint32_t env_id_base) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        float s[6];
        reset_state(s, seed, env_id_base + (uint32_t)i, ep_count ? ep_count[i] : 0u);
        store_state(state + (size_t)i * 6, s);
        ep_len[i] = 0;
        ep_ret[i] = 0.0f;
    }
}

// Tiles of 256 consecutive lanes: the 6 KB state tile and the 24 KB block of transition rows are contiguous in HBM, so
// they move as fully coalesced float4 streams (1 KiB per wave instruction) and are transposed to / from the per-lane
// view through LDS (row stride 7 float4 = 112 B: conflict-free ds_write_b128).  A partial last tile takes the direct
// per-lane path.
__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_step_kernel(StepArgs p, CartConsts c) {
    __shared__ float red[(RPO_BLOCK / RPO_WAVE) * kStepStats];
    __shared__ __attribute__((aligned(16))) float4 rows_s[RPO_BLOCK * 7];
    __shared__ __attribute__((aligned(16))) float state_s[RPO_BLOCK * 6];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const long long ring_base = p.rows ? (t % p.cap_steps) * (long long)p.n : 0;
    // per-thread statistics: reward, episodes, return, length, max_ineq, max_eq, viol_count, terminated | maxima
    float st[kStepStats];
#pragma unroll
    for (int k = 0; k < kStepStats; ++k) st[k] = 0.0f;
    const int tid = threadIdx.x;

    for (int base = blockIdx.x * RPO_BLOCK; base < p.n; base += gridDim.x * RPO_BLOCK) {
        const int i = base + tid;
        float s[6], ns[6];
        float4 row[6];
        if (p.tiles_ok && base + RPO_BLOCK <= p.n) {
            // ---- full tile: coalesced state load -> LDS -> per-lane view
            const float4* gs = reinterpret_cast<const float4*>(p.state + (size_t)base * 6);
            float4* ls = reinterpret_cast<float4*>(state_s);
            ls[tid] = gs[tid];
            if (tid < RPO_BLOCK / 2) ls[RPO_BLOCK + tid] = gs[RPO_BLOCK + tid];
            const float2 a = reinterpret_cast<const float2*>(p.action)[i];
            __syncthreads();
            {
                const float2* q = reinterpret_cast<const float2*>(state_s + tid * 6);
                const float2 u = q[0], v = q[1], w = q[2];
                s[0] = u.x; s[1] = u.y; s[2] = v.x; s[3] = v.y; s[4] = w.x; s[5] = w.y;
            }
            cart_lane(p, c, i, s, a, rpo_load_episode(p.ep_len, p.ep_ret, p.ep_count, i), ns, row, st);
            __syncthreads();                                   // every lane has read its state: the tile can be reused
            {
                float2* q = reinterpret_cast<float2*>(state_s + tid * 6);
                q[0] = make_float2(ns[0], ns[1]); q[1] = make_float2(ns[2], ns[3]); q[2] = make_float2(ns[4], ns[5]);
            }
            if (p.rows) {
#pragma unroll
                for (int k = 0; k < 6; ++k) rows_s[tid * 7 + k] = row[k];
            }
            __syncthreads();
            float4* gs_out = reinterpret_cast<float4*>(p.state + (size_t)base * 6);
            gs_out[tid] = ls[tid];
            if (tid < RPO_BLOCK / 2) gs_out[RPO_BLOCK + tid] = ls[RPO_BLOCK + tid];
            if (p.rows) {
                float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + base) * RPO_CART_RING);
                // whole ring rows, padding included (zeros): full 128-byte lines, fully coalesced.  Writing only the 96 bytes of the
                // transition into 128-byte-spaced rows (partial lines) made this kernel 33.9 -> 66.7 us at 1M lanes.
                constexpr int RC = RPO_CART_RING / 4;
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    const int ch = k * RPO_BLOCK + tid;         // float4 chunk of the tile's 256 ring rows
                    const int r = ch / RC, part = ch - r * RC;
                    const float4 v4 = part < 6 ? rows_s[r * 7 + part] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (p.n >= 65536) {                         // (streaming regime: the ring is written once and sampled much later)
                        typedef float v4f __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(v4f{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<v4f*>(&gr[ch]));
                    } else {
                        gr[ch] = v4;
                    }
                }
            }
            __syncthreads();                                   // LDS tiles are rewritten by the next iteration
        } else if (i < p.n) {
            // ---- partial last tile: direct per-lane accesses
            load_state(p.state + (size_t)i * 6, s);
            const float2 a = reinterpret_cast<const float2*>(p.action)[i];
            cart_lane(p, c, i, s, a, rpo_load_episode(p.ep_len, p.ep_ret, p.ep_count, i), ns, row, st);
            if (p.rows) {
                float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_CART_RING);
#pragma unroll
                for (int k = 0; k < 6; ++k) gr[k] = row[k];
            }
            store_state(p.state + (size_t)i * 6, ns);
        }
    }

    if (p.stats) {
        const int slot[kStepStats] = {RPO_STAT_REWARD_SUM, RPO_STAT_EPISODES, RPO_STAT_RETURN_SUM, RPO_STAT_LENGTH_SUM,
                                      RPO_STAT_MAX_INEQ_SUM, RPO_STAT_MAX_EQ_SUM, RPO_STAT_VIOL_COUNT, RPO_STAT_TERMINATED,
                                      RPO_STAT_MAX_INEQ_MAX, RPO_STAT_MAX_EQ_MAX};
        rpo_stats_flush<kStepStats>(st, 8, slot, rpo_stats_row(p.stats, p.stats_cap, t), red);
    }
    rpo_step_epilogue(p.ctrl, t, p.stats, p.stats_cap);
}

__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_act_project_kernel(ActArgs p, CartConsts c) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);   // eps_decay, ddpg_pa.py:118-119
    float iters_sum = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        int k;
        const float2 act = cart_explore_project(p, c, i, (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : p.ap_raw[i], eps_t, t, k);
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

// ------------------------------------------------------------------------------------ small constraint kernels
__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_complete_bwd_kernel(int n, const float* __restrict__ ga,
                                                                          float* __restrict__ gap, CartConsts c) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 g = reinterpret_cast<const float2*>(ga)[i];
        gap[i] = complete_bwd_row(c, g.x, g.y);
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_resid_kernel(int n, const float* __restrict__ action,
                                                                   float* __restrict__ eq, float* __restrict__ ineq,
                                                                   CartConsts c) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        float h, g[6];
        eq_ineq(c, a.x, a.y, h, g);
        if (eq) eq[i] = h;
        if (ineq) {
            float2* o = reinterpret_cast<float2*>(ineq + (size_t)i * 6);
            o[0] = make_float2(g[0], g[1]); o[1] = make_float2(g[2], g[3]); o[2] = make_float2(g[4], g[5]);
        }
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_ipg_kernel(int n, const float* __restrict__ action,
                                                                 float* __restrict__ step, CartConsts c) {
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        const float gp = reduced_grad(c, c.partial == 0 ? a.x : a.y);
        const float go = -(gp * c.C_p) * c.C_o_inv;
        reinterpret_cast<float2*>(step)[i] = c.partial == 0 ? make_float2(gp, go) : make_float2(go, gp);
    }
}

__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_lagrangian_kernel(int n, const float* __restrict__ action,
                                                                        const float* __restrict__ nu, float scale,
                                                                        float* __restrict__ loss_out,
                                                                        float* __restrict__ grad_action,
                                                                        float* __restrict__ grad_nu, CartConsts c,
                                                                        float* __restrict__ partials_out) {
    __shared__ float red[(RPO_BLOCK / RPO_WAVE) * 7];
    float nuv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) nuv[j] = nu[j];
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};   // loss, dist_0..5
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        float row_loss, dist[6], g0, g1;
        lagrangian_row(c, a.x, a.y, nuv, row_loss, dist, g0, g1);
        acc[0] += row_loss;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[1 + j] += dist[j];
        if (grad_action) reinterpret_cast<float2*>(grad_action)[i] = make_float2(scale * g0, scale * g1);
    }
    const int lane = threadIdx.x & (RPO_WAVE - 1), wave = threadIdx.x / RPO_WAVE;
    rpo_wave_reduce_many(acc, 0u);
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (lane == 0) red[wave * 7 + k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 7) {
        float r = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) r += red[w * 7 + threadIdx.x];
        r *= scale;
        // (a launch that adds into loss_out / grad_nu is always ONE workgroup -- a fixed order; wider batches leave their
        //  per-workgroup sums in partials_out: see rpo_cartsafe_lagrangian)
        if (partials_out) partials_out[(size_t)blockIdx.x * 8 + threadIdx.x] = r;
        else if (threadIdx.x == 0) { if (loss_out) atomicAdd(loss_out, r); }
        else if (grad_nu) atomicAdd(grad_nu + threadIdx.x - 1, r);
    }
}

// Sum of G per-workgroup partial vectors [G][8] (K <= 8 used) in a FIXED order: thread t takes workgroups t, t + 256, ..., the 256
// sums meet in an LDS tree; out_k[0] += total_k for the non-NULL outputs.  One workgroup.
__global__ __launch_bounds__(RPO_BLOCK) void lagrangian_reduce_kernel(int G, int K, const float* __restrict__ partials,
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

// ====================================================================================================== C ABI
extern "C" {

int rpo_cartsafe_reset(int n_envs, float* state, int* ep_len, float* ep_ret, const unsigned* ep_count,
                       unsigned long long seed, unsigned env_id_base, void* stream) {
    if (n_envs <= 0) return RPO_ERR_ARG;
    if (!state || !ep_len || !ep_ret) return RPO_ERR_NULL;
    hipLaunchKernelGGL(cartsafe_reset_kernel, dim3(rpo_grid_for(n_envs)), dim3(RPO_BLOCK), 0, (hipStream_t)stream,
                       n_envs, state, ep_len, ep_ret, ep_count, (uint64_t)seed, (uint32_t)env_id_base);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_step(int n_envs, float* state, const float* action, int* ep_len, float* ep_ret, unsigned* ep_count,
                      float* rows, long long cap_steps, float* stats, int stats_cap, long long* ctrl,
                      const float* consts_host, int partial, int max_episode_steps, int auto_reset,
                      float viol_thresh, unsigned long long seed, unsigned env_id_base, void* stream) {
    if (n_envs <= 0 || max_episode_steps <= 0) return RPO_ERR_ARG;
    if (!state || !action || !ep_len || !ep_ret || !ep_count) return RPO_ERR_NULL;
    if (rows && cap_steps <= 0) return RPO_ERR_ARG;
    if (stats && stats_cap <= 0) return RPO_ERR_ARG;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    const int tiles_ok = ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(rows)) & 15u) == 0;
    StepArgs a{n_envs, state, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, stats_cap, ctrl,
               max_episode_steps, auto_reset, viol_thresh, (uint64_t)seed, (uint32_t)env_id_base, tiles_ok};
    hipLaunchKernelGGL(cartsafe_step_kernel, dim3(rpo_grid_for(n_envs)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_act_project(int n, const float* ap_raw, const float* noise, float* action, int* iters,
                             int noise_mode, float eps_start, float eps_end, float eps_decay, float box_lo,
                             float box_hi, int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                             const float* consts_host, int partial, unsigned long long seed, unsigned env_id_base,
                             const long long* ctrl, float* stats, int stats_cap, void* stream) {
    if (n <= 0 || max_steps < 0 || noise_mode < RPO_NOISE_NONE || noise_mode > RPO_NOISE_CLIP_ONLY) return RPO_ERR_ARG;
    if (!action || (noise_mode != RPO_NOISE_UNIFORM && !ap_raw)) return RPO_ERR_NULL;
    if (noise_mode == RPO_NOISE_EXPLICIT && !noise) return RPO_ERR_NULL;
    if (stats && stats_cap <= 0) return RPO_ERR_ARG;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    ActArgs a{n, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, max_steps,
              corr_lr, corr_eps, corr_momentum, (uint64_t)seed, (uint32_t)env_id_base, ctrl, stats, stats_cap};
    hipLaunchKernelGGL(cartsafe_act_project_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, a, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_complete_bwd(int n, const float* grad_action, float* grad_ap, const float* consts_host,
                              int partial, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!grad_action || !grad_ap) return RPO_ERR_NULL;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    hipLaunchKernelGGL(cartsafe_complete_bwd_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream,
                       n, grad_action, grad_ap, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_resid(int n, const float* action, float* eq_resid, float* ineq_resid, const float* consts_host,
                       int partial, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!action) return RPO_ERR_NULL;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    hipLaunchKernelGGL(cartsafe_resid_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       action, eq_resid, ineq_resid, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_ineq_partial_grad(int n, const float* action, float* step, const float* consts_host, int partial,
                                   void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!action || !step) return RPO_ERR_NULL;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    hipLaunchKernelGGL(cartsafe_ipg_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action,
                       step, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_lagrangian(int n, const float* action, const float* nu, float scale, float* loss_out,
                            float* grad_action, float* grad_nu, const float* consts_host, int partial, void* stream) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!action || !nu) return RPO_ERR_NULL;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    if (n > RPO_BLOCK && (loss_out || grad_nu)) {
        // More than one workgroup would add its partial sums with float atomics in arrival order (found in round 4: the 2^20-row
        // updates were not bitwise reproducible from run to run).  Deterministic form: every workgroup leaves its 7 sums in a
        // scratch area, ONE workgroup adds them in a fixed order, then the elementwise launch writes grad_action -- whose
        // first 8 G floats ARE the scratch area (G = workgroups <= n / 256, so 8 G <= n / 32 floats of the 2 n).  Without a
        // grad_action buffer one workgroup recomputes and sums every row.
        const int G = rpo_grid_for(n);
        if (grad_action && (long long)8 * G <= (long long)2 * n) {
            hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, (float*)nullptr, (float*)nullptr, c, grad_action);
            RPO_LAUNCH_CHECK();
            hipLaunchKernelGGL(lagrangian_reduce_kernel, dim3(1), dim3(RPO_BLOCK), 0, (hipStream_t)stream, G, 7,
                               (const float*)grad_action, loss_out, grad_nu);
            RPO_LAUNCH_CHECK();
            hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, grad_action, (float*)nullptr, c, (float*)nullptr);
            RPO_LAUNCH_CHECK();
            return 0;
        }
        if (grad_action) {
            hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(G), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                               (float*)nullptr, grad_action, (float*)nullptr, c, (float*)nullptr);
            RPO_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(1), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, action, nu, scale,
                           loss_out, (float*)nullptr, grad_nu, c, (float*)nullptr);
        RPO_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       action, nu, scale, loss_out, grad_action, grad_nu, c, (float*)nullptr);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
