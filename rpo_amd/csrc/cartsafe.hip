// CartSafe-v0 on MI355X: vectorised env step (+TimeLimit, violations, replay scatter, auto-reset), exploration +
// equation solver + GRG projection, and the constraint pieces of the actor loss.  One lane per env / sample.
// Reference semantics: rpo/env/classic_control/cartpole.py, rpo/algo/rpo_ddpg.py (cited per function in
// include/rpo_hip.h).  All kernels are HBM-bound streaming kernels (O(100) flops per ~150 B).
#include "common.h"

namespace {

struct CartConsts {
    float C[2], C_p, C_o_inv, b;
    float G[12], d[6], G_r[6], d_r[6];
    int partial;
};

int load_consts(CartConsts& c, const float* h, int partial) {
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
    h = c.b - (a0 * c.C[0] + a1 * c.C[1]);                       // eq_resid, cartpole.py:375-376
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = (a0 * c.G[2 * i] + a1 * c.G[2 * i + 1]) - c.d[i];   // ineq_resid :378-379
}

__device__ __forceinline__ void reset_state(float (&s)[6], uint64_t seed, uint32_t env_id, uint32_t episode) {
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
                                          float (&ns)[6], float4 (&row)[6], float (&st)[kStepStats]) {
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
    const int len = p.ep_len[i] + 1;
    const bool done = terminated || len >= p.max_episode_steps;                        // gym TimeLimit
    const float reward = 1.0f;                                                         // cartpole.py:215-221
    const float ret = p.ep_ret[i] + reward;

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
        const unsigned ep = p.ep_count[i] + 1u;
        p.ep_count[i] = ep;
        reset_state(ns, p.seed, p.env_id_base + (uint32_t)i, ep);
        p.ep_len[i] = 0;
        p.ep_ret[i] = 0.0f;
    } else {
        p.ep_len[i] = len;
        p.ep_ret[i] = ret;
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
            cart_lane(p, c, i, s, a, ns, row, st);
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
                float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + base) * RPO_CART_ROW);
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int ch = k * RPO_BLOCK + tid;         // float4 chunk of the 24 KB row block
                    const int r = ch / 6, part = ch - r * 6;
                    gr[ch] = rows_s[r * 7 + part];
                }
            }
            __syncthreads();                                   // LDS tiles are rewritten by the next iteration
        } else if (i < p.n) {
            // ---- partial last tile: direct per-lane accesses
            load_state(p.state + (size_t)i * 6, s);
            const float2 a = reinterpret_cast<const float2*>(p.action)[i];
            cart_lane(p, c, i, s, a, ns, row, st);
            if (p.rows) {
                float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_CART_ROW);
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
    // ineq_partial_grad, cartpole.py:396-403: sign-based (sub)gradient of the reduced inequalities
    float grad = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) grad += ((ap * c.G_r[i] - c.d_r[i]) > 0.0f) ? c.G_r[i] : 0.0f;
    return grad;
}

__global__ __launch_bounds__(RPO_BLOCK) void cartsafe_act_project_kernel(ActArgs p, CartConsts c) {
    __shared__ float red[RPO_BLOCK / RPO_WAVE];
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);   // eps_decay, ddpg_pa.py:118-119
    float iters_sum = 0.0f;
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < p.n; i += gridDim.x * RPO_BLOCK) {
        float ap = (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : p.ap_raw[i];
        if (p.noise_mode == RPO_NOISE_EXPLICIT) {
            ap = fminf(fmaxf(ap + eps_t * p.noise[i], p.box_lo), p.box_hi);                  // ddpg_pa.py:108-110
        } else if (p.noise_mode == RPO_NOISE_PHILOX) {
            const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
            ap = fminf(fmaxf(ap + eps_t * rpo_normal(r.x, r.y), p.box_lo), p.box_hi);
        } else if (p.noise_mode == RPO_NOISE_UNIFORM) {                                      // model/utils.py:53-62
            const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
            const float scale = (p.box_hi - p.box_lo) * 0.5f;
            ap = scale * (2.0f * rpo_u01(r.x) - 1.0f) + (p.box_lo + scale);
        } else if (p.noise_mode == RPO_NOISE_CLIP_ONLY) {
            ap = fminf(fmaxf(ap, p.box_lo), p.box_hi);
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
        reinterpret_cast<float2*>(p.action)[i] = c.partial == 0 ? make_float2(ap, ao) : make_float2(ao, ap);
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
    const float k = -(c.C_p * c.C_o_inv);
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 g = reinterpret_cast<const float2*>(ga)[i];
        gap[i] = c.partial == 0 ? g.x + k * g.y : g.y + k * g.x;
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
                                                                        float* __restrict__ grad_nu, CartConsts c) {
    __shared__ float red[(RPO_BLOCK / RPO_WAVE) * 7];
    float nuv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) nuv[j] = nu[j];
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};   // loss, dist_0..5
    for (int i = blockIdx.x * RPO_BLOCK + threadIdx.x; i < n; i += gridDim.x * RPO_BLOCK) {
        const float2 a = reinterpret_cast<const float2*>(action)[i];
        float h, g[6];
        eq_ineq(c, a.x, a.y, h, g);
        float g0 = 0.0f, g1 = 0.0f;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float dist = fmaxf(g[j], 0.0f);                   // ineq_dist, cartpole.py:385-387
            acc[0] += nuv[j] * dist;                                // Dual.forward, dual.py:63-65
            acc[1 + j] += dist;
            if (g[j] > 0.0f) { g0 += nuv[j] * c.G[2 * j]; g1 += nuv[j] * c.G[2 * j + 1]; }
        }
        if (grad_action) reinterpret_cast<float2*>(grad_action)[i] = make_float2(scale * g0, scale * g1);
    }
    const int lane = threadIdx.x & (RPO_WAVE - 1), wave = threadIdx.x / RPO_WAVE;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const float r = rpo_wave_sum(acc[k]);
        if (lane == 0) red[wave * 7 + k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        float r = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) r += red[w * 7 + threadIdx.x];
        r *= scale;
        if (threadIdx.x == 0) { if (loss_out) atomicAdd(loss_out, r); }
        else if (grad_nu) atomicAdd(grad_nu + threadIdx.x - 1, r);
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
    hipLaunchKernelGGL(cartsafe_lagrangian_kernel, dim3(rpo_grid_for(n)), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n,
                       action, nu, scale, loss_out, grad_action, grad_nu, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
