// EVOPF-v0 kernels (reference: rpo/env/electrical_grid/evopf.py, data/demand.py, data/price.py; projection loop of
// rpo/algo/rpo_ddpg.py:72-77,266-305).  One wavefront per env lane / batch row -- see evopf_dev.h.
#include "evopf_dev.h"
#include "heads_dev.h"

using namespace rpo_evopf_dev;

namespace {

__device__ __forceinline__ void load_row(float* dst, const float* __restrict__ src, int n) {
    for (int i = lane_id(); i < n; i += RPO_WAVE) dst[i] = src[i];
}

// Env lanes per workgroup of the one-wave-per-lane kernels: four waves = one on each SIMD of a CU.  As 64-thread workgroups the
// dispatcher packed several waves onto one SIMD while others stayed empty -- and two issue-bound waves on a SIMD run at 3/4 of
// the speed each: the 1024-lane projection took 171 us inside the training windows and 123 us in this form.
constexpr int kActWaves = 4;

// ------------------------------------------------------------------------------------------------------- reset
__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_reset_kernel(int n, float* __restrict__ state, int* __restrict__ ep_len,
                                                               float* __restrict__ ep_ret,
                                                               const unsigned* __restrict__ ep_count,
                                                               const float* __restrict__ consts, uint64_t seed,
                                                               uint32_t env_id_base) {
    __shared__ Ws wss[kActWaves];
    Ws& w = wss[threadIdx.x / RPO_WAVE];
    const int i = blockIdx.x * kActWaves + threadIdx.x / RPO_WAVE, tid = lane_id();
    if (i >= n) return;
    load_consts(w, consts);
    sync();
    episode_obs(w, w.s, seed, env_id_base + (uint32_t)i, ep_count ? ep_count[i] : 0u, 0);
    if (tid < NE) w.s[2 * NB + tid] = kBInit;
    sync();
    for (int k = tid; k < NS; k += RPO_WAVE) state[(size_t)i * NS + k] = w.s[k];
    if (tid == 0) { ep_len[i] = 0; ep_ret[i] = 0.0f; }
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
    const float* consts;
    uint64_t seed;
    uint32_t env_id_base;
};

// EVOPFEnv.step (evopf.py:348-366) + Battery.step (:74-102) + the bookkeeping of the run loop (rpo_ddpg.py:120-145):
// violations of the pre-step observation, reward, next hour of the episode data, replay scatter, statistics, auto-reset.
__device__ __forceinline__ void evopf_step_lane(const StepArgs& p, Ws& w, float* row, int i, long long t) {
    RPO_FP_STRICT
    const int tid = lane_id();
    load_consts(w, p.consts);
    load_row(w.s, p.state + (size_t)i * NS, NS);
    load_row(w.a, p.action + (size_t)i * NY, NY);
    sync();
    flows(w);
    eq_resid(w);
    ineq_resid(w);
    const uint32_t env_id = p.env_id_base + (uint32_t)i;
    const int len = p.ep_len[i] + 1;                                   // == the loaders' counter (demand.py:69-73)
    const unsigned ep = p.ep_count[i];
    const bool data_done = len >= T;
    const bool done = data_done || len >= p.max_episode_steps;

    // reward = wg * (-obj_fn) + we * (-sum(pe * price)) (evopf.py:350-353,509-518; Battery.step :99-100)
    float part = 0.0f;
    if (tid < NG) {
        const float pg = w.a[PG0 + tid];
        part = -kWg * (w.c[RPO_EVOPF_C_QUAD + tid] * pg * pg + w.c[RPO_EVOPF_C_LIN + tid] * pg) -
               kWe * (w.a[PE0 + tid] * w.s[2 * NB + NE]);
    }
    const float reward = rpo_wave_sum(part) - kWg * w.c[RPO_EVOPF_C_CONST];
    const float max_eq = rpo_wave_max(tid < NEQ ? fabsf(w.eq[tid]) : 0.0f);
    const float max_ineq = rpo_wave_max(tid < NINEQ ? fmaxf(w.ineq[tid], 0.0f) : 0.0f);

    // transition row: state | action | next_state | reward | done | eq_viol | ineq_viol (ReplayBuffer.add, buffer.py:22-29)
    for (int k = tid; k < NS; k += RPO_WAVE) row[k] = w.s[k];
    if (tid < NY) row[NS + tid] = w.a[tid];
    float* nxt = row + NS + NY;
    episode_obs(w, nxt, p.seed, env_id, ep, len);
    if (tid < NE) {                                                    // Battery.step :86-97
        const float soc = w.s[2 * NB + tid];
        float p_max, p_min;
        battery_bounds(soc, p_max, p_min);
        float x = fminf(fmaxf(w.a[PE0 + tid], p_min), p_max);
        x = x >= 0.0f ? x * kEtaIn : x / kEtaOut;                       // process_action :145-149
        nxt[2 * NB + tid] = soc + x;
    }
    if (tid == 0) { row[2 * NS + NY] = reward; row[2 * NS + NY + 1] = done ? 1.0f : 0.0f; }
    if (tid < NEQ) row[2 * NS + NY + 2 + tid] = w.eq[tid];
    if (tid < NINEQ) row[2 * NS + NY + 2 + NEQ + tid] = fmaxf(w.ineq[tid], 0.0f);
    if (tid < RPO_EVOPF_ROW - (2 * NS + NY + 2 + NEQ + NINEQ)) row[2 * NS + NY + 2 + NEQ + NINEQ + tid] = 0.0f;
    sync();
    {   // failure detection (RPO_CTRL_NONFINITE): the reference has no assertion in EVOPFEnv.step (evopf.py:348-366) and would go
        // on with NaNs; here a non-finite entry anywhere in the transition row (action, next state, reward, violations) stops the run
        bool bad = false;
        for (int k = tid; k < 2 * NS + NY + 2 + NEQ + NINEQ; k += RPO_WAVE) bad |= !__builtin_isfinite(row[k]);
        if (__ballot(bad) != 0ull) rpo_flag_nonfinite(p.ctrl, tid == 0);
    }
    if (p.rows) {
        const long long ring = (t % p.cap_steps) * (long long)p.n + i;
        float4* dst = reinterpret_cast<float4*>(p.rows + (size_t)ring * RPO_EVOPF_ROW);
        if (tid < RPO_EVOPF_ROW / 4) dst[tid] = reinterpret_cast<const float4*>(row)[tid];
    }
    // the lane continues from the next observation, or from a fresh episode (env.reset() after a done, rpo_ddpg.py:142)
    const float ret = p.ep_ret[i] + reward;
    if (done && p.auto_reset) {
        episode_obs(w, w.s, p.seed, env_id, ep + 1u, 0);
        if (tid < NE) w.s[2 * NB + tid] = kBInit;
        sync();
        for (int k = tid; k < NS; k += RPO_WAVE) p.state[(size_t)i * NS + k] = w.s[k];
        if (tid == 0) { p.ep_count[i] = ep + 1u; p.ep_len[i] = 0; p.ep_ret[i] = 0.0f; }
    } else {
        for (int k = tid; k < NS; k += RPO_WAVE) p.state[(size_t)i * NS + k] = nxt[k];
        if (tid == 0) { p.ep_len[i] = len; p.ep_ret[i] = ret; }
    }
    if (p.stats) {                                             // (uniform values: one request per lane, rpo_stats_commit)
        const float dn = done ? 1.0f : 0.0f;
        const float vals[10] = {reward, max_ineq, max_eq, fmaxf(max_ineq, max_eq) > p.viol_thresh ? 1.0f : 0.0f, max_ineq, max_eq,
                                dn, dn * ret, dn * (float)len, (done && data_done) ? 1.0f : 0.0f};
        const int slot[10] = {RPO_STAT_REWARD_SUM, RPO_STAT_MAX_INEQ_SUM, RPO_STAT_MAX_EQ_SUM, RPO_STAT_VIOL_COUNT,
                              RPO_STAT_MAX_INEQ_MAX, RPO_STAT_MAX_EQ_MAX, RPO_STAT_EPISODES, RPO_STAT_RETURN_SUM,
                              RPO_STAT_LENGTH_SUM, RPO_STAT_TERMINATED};
        rpo_stats_commit(vals, 3u << 4, slot, rpo_stats_row(p.stats, p.stats_cap, t));
    }
}

__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_step_kernel(StepArgs p) {
    __shared__ Ws wss[kActWaves];
    __shared__ __align__(16) float rows_s[kActWaves][RPO_EVOPF_ROW];
    const int wave = threadIdx.x / RPO_WAVE, i = blockIdx.x * kActWaves + wave;
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    if (i < p.n) evopf_step_lane(p, wss[wave], rows_s[wave], i, t);   // (every wave reaches the epilogue's workgroup barrier)
    rpo_step_epilogue(p.ctrl, t, p.stats, p.stats_cap);
}

// ------------------------------------------------------------------------------- explore + complete + project
struct ActArgs {
    int n;
    const float* state;
    int state_stride;
    const float* ap_raw;
    const float* noise;
    float* action;
    int* iters;
    int noise_mode;
    int ap_is_raw;
    float eps_start, eps_end, eps_decay;
    int max_steps;
    float corr_lr, corr_eps, corr_momentum;
    float newton_tol;
    int newton_iters;
    const float* consts;
    uint64_t seed;
    uint32_t env_id_base;
    const long long* ctrl;
    float* stats;
    int stats_cap;
    int place;     // rpo_tuning(RPO_TUNE_EVOPF_PLACE): 0 none; 1 only the workgroups on XCDs 0-1 work; 2 only those on XCDs 2-7
};

// take_action's exploration + clip to the state-dependent box (agent/ddpg_pa.py:101-112, model/utils.py:53-62,90-101,
// EVOPFEnv.update evopf.py:769-783) -> complete_partial -> grad_steps, one lane per wave.
__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_act_project_kernel(ActArgs p) {
    __shared__ Ws wss[kActWaves];
    Ws& w = wss[threadIdx.x / RPO_WAVE];
    // A projection of a sampled BATCH (<= 512 rows: the target actions of the critic update, on the iteration's critical path)
    // shares the chip with the 1024-lane projection of the next rollout (one wave per SIMD, the other branch of the window,
    // which has slack): its waves then sit on SIMDs that already hold a wave, and both are bound by instruction issue.  Raised
    // priority gives the batch's waves the issue slots: 171 -> 1xx us measured for it, the rollout's waves on those SIMDs lag.
    int blk = blockIdx.x;
    if (p.place) {                                               // (experiment: block b runs on XCD b % 8 -- observed, not promised)
        const int x = blk & 7, q = blk >> 3;
        if (p.place == 1) { if (x >= 2) return; blk = q * 2 + x; }
        else { if (x < 2) return; blk = q * 6 + (x - 2); }
    }
    if ((p.place ? p.n : (int)gridDim.x * kActWaves) <= 512) __builtin_amdgcn_s_setprio(3);
    const int i = blk * kActWaves + threadIdx.x / RPO_WAVE, tid = lane_id();
    if (i >= p.n) return;                                        // (waves never meet at a workgroup barrier)
    const long long t = p.ctrl ? p.ctrl[RPO_CTRL_T] : 0;
    load_consts(w, p.consts);
    load_row(w.s, p.state + (size_t)i * p.state_stride, NS);
    sync();
    float z = 0.0f;
    if (tid < NP) {
        const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);
        const uint32_t env_id = p.env_id_base + (uint32_t)i;
        float lo, hi;
        partial_box(w, tid, lo, hi);
        z = (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : p.ap_raw[(size_t)i * NP + tid];
        if (p.ap_is_raw && p.noise_mode != RPO_NOISE_UNIFORM) {         // BoxConstraint.__call__ (model/utils.py:40-51,75-88)
            const float scale = (hi - lo) * 0.5f;
            z = scale * tanhf(z) + (lo + scale);
        }
        if (p.noise_mode == RPO_NOISE_EXPLICIT) {
            z = rpo_clamp(z + eps_t * p.noise[(size_t)i * NP + tid], lo, hi);
        } else if (p.noise_mode == RPO_NOISE_PHILOX) {
            const rpo_u4 r = rpo_philox(p.seed, env_id, (uint32_t)t, RPO_STREAM_ACT, (uint32_t)(tid >> 1));
            const float nz = (tid & 1) ? rpo_normal(r.z, r.w) : rpo_normal(r.x, r.y);
            z = rpo_clamp(z + eps_t * nz, lo, hi);
        } else if (p.noise_mode == RPO_NOISE_UNIFORM) {
            const rpo_u4 r = rpo_philox(p.seed, env_id, (uint32_t)t, RPO_STREAM_ACT, (uint32_t)(tid >> 2));
            const uint32_t word = (tid & 3) == 0 ? r.x : ((tid & 3) == 1 ? r.y : ((tid & 3) == 2 ? r.z : r.w));
            const float scale = (hi - lo) * 0.5f;
            z = scale * (2.0f * rpo_u01(word) - 1.0f) + (lo + scale);
        } else if (p.noise_mode == RPO_NOISE_CLIP_ONLY) {
            z = rpo_clamp(z, lo, hi);
        }
    }
    const RowLane L = make_row_lane(w);
    complete_partial_v2(w, L, z, p.newton_tol, p.newton_iters);
    const int k = grad_steps_v2(w, L, p.max_steps, p.corr_lr, p.corr_eps, p.corr_momentum);
    if (tid < NY) p.action[(size_t)i * NY + tid] = w.a[tid];
    if (tid == 0) {
        if (p.iters) p.iters[i] = k;
        if (p.stats && k) atomicAdd(rpo_stats_row(p.stats, p.stats_cap, t) + RPO_STAT_PROJ_ITERS, (float)k);
    }
}

// PFFunction.backward (evopf.py:857-910) with the Jacobians re-evaluated at the completed action (the reference keeps
// those of the last Newton point, one update of size < tol earlier).  grad_ap [n,14] = dL/dz.
__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_complete_bwd_kernel(int n, const float* __restrict__ action,
                                                                      const float* __restrict__ grad_action,
                                                                      const float* __restrict__ grad_action_b,
                                                                      const float* __restrict__ grad_action2,
                                                                      float* __restrict__ grad_ap,
                                                                      const float* __restrict__ consts) {
    RPO_FP_STRICT
    __shared__ Ws wss[kActWaves];
    Ws& w = wss[threadIdx.x / RPO_WAVE];
    const int i = blockIdx.x * kActWaves + threadIdx.x / RPO_WAVE, tid = lane_id();
    if (i >= n) return;
    load_consts(w, consts);
    load_row(w.a, action + (size_t)i * NY, NY);
    // dl_dy = (grad_action [+ grad_action_b]) [+ grad_action2]: d(-Q)/da from the critic (twin critics: the two terms) + the
    // Lagrangian term, rpo_ddpg.py:319 / rpo_sac.py:331-337, added here instead of by launches of their own -- the same float
    // additions in the same order
    for (int j = tid; j < NY; j += RPO_WAVE) {
        float v = grad_action[(size_t)i * NY + j];
        if (grad_action_b) v = v + grad_action_b[(size_t)i * NY + j];
        if (grad_action2) v = v + grad_action2[(size_t)i * NY + j];
        w.dir[j] = v;
    }
    sync();
    flows(w);
    // step 3 (:865-880): dl/d(vm, va) through pg_slack = -eq[0] and qg_j = -eq[14 + spv_j]
    if (tid < 2 * NB) {
        float acc = jac_entry(w, 0, VM0 + tid) * w.dir[PG0];
        for (int j = 0; j < NG; ++j) acc += jac_entry(w, NB + kSpv[j], VM0 + tid) * w.dir[QG0 + j];
        w.vec[tid] = w.dir[VM0 + tid] - acc;                            // dl_dy_total on the voltage block (:886)
    }
    sync();
    // step 1 (:889-891): d_int = inv(J_newton)^T dl_dy_total[newton vars]
    const bool force_dyn = w.c[RPO_EVOPF_C_FLAGS] != RPO_EVOPF_STATIC_OK;
    bool solved = false;
    float dint = 0.0f;
    int mycol = tid;
    if (!force_dyn) {                                                   // static order (evopf_dev.h), accepted by its pivots
        float row[NN + 1];                                              // row tid of [J_newton^T | rhs]
#pragma unroll
        for (int c = 0; c < NN; ++c) row[c] = tid < NN ? jac_entry(w, kKeep[c], kNewtonVars[tid]) : 0.0f;
        row[NN] = tid < NN ? w.vec[kNewtonVars[tid] - VM0] : 0.0f;
        const float mypiv = gauss_jordan_static<NN, NN + 1, 0, TabNewtonT>(row);
        solved = pivots_ok<NN, 0>(mypiv);
        dint = row[NN] / mypiv;
    }
    if (!solved) {                                                      // partial pivoting
        float row[NN + 1];
#pragma unroll
        for (int c = 0; c < NN; ++c) row[c] = tid < NN ? jac_entry(w, kKeep[c], kNewtonVars[tid]) : 0.0f;
        row[NN] = tid < NN ? w.vec[kNewtonVars[tid] - VM0] : 0.0f;
        float mypiv;
        gauss_jordan_rows<NN, NN + 1, 0>(row, mycol, mypiv);
        dint = row[NN] / mypiv;
    }
    sync();
    if (tid < NN) w.old[mycol] = dint;                                  // d_int, indexed like kKeep
    sync();
    if (tid < NP) {
        float g;
        if (tid < 4) {                                                  // pg at pv gens (:894) + direct term (:907)
            g = -w.old[kPvPos[tid]] + w.dir[PG0 + 1 + tid];
        } else if (tid < 9) {                                           // vm at generator buses (:895-896)
            const int var = VM0 + kSpv[tid - 4];
            float acc = 0.0f;
            for (int r = 0; r < NN; ++r) acc += jac_entry(w, kKeep[r], var) * w.old[r];
            g = -acc + w.vec[var - VM0];
        } else if (tid == 9) {                                          // pe at the slack generator (:898)
            g = w.dir[PG0] + w.dir[PE0];
        } else {                                                        // pe at pv gens (:897)
            g = w.old[kPvPos[tid - 10]] + w.dir[PE0 + tid - 9];
        }
        grad_ap[(size_t)i * NP + tid] = g;
    }
}

__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_resid_kernel(int n, const float* __restrict__ state, int state_stride,
                                                               const float* __restrict__ action, float* __restrict__ eq_out,
                                                               float* __restrict__ ineq_out,
                                                               const float* __restrict__ consts) {
    __shared__ Ws wss[kActWaves];
    Ws& w = wss[threadIdx.x / RPO_WAVE];
    const int i = blockIdx.x * kActWaves + threadIdx.x / RPO_WAVE, tid = lane_id();
    if (i >= n) return;
    load_consts(w, consts);
    load_row(w.s, state + (size_t)i * state_stride, NS);
    load_row(w.a, action + (size_t)i * NY, NY);
    sync();
    flows(w);
    eq_resid(w);
    ineq_resid(w);
    if (eq_out && tid < NEQ) eq_out[(size_t)i * NEQ + tid] = w.eq[tid];
    if (ineq_out && tid < NINEQ) ineq_out[(size_t)i * NINEQ + tid] = w.ineq[tid];
}

__global__ __launch_bounds__(RPO_WAVE * kActWaves) void evopf_ipg_kernel(int n, const float* __restrict__ state, int state_stride,
                                                             const float* __restrict__ action, float* __restrict__ step_out,
                                                             const float* __restrict__ consts) {
    __shared__ Ws wss[kActWaves];
    Ws& w = wss[threadIdx.x / RPO_WAVE];
    const int i = blockIdx.x * kActWaves + threadIdx.x / RPO_WAVE, tid = lane_id();
    if (i >= n) return;
    load_consts(w, consts);
    load_row(w.s, state + (size_t)i * state_stride, NS);
    load_row(w.a, action + (size_t)i * NY, NY);
    sync();
    if (tid < NY) w.dir[tid] = 0.0f;                           // (all 43 components are written below: 28 others + 15 partial vars)
    sync();
    const RowLane L = make_row_lane(w);
    sync();
    grg_iteration_v2(w, L, true, 0.0f, 0.0f, 0.0f, false, w.dir);
    if (tid < NY) step_out[(size_t)i * NY + tid] = w.dir[tid];
}

// Vector-Jacobian product of eq_resid: out[v] = sum_r grad_eq[r] * d eq_resid[r] / d action[v] with the entries of
// eq_jac (evopf.py:614-661) -- what torch autograd computes through eq_resid (:520-546) in the Lagrangian baselines'
// actor loss (ddpg_lag.py:257-263), except for the battery columns, where autograd differentiates the +pe of the
// residual (:532) while eq_jac carries -I (hazard E1): `autograd_sign` != 0 flips them to the autograd value.
__global__ __launch_bounds__(RPO_WAVE) void evopf_eq_vjp_kernel(int n, const float* __restrict__ action,
                                                                const float* __restrict__ grad_eq, float* __restrict__ out,
                                                                int autograd_sign, const float* __restrict__ consts) {
    __shared__ Ws w;
    const int i = blockIdx.x, tid = threadIdx.x;
    load_consts(w, consts);
    load_row(w.a, action + (size_t)i * NY, NY);
    load_row(w.eq, grad_eq + (size_t)i * NEQ, NEQ);
    sync();
    flows(w);
    if (tid < NY) {
        float acc = 0.0f;
        for (int r = 0; r < NEQ; ++r) acc = fmaf(w.eq[r], jac_entry(w, r, tid), acc);
        if (autograd_sign && tid >= PE0) acc = -acc;
        out[(size_t)i * NY + tid] = acc;
    }
}

// mean_b sum_j nu_j relu(g_j(s_b, a_b)) with its gradients (Dual.forward dual.py:63-65 on ineq_dist, rpo_ddpg.py:312-319).
// The 58 inequalities are box bounds on single variables, so the work is elementwise: thread v of each of the 4 waves owns
// action component v (its upper and its lower bound) for the rows b = wave (mod 4); the four partial sums are combined in
// a fixed order by wave 0 -- one workgroup, no float atomics between workgroups, bitwise reproducible.
__global__ __launch_bounds__(RPO_BLOCK) void evopf_lagrangian_kernel(int n, const float* __restrict__ state, int state_stride,
                                                                     const float* __restrict__ action,
                                                                     const float* __restrict__ nu, float scale,
                                                                     float* __restrict__ loss_out,
                                                                     float* __restrict__ grad_action,
                                                                     float* __restrict__ grad_nu,
                                                                     const float* __restrict__ consts, int overwrite) {
    RPO_FP_STRICT
    __shared__ float red[RPO_BLOCK / RPO_WAVE][2 * RPO_WAVE];
    const int v = threadIdx.x & (RPO_WAVE - 1), q = threadIdx.x / RPO_WAVE;
    int up = -1, lo = -1;                                        // rows of ineq_resid (evopf.py:550-561) bounding component v
    float hi_b = 0.0f, lo_b = 0.0f;
    if (v < QG0) { up = v; lo = 5 + v; hi_b = consts[RPO_EVOPF_C_PMAX + v]; lo_b = consts[RPO_EVOPF_C_PMIN + v]; }
    else if (v < VM0) { up = 10 + v - QG0; lo = 15 + v - QG0; hi_b = consts[RPO_EVOPF_C_QMAX + v - QG0]; lo_b = consts[RPO_EVOPF_C_QMIN + v - QG0]; }
    else if (v < VA0) { up = 20 + v - VM0; lo = 34 + v - VM0; hi_b = consts[RPO_EVOPF_C_VMAX + v - VM0]; lo_b = consts[RPO_EVOPF_C_VMIN + v - VM0]; }
    else if (v >= PE0 && v < NY) { up = 48 + v - PE0; lo = 53 + v - PE0; }
    const float nu_up = up >= 0 ? nu[up] : 0.0f, nu_lo = lo >= 0 ? nu[lo] : 0.0f;
    float acc_up = 0.0f, acc_lo = 0.0f;
    if (v < NY) {
        // 64 rows of loads in flight (a whole batch of 256 in one round trip), then the rows in order (the sums keep their association; one dependent round trip per
        // row made this one-workgroup launch 25 us at batch 256)
        constexpr int W = RPO_BLOCK / RPO_WAVE, CH = 64;
        for (int b0 = q; b0 < n; b0 += W * CH) {
            float av[CH], sv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int b = b0 + u * W, bc = b < n ? b : n - 1;
                av[u] = up >= 0 ? action[(size_t)bc * NY + v] : 0.0f;
                sv[u] = v >= PE0 ? state[(size_t)bc * state_stride + 2 * NB + v - PE0] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int b = b0 + u * W;
                if (b < n) {
                    float g = 0.0f;
                    if (up >= 0) {
                        const float a = av[u];
                        if (v >= PE0) battery_bounds(sv[u], hi_b, lo_b);
                        const float ru = a - hi_b, rl = lo_b - a;
                        acc_up += fmaxf(ru, 0.0f);
                        acc_lo += fmaxf(rl, 0.0f);
                        g = (ru > 0.0f ? nu_up : 0.0f) - (rl > 0.0f ? nu_lo : 0.0f);      // ineq_jac^T (nu * 1[g > 0]), :663-707
                    }
                    if (grad_action) grad_action[(size_t)b * NY + v] = scale * g;
                }
            }
        }
    }
    red[q][v] = acc_up;
    red[q][RPO_WAVE + v] = acc_lo;
    __syncthreads();
    if (q == 0) {
        float su = 0.0f, sl = 0.0f;
        for (int w = 0; w < RPO_BLOCK / RPO_WAVE; ++w) { su += red[w][v]; sl += red[w][RPO_WAVE + v]; }
        if (grad_nu && up >= 0) { grad_nu[up] += scale * su; grad_nu[lo] += scale * sl; }   // one writer per address
        const float total = rpo_wave_sum(nu_up * su + nu_lo * sl);
        if (v == 0 && loss_out) *loss_out = overwrite ? 0.0f + scale * total : *loss_out + scale * total;
    }
}

// Box of basic action j of row i straight from global memory (EVOPFEnv.update, evopf.py:769-783)
__device__ __forceinline__ void row_box(const float* __restrict__ state, int state_stride, const float* __restrict__ consts,
                                        int i, int j, float& lo, float& hi) {
    if (j < 4) { lo = consts[RPO_EVOPF_C_PMIN + 1 + j]; hi = consts[RPO_EVOPF_C_PMAX + 1 + j]; }
    else if (j < 9) { lo = consts[RPO_EVOPF_C_VMIN + kSpv[j - 4]]; hi = consts[RPO_EVOPF_C_VMAX + kSpv[j - 4]]; }
    else battery_bounds(state[(size_t)i * state_stride + 2 * NB + j - 9], hi, lo);
}

// Squashed-Gaussian policy head with the state-dependent box (GaussianSharedPolicy.forward, model/policy.py:53-66, with
// BoxConstraint.update_box; PDSAC_PA.take_action's clip, agent/sac_pa.py:111): raw [n,28] = (mean[14] | log-std[14]),
// eps [n,14] -> ap [n,14], logp [n] = sum over the 14 dimensions.  One 16-lane group per row.
__global__ __launch_bounds__(RPO_BLOCK) void evopf_gauss_head_kernel(int n, const float* __restrict__ state, int state_stride,
                                                                     const float* __restrict__ raw,
                                                                     const float* __restrict__ eps, int deterministic,
                                                                     float* __restrict__ ap, float* __restrict__ logp,
                                                                     const float* __restrict__ consts) {
    const int j = threadIdx.x & 15;
    for (int i = blockIdx.x * (RPO_BLOCK / 16) + (threadIdx.x >> 4); i < n; i += gridDim.x * (RPO_BLOCK / 16)) {
        float lp = 0.0f;
        if (j < NP) {
            float lo, hi;
            row_box(state, state_stride, consts, i, j, lo, hi);
            const float scale = (hi - lo) * 0.5f;
            ap[(size_t)i * NP + j] = rpo_head_dev::gauss_head_row(raw[(size_t)i * 2 * NP + j], raw[(size_t)i * 2 * NP + NP + j],
                                                                  eps[(size_t)i * NP + j], scale, lo + scale, lo, hi,
                                                                  deterministic, &lp);
        }
        lp += __shfl_xor(lp, 1, 16); lp += __shfl_xor(lp, 2, 16); lp += __shfl_xor(lp, 4, 16); lp += __shfl_xor(lp, 8, 16);
        if (j == 0 && logp) logp[i] = lp;
    }
}

// its backward: draw [n,28] = d loss / d (mean | log-std heads) given dap [n,14] and the coefficient of log pi
__global__ __launch_bounds__(RPO_BLOCK) void evopf_gauss_head_bwd_kernel(int n, const float* __restrict__ state,
                                                                         int state_stride, const float* __restrict__ raw,
                                                                         const float* __restrict__ eps,
                                                                         const float* __restrict__ dap, float dlogp,
                                                                         float* __restrict__ draw,
                                                                         const float* __restrict__ consts) {
    for (int idx = blockIdx.x * RPO_BLOCK + threadIdx.x; idx < n * NP; idx += gridDim.x * RPO_BLOCK) {
        const int i = idx / NP, j = idx - i * NP;
        float lo, hi;
        row_box(state, state_stride, consts, i, j, lo, hi);
        const float scale = (hi - lo) * 0.5f;
        const float2 g = rpo_head_dev::gauss_head_bwd_row(raw[(size_t)i * 2 * NP + j], raw[(size_t)i * 2 * NP + NP + j], eps[idx],
                                                          dap[idx], dlogp, scale, lo + scale, lo, hi);
        draw[(size_t)i * 2 * NP + j] = g.x;
        draw[(size_t)i * 2 * NP + NP + j] = g.y;
    }
}

// d/d(raw) of ap = clip(scale(s) * tanh(raw) + base(s) + eps_t * noise, lo(s), hi(s)) (SharedPolicy.forward
// model/policy.py:24-33 with the volatile box of evopf.py:769-783, then take_action ddpg_pa.py:108-110); elementwise.
__global__ __launch_bounds__(RPO_BLOCK) void evopf_tanh_box_bwd_kernel(int n, const float* __restrict__ state,
                                                                       int state_stride, const float* __restrict__ raw,
                                                                       const float* __restrict__ noise, float eps_start,
                                                                       float eps_end, float eps_decay,
                                                                       const long long* __restrict__ ctrl,
                                                                       const float* __restrict__ dap,
                                                                       float* __restrict__ dout,
                                                                       const float* __restrict__ consts) {
    RPO_FP_STRICT
    const long long t = ctrl ? ctrl[RPO_CTRL_T] : 0;
    const float eps_t = fmaxf(eps_end, eps_start - eps_decay * (float)t);
    for (int idx = blockIdx.x * RPO_BLOCK + threadIdx.x; idx < n * NP; idx += gridDim.x * RPO_BLOCK) {
        const int i = idx / NP, j = idx - i * NP;
        float lo, hi;
        if (j < 4) { lo = consts[RPO_EVOPF_C_PMIN + 1 + j]; hi = consts[RPO_EVOPF_C_PMAX + 1 + j]; }
        else if (j < 9) { lo = consts[RPO_EVOPF_C_VMIN + kSpv[j - 4]]; hi = consts[RPO_EVOPF_C_VMAX + kSpv[j - 4]]; }
        else battery_bounds(state[(size_t)i * state_stride + 2 * NB + j - 9], hi, lo);
        const float scale = (hi - lo) * 0.5f;
        const float th = tanhf(raw[idx]);
        float g = dap[idx] * scale * (1.0f - th * th);
        if (noise) {
            const float pre = scale * th + (lo + scale) + eps_t * noise[idx];
            if (pre < lo || pre > hi) g = 0.0f;                      // torch.clip's backward: zero outside [lo, hi]
        }
        dout[idx] = g;
    }
}

int check_common(int n, const void* a, const void* b, const float* consts) {
    if (n <= 0) return RPO_ERR_ARG;
    if (!a || !b || !consts) return RPO_ERR_NULL;
    return 0;
}

}  // namespace

// ====================================================================================================== C ABI
extern "C" {

int rpo_evopf_reset(int n_envs, float* state, int* ep_len, float* ep_ret, const unsigned* ep_count,
                    const float* consts_dev, unsigned long long seed, unsigned env_id_base, void* stream) {
    if (int e = check_common(n_envs, state, ep_len, consts_dev)) return e;
    if (!ep_ret) return RPO_ERR_NULL;
    hipLaunchKernelGGL(evopf_reset_kernel, dim3((n_envs + kActWaves - 1) / kActWaves), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, n_envs, state, ep_len,
                       ep_ret, ep_count, consts_dev, (uint64_t)seed, (uint32_t)env_id_base);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_step(int n_envs, float* state, const float* action, int* ep_len, float* ep_ret, unsigned* ep_count,
                   float* rows, long long cap_steps, float* stats, int stats_cap, long long* ctrl, int max_episode_steps,
                   int auto_reset, float viol_thresh, const float* consts_dev, unsigned long long seed,
                   unsigned env_id_base, void* stream) {
    if (int e = check_common(n_envs, state, action, consts_dev)) return e;
    if (!ep_len || !ep_ret || !ep_count) return RPO_ERR_NULL;
    if (max_episode_steps <= 0 || (rows && cap_steps <= 0) || (stats && stats_cap <= 0)) return RPO_ERR_ARG;
    StepArgs p{n_envs, state, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, stats_cap, ctrl, max_episode_steps,
               auto_reset, viol_thresh, consts_dev, (uint64_t)seed, (uint32_t)env_id_base};
    hipLaunchKernelGGL(evopf_step_kernel, dim3((n_envs + kActWaves - 1) / kActWaves), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, p);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_act_project(int n, const float* state, int state_stride, const float* ap_raw, const float* noise, float* action, int* iters,
                          int noise_mode, int ap_is_raw, float eps_start, float eps_end, float eps_decay, int max_steps,
                          float corr_lr, float corr_eps, float corr_momentum, float newton_tol, int newton_max_iters,
                          const float* consts_dev, unsigned long long seed, unsigned env_id_base, const long long* ctrl,
                          float* stats, int stats_cap, void* stream) {
    if (int e = check_common(n, state, action, consts_dev)) return e;
    if (max_steps < 0 || newton_max_iters <= 0 || (stats && stats_cap <= 0)) return RPO_ERR_ARG;
    if (noise_mode < RPO_NOISE_NONE || noise_mode > RPO_NOISE_CLIP_ONLY) return RPO_ERR_ARG;
    if (noise_mode != RPO_NOISE_UNIFORM && !ap_raw) return RPO_ERR_NULL;
    if (noise_mode == RPO_NOISE_EXPLICIT && !noise) return RPO_ERR_NULL;
    if (state_stride < RPO_EVOPF_STATE) return RPO_ERR_ARG;
    ActArgs p{n, state, state_stride, ap_raw, noise, action, iters, noise_mode, ap_is_raw, eps_start, eps_end, eps_decay, max_steps, corr_lr,
              corr_eps, corr_momentum, newton_tol, newton_max_iters, consts_dev, (uint64_t)seed, (uint32_t)env_id_base, ctrl,
              stats, stats_cap, 0};
    int blocks = (n + kActWaves - 1) / kActWaves;
    if (rpo_tune(RPO_TUNE_EVOPF_PLACE)) {
        p.place = n <= 512 ? 1 : 2;
        const int per = p.place == 1 ? 2 : 6;
        blocks = (blocks + per - 1) / per * 8;
    }
    hipLaunchKernelGGL(evopf_act_project_kernel, dim3(blocks), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, p);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_tanh_box_bwd(int n, const float* state, int state_stride, const float* raw, const float* noise,
                           float eps_start, float eps_end, float eps_decay, const long long* ctrl, const float* dap,
                           float* dout, const float* consts_dev, void* stream) {
    if (int e = check_common(n, state, raw, consts_dev)) return e;
    if (!dap || !dout) return RPO_ERR_NULL;
    if (state_stride < RPO_EVOPF_STATE) return RPO_ERR_ARG;
    hipLaunchKernelGGL(evopf_tanh_box_bwd_kernel, dim3(rpo_grid_for((long long)n * NP)), dim3(RPO_BLOCK), 0,
                       (hipStream_t)stream, n, state, state_stride, raw, noise, eps_start, eps_end, eps_decay, ctrl, dap,
                       dout, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_gauss_head(int n, const float* state, int state_stride, const float* raw, const float* eps, int deterministic,
                         float* ap_out, float* logp_out, const float* consts_dev, void* stream) {
    if (int e = check_common(n, state, raw, consts_dev)) return e;
    if (!eps || !ap_out) return RPO_ERR_NULL;
    if (state_stride < RPO_EVOPF_STATE) return RPO_ERR_ARG;
    hipLaunchKernelGGL(evopf_gauss_head_kernel, dim3(rpo_grid_for((long long)n * 16)), dim3(RPO_BLOCK), 0, (hipStream_t)stream,
                       n, state, state_stride, raw, eps, deterministic, ap_out, logp_out, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_gauss_head_bwd(int n, const float* state, int state_stride, const float* raw, const float* eps,
                             const float* dap, float dlogp, float* draw, const float* consts_dev, void* stream) {
    if (int e = check_common(n, state, raw, consts_dev)) return e;
    if (!eps || !dap || !draw) return RPO_ERR_NULL;
    if (state_stride < RPO_EVOPF_STATE) return RPO_ERR_ARG;
    hipLaunchKernelGGL(evopf_gauss_head_bwd_kernel, dim3(rpo_grid_for((long long)n * NP)), dim3(RPO_BLOCK), 0,
                       (hipStream_t)stream, n, state, state_stride, raw, eps, dap, dlogp, draw, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_complete_bwd(int n, const float* action, const float* grad_action, const float* grad_action_b, const float* grad_action2,
                           float* grad_ap, const float* consts_dev, void* stream) {
    if (int e = check_common(n, action, grad_action, consts_dev)) return e;
    if (!grad_ap) return RPO_ERR_NULL;
    hipLaunchKernelGGL(evopf_complete_bwd_kernel, dim3((n + kActWaves - 1) / kActWaves), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, n, action, grad_action,
                       grad_action_b, grad_action2, grad_ap, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_resid(int n, const float* state, int state_stride, const float* action, float* eq_out, float* ineq_out,
                    const float* consts_dev, void* stream) {
    if (int e = check_common(n, state, action, consts_dev)) return e;
    hipLaunchKernelGGL(evopf_resid_kernel, dim3((n + kActWaves - 1) / kActWaves), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, n, state, state_stride, action, eq_out,
                       ineq_out, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_ineq_partial_grad(int n, const float* state, int state_stride, const float* action, float* step_out, const float* consts_dev,
                                void* stream) {
    if (int e = check_common(n, state, action, consts_dev)) return e;
    if (!step_out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(evopf_ipg_kernel, dim3((n + kActWaves - 1) / kActWaves), dim3(RPO_WAVE * kActWaves), 0, (hipStream_t)stream, n, state, state_stride, action, step_out,
                       consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_eq_vjp(int n, const float* action, const float* grad_eq, float* grad_action, int autograd_sign,
                     const float* consts_dev, void* stream) {
    if (int e = check_common(n, action, grad_eq, consts_dev)) return e;
    if (!grad_action) return RPO_ERR_NULL;
    hipLaunchKernelGGL(evopf_eq_vjp_kernel, dim3(n), dim3(RPO_WAVE), 0, (hipStream_t)stream, n, action, grad_eq, grad_action,
                       autograd_sign, consts_dev);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_evopf_lagrangian(int n, const float* state, int state_stride, const float* action, const float* nu, float scale, float* loss_out,
                         float* grad_action, float* grad_nu, const float* consts_dev, int overwrite, void* stream) {
    if (int e = check_common(n, state, action, consts_dev)) return e;
    if (!nu) return RPO_ERR_NULL;
    hipLaunchKernelGGL(evopf_lagrangian_kernel, dim3(1), dim3(RPO_BLOCK), 0, (hipStream_t)stream, n, state, state_stride,
                       action, nu, scale, loss_out, grad_action, grad_nu, consts_dev, overwrite);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
