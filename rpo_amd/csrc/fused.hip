// Fused pipelines of the RPODDPG iteration on CartSafe-v0 (MI355X): at 4096 lanes / batch 256 every launch costs
// ~2 us of dispatch plus a cold start of its caches, so the row-local stages are chained inside one workgroup instead:
//
//   rollout        obs tile -> actor MLP (f32 MFMA) -> exploration noise + box clip -> equation solver -> GRG projection
//                  -> env step + violations + TimeLimit -> replay scatter -> statistics -> auto-reset
//                  (rpo_ddpg.py:93-145; replaces rpo_mlp_forward + rpo_cartsafe_act_project + rpo_cartsafe_step)
//   critic forward ReplayBuffer.sample -> pi_targ(s') -> Complete + Proj -> Q_targ(s', a') -> Q(s, a) -> TD target +
//                  Huber loss and dLoss/dQ (rpo_ddpg.py:165-174, 327-337; replaces rpo_replay_sample_gather, three
//                  rpo_mlp_forward launches, one projection launch and rpo_td_huber)
//
// One workgroup = 16 rows (lanes / batch samples) and 512 threads; the MLP tile forward is mlp_tile.h.
#include <stdlib.h>

#include <type_traits>

#include "cartsafe_dev.h"
#include "heads_dev.h"
#include "mlp_bwd.h"
#include "mlp_tile.h"
#include "pendulum_dev.h"
#include "rollout_args.h"
#include "rollout_env.h"

namespace {

using namespace rpo_mlp_dev;
using rpo_cart_dev::ActArgs;
using rpo_cart_dev::CartConsts;
using rpo_cart_dev::cart_explore_project;
using rpo_cart_dev::load_consts;

Mlp to_dev(const rpo_mlp* h) {
    return Mlp{h->Ws, h->bs, h->Wa, h->ba, h->W0, h->b0, h->W1, h->b1, h->W1b, h->b1b, h->S, h->A, h->E, h->H, h->n_out, h->cat, h->head_dim};
}

// ------------------------------------------------------------------------------------------------------ rollout
// (RolloutArgs<ENV>: rollout_args.h)

#ifndef RPO_ROLLOUT_SKIP
#define RPO_ROLLOUT_SKIP 0         // timing-only builds (tools/probe/build_stream_variants.sh KIND=rollout): 1 no actor MLP, 2 no
#endif                             // projection, 4 no env step / ring row, 8 no Philox draw, 16 no statistics / clock epilogue

template <class ENV, int EIN, int H, int RT>
__global__ __launch_bounds__(kFwdThreads) void rollout_kernel(RolloutArgs<ENV> p, typename ENV::Consts c) {
    typedef TileLds<EIN, RT, 8, 8> Lds;                          // 16 * RT lanes per workgroup; OBS <= 8
    __shared__ Lds lds;
    constexpr int kInS = Lds::kS;
    constexpr int kLanes = rpo_mlp_dev::kRows * RT;
    constexpr int kStats = 10;
    const int row0 = blockIdx.x * kLanes;
    const int tid = threadIdx.x;
    const int n = p.step.n;
    const long long t = p.step.ctrl ? p.step.ctrl[RPO_CTRL_T] : 0;
    RpoEpisode ep{0, 0.0f, 0u};                                  // requested before the MLP: needed only at the very end
    if (tid < kLanes && row0 + tid < n) ep = ENV::episode(p.step, row0 + tid);
    ENV::stage_obs(p.step, row0, kLanes, lds.in_s, kInS);
    // (Round 5, measured and rejected: every weight request in front of the staging -- the staging's wait then covers them all,
    //  12.4 -> 13.0 us --, and the observation requested first, the weights next, the staging waiting for the observation alone:
    //  12.7 us.  The order below stays.)
    // The lane's N(0,1) draw (exploration noise / rsample) depends on nothing the MLP produces: ~160 dependent VALU
    // instructions that would otherwise head the per-lane chain behind the MLP run in the shadow of its weight loads.
    float draw = 0.0f;
    const bool philox_noise = !p.gauss && p.act.noise_mode == RPO_NOISE_PHILOX;
    if (!(RPO_ROLLOUT_SKIP & 8) && tid < kLanes && row0 + tid < n && (p.gauss || philox_noise)) {
        const rpo_u4 u = p.gauss ? rpo_philox(p.act.seed, p.act.env_id_base + (uint32_t)(row0 + tid), (uint32_t)t, RPO_STREAM_POLICY,
                                              (uint32_t)p.step.ctrl[RPO_CTRL_UPDATES])
                                 : rpo_philox(p.act.seed, p.act.env_id_base + (uint32_t)(row0 + tid), (uint32_t)t, RPO_STREAM_ACT);
        draw = rpo_normal(u.x, u.y);
    }
    if (RPO_ROLLOUT_SKIP & 1) {
        if (tid < 2 * kLanes) lds.out[tid] = 0.25f;
        __syncthreads();
    } else {
        mlp_tile_forward<EIN, H, RT, Lds>(p.actor, lds, row0, n, nullptr, nullptr, p.gauss ? 0 : 1, p.scale, p.base);
    }

    float st[kStats];
#pragma unroll
    for (int k = 0; k < kStats; ++k) st[k] = 0.0f;
    float iters_f = 0.0f;
    const int i = row0 + tid;
    if (tid < kLanes && i < n) {
        const float eps_t = fmaxf(p.act.eps_end, p.act.eps_start - p.act.eps_decay * (float)t);
        float ap = lds.out[tid * 2];
        if (p.gauss) {
            // rsample of the squashed Gaussian (model/policy.py:58-66) with the Philox stream of the policy draw,
            // then the box clip of take_action (agent/sac_pa.py:111); p.act.noise_mode is RPO_NOISE_NONE
            ap = rpo_head_dev::gauss_head_row(ap, lds.out[tid * 2 + 1], draw, p.scale, p.base, p.act.box_lo, p.act.box_hi, 0, nullptr);
        }
        int k;
        typename ENV::ActArgs act = p.act;
        if (philox_noise) {                                      // the draw above, applied with the arithmetic of *_explore_project
            ap = rpo_explore_clip(ap, eps_t, draw, p.act.box_lo, p.act.box_hi);
            act.noise_mode = RPO_NOISE_NONE;
        }
        float2 a;
        if (RPO_ROLLOUT_SKIP & 2) { a = make_float2(ap, 0.5f * ap); k = 0; }
        else a = ENV::project(act, c, lds.in_s + tid * kInS, i, ap, eps_t, t, k);
        iters_f = (float)k;
        reinterpret_cast<float2*>(p.act.action)[i] = a;
        const long long ring_base = p.step.rows ? (t % p.step.cap_steps) * (long long)n : 0;
        if (!(RPO_ROLLOUT_SKIP & 4)) ENV::lane(p.step, c, i, lds.in_s + tid * kInS, a, ep, ring_base, st);
    }
    if (RPO_ROLLOUT_SKIP & 16) return;
    if (p.step.stats && tid < 64) {                            // only wave 0 holds data: wave reduction, lane 0 adds
        float* srow = rpo_stats_row(p.step.stats, p.step.stats_cap, t);
        const int slot[kStats + 1] = {RPO_STAT_REWARD_SUM, RPO_STAT_EPISODES, RPO_STAT_RETURN_SUM, RPO_STAT_LENGTH_SUM,
                                      RPO_STAT_MAX_INEQ_SUM, RPO_STAT_MAX_EQ_SUM, RPO_STAT_VIOL_COUNT, RPO_STAT_TERMINATED,
                                      RPO_STAT_MAX_INEQ_MAX, RPO_STAT_MAX_EQ_MAX, RPO_STAT_PROJ_ITERS};
        float red[kStats + 1];
#pragma unroll
        for (int k = 0; k < kStats; ++k) red[k] = st[k];
        red[kStats] = iters_f;
        if (kLanes <= 16) rpo_row16_reduce_many(red, 3u << 8);  // (only lanes 0..15 hold data: DPP, no LDS round trips)
        else rpo_wave_reduce_many<kStats + 1, 32>(red, 3u << 8);
        rpo_stats_commit(red, 3u << 8, slot, srow);
    }
    if (!p.defer_clock) rpo_step_epilogue(p.step.ctrl, t, p.step.stats, p.step.stats_cap);
}

template <class ENV>
int launch_rollout(const RolloutArgs<ENV>& args, const typename ENV::Consts& c, int n_envs, void* stream) {
    // RPO_TUNE_ROLLOUT_WIDE: 0 / 1 force the 16- / 64-lane row tiles, 3 the streaming form (tests), 2 (default) by size:
    // 16-lane tiles below 64 x 192 lanes, 64-lane tiles up to RPO_ROLLOUT_STREAM_FROM, the streaming form from there
    {
        const int sel = rpo_tune(RPO_TUNE_ROLLOUT_WIDE);
        if (sel == 3 || sel == 4 || (sel == 2 && n_envs >= RPO_ROLLOUT_STREAM_FROM)) {   // (4: 64-lane groups forced, tests)
            const int e = rpo_rollout_stream_launch(std::is_same<ENV, CartEnv>::value ? 0 : 1, &args, &c, n_envs, stream);
            if (e >= 0) return e;                                // (-1: the streaming form does not apply to this network / env)
        }
    }
    // 64 lanes per workgroup once that still fills the chip: the 128 KB hidden-layer matrix is streamed once per
    // workgroup, so wider tiles cut the L2 traffic 4x (rpo_tuning(RPO_TUNE_ROLLOUT_WIDE, 0 / 1) overrides the size rule)
    const int wide_sel = rpo_tune(RPO_TUNE_ROLLOUT_WIDE);        // 0 / 1: forced (tests), 2: by size
    const int E = args.actor.E;
    const bool wide = E == 128 && (wide_sel == 0 || wide_sel == 1 ? wide_sel != 0 : n_envs >= 64 * 192);
    if (E == 128 && wide) {
        hipLaunchKernelGGL((rollout_kernel<ENV, 128, 256, 4>), dim3((n_envs + 63) / 64), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, args, c);
    } else if (E == 128) {
        hipLaunchKernelGGL((rollout_kernel<ENV, 128, 256, 1>), dim3((n_envs + 15) / 16), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, args, c);
    } else if (E == 256) {
        hipLaunchKernelGGL((rollout_kernel<ENV, 256, 256, 1>), dim3((n_envs + 15) / 16), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, args, c);
    } else {
        return RPO_ERR_ARG;
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int check_actor(const Mlp& actor, int obs_dim, int gauss) {
    if (actor.hd > 1 || actor.S != obs_dim || actor.A != 0 || actor.n_out != (gauss ? 2 : 1) || actor.cat || actor.H != 256 || !actor.Ws ||
        !actor.W0 || !actor.W1 || (gauss && !actor.W1b))
        return RPO_ERR_ARG;
    return 0;
}

// ----------------------------------------------------------------------------------------------- critic forward
struct CriticFwdArgs {
    Mlp actor_target, critic_target, critic;
    float scale, base;
    const float* rows;            // replay ring
    long long cap_steps;
    int n_envs;
    int batch;
    float* batch_out;             // [B, 24] gathered rows (the backward pass reads s, a from it)
    long long* idx_out;           // [B] or NULL
    const long long* idx_in;      // [B] or NULL: caller-provided indices instead of the Philox draw (tests)
    uint64_t seed;
    uint32_t salt;
    const long long* ctrl;
    int max_steps; float corr_lr, corr_eps, corr_momentum, box_lo, box_hi;
    float* q_out;                 // [B] Q(s, a)
    float* qn_out;                // [B] Q_targ(s', a')
    float* x0_save; float* h1_save;   // critic pre-activations for rpo_mlp_backward
};

template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void cart_ddpg_critic_forward_kernel(CriticFwdArgs p, CartConsts c) {
    __shared__ TileLds<EIN> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * 6];      // the 16 sampled transition rows
    const int row0 = blockIdx.x * kRows;
    const int tid = threadIdx.x;
    const int B = p.batch;
    const long long t = p.ctrl[RPO_CTRL_T];
    // ---- ReplayBuffer.sample (buffer.py:31-34): counter-based draw + gather, 6 float4 chunks per row
    if (tid < kRows * 6) {
        const int r = tid / 6, ch = tid - r * 6;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row0 + r < B) {
            long long row;
            if (p.idx_in) {
                row = p.idx_in[row0 + r];
            } else {
                const unsigned long long n_valid = (unsigned long long)((t < p.cap_steps ? t : p.cap_steps) * (long long)p.n_envs);
                const rpo_u4 u = rpo_philox(p.seed, (uint32_t)(row0 + r), (uint32_t)t + p.salt, RPO_STREAM_SAMPLE,
                                            (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                row = (long long)__umul64hi(((unsigned long long)u.x << 32) | u.y, n_valid);
            }
            v = reinterpret_cast<const float4*>(p.rows)[row * (RPO_CART_RING / 4) + ch];
            if (blockIdx.y == 0) {                             // both roles draw the same rows; one publishes them
                reinterpret_cast<float4*>(p.batch_out)[(size_t)(row0 + r) * 6 + ch] = v;
                if (p.idx_out && ch == 0) p.idx_out[row0 + r] = row;
            }
        }
        tile[tid] = v;
    }
    __syncthreads();
    const float* tf = reinterpret_cast<const float*>(tile);    // row r: s[0:6] a[6:8] s'[8:14] r[14] done[15] ...
    if (blockIdx.y == 0) {
        // ---- target role: pi_targ(s') (deterministic, rpo_ddpg.py:329)
        if (tid < kRows * 6) {
            const int r = tid / 6, i = tid - r * 6;
            lds.in_s[r * kInS + i] = tf[r * 24 + 8 + i];       // next_state
        }
        mlp_tile_forward<EIN, H>(p.actor_target, lds, row0, B, nullptr, nullptr, 1, p.scale, p.base);
        // ---- Complete + Proj (rpo_ddpg.py:330): per-row stop test == the reference's batched call for this env
        if (tid < kRows) {
            ActArgs a{};
            a.noise_mode = RPO_NOISE_NONE;
            a.max_steps = p.max_steps; a.corr_lr = p.corr_lr; a.corr_eps = p.corr_eps; a.corr_momentum = p.corr_momentum;
            a.box_lo = p.box_lo; a.box_hi = p.box_hi;
            int k;
            const float2 act = cart_explore_project(a, c, row0 + tid, lds.out[tid * 2], 0.0f, t, k);
            lds.in_a[tid * kInA] = act.x;
            lds.in_a[tid * kInA + 1] = act.y;
        }
        // ---- Q_targ(s', a')
        mlp_tile_forward<EIN, H>(p.critic_target, lds, row0, B, nullptr, nullptr, 0, 1.0f, 0.0f);
        if (tid < kRows && row0 + tid < B) p.qn_out[row0 + tid] = lds.out[tid * 2];
    } else {
        // ---- critic role: Q(s, a), pre-activations kept for the backward pass
        if (tid < kRows * 6) {
            const int r = tid / 6, i = tid - r * 6;
            lds.in_s[r * kInS + i] = tf[r * 24 + i];           // state
        }
        if (tid < kRows * 2) {
            const int r = tid >> 1, i = tid & 1;
            lds.in_a[r * kInA + i] = tf[r * 24 + 6 + i];       // stored action
        }
        mlp_tile_forward<EIN, H>(p.critic, lds, row0, B, p.x0_save, p.h1_save, 0, 1.0f, 0.0f);
        if (tid < kRows && row0 + tid < B) p.q_out[row0 + tid] = lds.out[tid * 2];
    }
}

// ------------------------------------------------------------------------------------- critic forward, RPOSAC
// ReplayBuffer.sample -> a' ~ pi(s') (squashed Gaussian, online actor; rpo_sac.py:344-345) -> Complete + Proj ->
// min(Q1_targ, Q2_targ)(s', a') - alpha log pi -> Q1(s, a), Q2(s, a) -> TD target + both Huber losses and dLoss/dQ_k
// (rpo_sac.py:342-353).  CartSafe: all five MLP tiles chained in one workgroup of 16 rows.  SpringPendulum: the
// reference's batched projection couples the samples of a batch (pendulum.py:337-339), so the chain is cut
// there: front (sample -> policy) | rpo_pendulum_project_batchref | back (target critics -> critics -> TD).
struct CartRow { static constexpr int ROW = 24, CH = 6, RCH = RPO_CART_RING / 4, S = 6, A_OFF = 6, NS_OFF = 8; };
struct PendRow { static constexpr int ROW = 16, CH = 4, RCH = RPO_PEND_RING / 4, S = 5, A_OFF = 5, NS_OFF = 7; };

struct SacCriticFwdArgs {
    Mlp actor, critic_target1, critic_target2, critic1, critic2;
    float scale, base;
    const float* rows;
    long long cap_steps;
    int n_envs;
    int batch;
    float* batch_out;             // [B, ROW] gathered rows (front writes, back / backward pass read)
    long long* idx_out;
    const long long* idx_in;      // [B] or NULL (tests)
    const float* eps_in;          // [B] or NULL: caller-provided N(0,1) draws of rsample instead of Philox (tests)
    uint64_t seed, sample_seed;
    uint32_t salt, noise_salt, noise_id_base;
    const long long* ctrl;
    int max_steps; float corr_lr, corr_eps, corr_momentum, box_lo, box_hi;
    float* q1_out; float* q2_out;       // [B] Q1(s, a), Q2(s, a)   (RPODDPG back: q1_out only)
    float* qn1_out; float* qn2_out;     // [B] target critics at (s', a')
    float* x0_save1; float* h1_save1; float* x0_save2; float* h1_save2;
    float* ap_out; float* logp_out;                 // split form, front: clipped basic action and log pi of a'
    const float* next_actions; const float* logp_in;   // split form, back: projected a' [B,2] and log pi
};

// ReplayBuffer.sample (buffer.py:31-34): counter-based draw + gather of the workgroup's 16 rows into `tile` / batch_out
template <class L>
__device__ __forceinline__ void sac_sample(const SacCriticFwdArgs& p, float4* tile, int row0, long long t) {
    const int tid = threadIdx.x, B = p.batch;
    if (tid < kRows * L::CH) {
        const int r = tid / L::CH, ch = tid - r * L::CH;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row0 + r < B) {
            long long row;
            if (p.idx_in) {
                row = p.idx_in[row0 + r];
            } else {
                const unsigned long long n_valid = (unsigned long long)((t < p.cap_steps ? t : p.cap_steps) * (long long)p.n_envs);
                const rpo_u4 u = rpo_philox(p.sample_seed, (uint32_t)(row0 + r), (uint32_t)t + p.salt, RPO_STREAM_SAMPLE,
                                            (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                row = (long long)__umul64hi(((unsigned long long)u.x << 32) | u.y, n_valid);
            }
            v = reinterpret_cast<const float4*>(p.rows)[row * L::RCH + ch];
            if (blockIdx.y == 0) {                             // every role draws the same rows; one publishes them
                reinterpret_cast<float4*>(p.batch_out)[(size_t)(row0 + r) * L::CH + ch] = v;
                if (p.idx_out && ch == 0) p.idx_out[row0 + r] = row;
            }
        }
        tile[tid] = v;
    }
    __syncthreads();
}

// a' ~ pi(s'): mean / log-std heads, rsample with the Philox draw of the update step, box clip.  Threads < 16 return the
// clipped basic action of their row and its log-probability.  Leaves s' staged in lds.in_s.
template <class L, int EIN, int H>
__device__ __forceinline__ float sac_policy(const SacCriticFwdArgs& p, TileLds<EIN>& lds, const float* tf, int row0,
                                            long long t, float& logp) {
    const int tid = threadIdx.x, B = p.batch;
    if (tid < kRows * L::S) {
        const int r = tid / L::S, i = tid - r * L::S;
        lds.in_s[r * kInS + i] = tf[r * L::ROW + L::NS_OFF + i];
    }
    mlp_tile_forward<EIN, H>(p.actor, lds, row0, B, nullptr, nullptr, 0, 1.0f, 0.0f);
    float ap = 0.0f;
    logp = 0.0f;
    if (tid < kRows) {
        float e;
        if (p.eps_in) {
            e = row0 + tid < B ? p.eps_in[row0 + tid] : 0.0f;
        } else {                                               // == rpo_philox_normal(id_base, salt, RPO_STREAM_POLICY)
            const rpo_u4 u = rpo_philox(p.seed, p.noise_id_base + (uint32_t)(row0 + tid), (uint32_t)t + p.noise_salt,
                                        RPO_STREAM_POLICY, (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
            e = rpo_normal(u.x, u.y);
        }
        ap = rpo_head_dev::gauss_head_row(lds.out[tid * 2], lds.out[tid * 2 + 1], e, p.scale, p.base, p.box_lo, p.box_hi, 0,
                                          &logp);
    }
    return ap;
}

// One target critic on (s', a') staged in lds.in_s / lds.in_a  -> qn_out
template <int EIN, int H>
__device__ __forceinline__ void target_role(const Mlp& net, TileLds<EIN>& lds, int row0, int B, float* qn_out) {
    mlp_tile_forward<EIN, H>(net, lds, row0, B, nullptr, nullptr, 0, 1.0f, 0.0f);
    if (threadIdx.x < kRows && row0 + threadIdx.x < B) qn_out[row0 + threadIdx.x] = lds.out[threadIdx.x * 2];
}

// One critic on the stored (s, a) of the tile, pre-activations kept for the backward pass  -> q_out
template <class L, int EIN, int H>
__device__ __forceinline__ void critic_role(const Mlp& net, TileLds<EIN>& lds, const float* tf, int row0, int B, float* q_out,
                                            float* x0_save, float* h1_save) {
    const int tid = threadIdx.x;
    if (tid < kRows * L::S) {
        const int r = tid / L::S, i = tid - r * L::S;
        lds.in_s[r * kInS + i] = tf[r * L::ROW + i];           // state
    }
    if (tid < kRows * 2) {
        const int r = tid >> 1, i = tid & 1;
        lds.in_a[r * kInA + i] = tf[r * L::ROW + L::A_OFF + i];   // stored action
    }
    mlp_tile_forward<EIN, H>(net, lds, row0, B, x0_save, h1_save, 0, 1.0f, 0.0f);
    if (tid < kRows && row0 + tid < B) q_out[row0 + tid] = lds.out[tid * 2];
}

// gridDim.y = 4 roles per 16-row tile, independent of each other (their results meet in the TD prologue of the backward
// pass, rpo_td): 0 / 1 = a' ~ pi(s') -> Complete + Proj -> Q1_targ / Q2_targ (role 0 also publishes the batch and log pi),
// 2 / 3 = Q1 / Q2 on the stored (s, a).
template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void cart_sac_critic_forward_kernel(SacCriticFwdArgs p, CartConsts c) {
    __shared__ TileLds<EIN> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * CartRow::CH];
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, role = blockIdx.y;
    const long long t = p.ctrl[RPO_CTRL_T];
    sac_sample<CartRow>(p, tile, row0, t);
    const float* tf = reinterpret_cast<const float*>(tile);
    if (role >= 2) {
        if (role == 2) critic_role<CartRow, EIN, H>(p.critic1, lds, tf, row0, p.batch, p.q1_out, p.x0_save1, p.h1_save1);
        else critic_role<CartRow, EIN, H>(p.critic2, lds, tf, row0, p.batch, p.q2_out, p.x0_save2, p.h1_save2);
        return;
    }
    float logp;
    const float ap = sac_policy<CartRow, EIN, H>(p, lds, tf, row0, t, logp);
    if (tid < kRows) {                                         // Complete + Proj: per-row stop test == the batched call
        ActArgs a{};
        a.noise_mode = RPO_NOISE_NONE;
        a.max_steps = p.max_steps; a.corr_lr = p.corr_lr; a.corr_eps = p.corr_eps; a.corr_momentum = p.corr_momentum;
        a.box_lo = p.box_lo; a.box_hi = p.box_hi;
        int k;
        const float2 act = cart_explore_project(a, c, row0 + tid, ap, 0.0f, t, k);
        lds.in_a[tid * kInA] = act.x;
        lds.in_a[tid * kInA + 1] = act.y;
        if (role == 0 && row0 + tid < p.batch) p.logp_out[row0 + tid] = logp;
    }
    if (role == 0) target_role<EIN, H>(p.critic_target1, lds, row0, p.batch, p.qn1_out);
    else target_role<EIN, H>(p.critic_target2, lds, row0, p.batch, p.qn2_out);
}

template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void pend_sac_critic_front_kernel(SacCriticFwdArgs p) {
    __shared__ TileLds<EIN> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * PendRow::CH];
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x;
    const long long t = p.ctrl[RPO_CTRL_T];
    sac_sample<PendRow>(p, tile, row0, t);
    float logp;
    const float ap = sac_policy<PendRow, EIN, H>(p, lds, reinterpret_cast<const float*>(tile), row0, t, logp);
    if (tid < kRows && row0 + tid < p.batch) {
        p.ap_out[row0 + tid] = ap;
        p.logp_out[row0 + tid] = logp;
    }
}

// Stage the projected a' and s' of the tile for a target critic
template <int EIN>
__device__ __forceinline__ void stage_next(const SacCriticFwdArgs& p, TileLds<EIN>& lds, const float* tf, int row0) {
    const int tid = threadIdx.x, B = p.batch;
    if (tid < kRows * PendRow::S) {
        const int r = tid / PendRow::S, i = tid - r * PendRow::S;
        lds.in_s[r * kInS + i] = tf[r * PendRow::ROW + PendRow::NS_OFF + i];
    }
    if (tid < kRows) {
        const bool live = row0 + tid < B;
        lds.in_a[tid * kInA] = live ? p.next_actions[(size_t)(row0 + tid) * 2] : 0.0f;
        lds.in_a[tid * kInA + 1] = live ? p.next_actions[(size_t)(row0 + tid) * 2 + 1] : 0.0f;
    }
}

// back: gridDim.y = 4 independent roles per tile (Q1_targ | Q2_targ on (s', a'); Q1 | Q2 on (s, a)); RPODDPG: 2 roles
// (Q_targ | Q).  The TD target / Huber loss is the prologue of the backward pass (rpo_td).
template <int EIN, int H, int TWIN>
__global__ __launch_bounds__(kFwdThreads) void pend_critic_back_kernel(SacCriticFwdArgs p) {
    __shared__ TileLds<EIN> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * PendRow::CH];
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, B = p.batch;
    const int role = TWIN ? blockIdx.y : (blockIdx.y == 0 ? 0 : 2);
    if (tid < kRows * PendRow::CH) {
        const int r = tid / PendRow::CH;
        tile[tid] = row0 + r < B ? reinterpret_cast<const float4*>(p.batch_out)[(size_t)row0 * PendRow::CH + tid]
                                 : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    const float* tf = reinterpret_cast<const float*>(tile);
    if (role < 2) {
        stage_next<EIN>(p, lds, tf, row0);
        if (role == 0) target_role<EIN, H>(p.critic_target1, lds, row0, B, p.qn1_out);
        else target_role<EIN, H>(p.critic_target2, lds, row0, B, p.qn2_out);
    } else if (role == 2) {
        critic_role<PendRow, EIN, H>(p.critic1, lds, tf, row0, B, p.q1_out, p.x0_save1, p.h1_save1);
    } else {
        critic_role<PendRow, EIN, H>(p.critic2, lds, tf, row0, B, p.q2_out, p.x0_save2, p.h1_save2);
    }
}

template <int EIN, int H>
__global__ __launch_bounds__(kThreads) void actor_weights_kernel(BwdArgs p) {
    gradmax_flush(p.gradmax, mlp_bwd_weights_body<EIN, H>(p));
}

// RPODDPG critic forward on SpringPendulum, cut at the batch-coupled projection like the SAC form:
//   front: sample -> pi_targ(s') (tanh box)                       -> ap_out
//   back:  Q_targ(s', next_actions) | Q(s, a) saved  (pend_critic_back_kernel<.., 0>; rpo_ddpg.py:327-337)
template <int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void pend_ddpg_critic_front_kernel(SacCriticFwdArgs p) {
    __shared__ TileLds<EIN> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * PendRow::CH];
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x;
    const long long t = p.ctrl[RPO_CTRL_T];
    sac_sample<PendRow>(p, tile, row0, t);
    const float* tf = reinterpret_cast<const float*>(tile);
    if (tid < kRows * PendRow::S) {
        const int r = tid / PendRow::S, i = tid - r * PendRow::S;
        lds.in_s[r * kInS + i] = tf[r * PendRow::ROW + PendRow::NS_OFF + i];
    }
    mlp_tile_forward<EIN, H>(p.actor, lds, row0, p.batch, nullptr, nullptr, 1, p.scale, p.base);
    if (tid < kRows && row0 + tid < p.batch) p.ap_out[row0 + tid] = lds.out[tid * 2];
}

// Env policies of the actor-update pipelines: Complete, the Lagrangian row and the backward of Complete.
struct CartActEnv {
    typedef CartConsts Consts;
    static constexpr int ROW = RPO_CART_ROW, S = 6, NI = 6;
    __device__ static __forceinline__ float2 complete(const Consts& c, const float* obs, int i, float ap, long long t) {
        ActArgs a{};
        a.noise_mode = RPO_NOISE_NONE; a.max_steps = 0;
        int k;
        return cart_explore_project(a, c, i, ap, 0.0f, t, k);
    }
    // take_action's exploration: clip(ap_det + eps_t * noise[i]) (ddpg_pa.py:108-110), then Complete
    __device__ static __forceinline__ float2 complete_noisy(const Consts& c, const float* obs, int i, float ap_det, float eps_t,
                                                            const float* noise, float lo, float hi, long long t) {
        ActArgs a{};
        a.noise_mode = RPO_NOISE_EXPLICIT; a.noise = noise; a.box_lo = lo; a.box_hi = hi; a.max_steps = 0;
        int k;
        return cart_explore_project(a, c, i, ap_det, eps_t, t, k);
    }
    __device__ static __forceinline__ float lagr(const Consts& c, float a0, float a1, const float* nu_p, float scale,
                                                 float (&dist)[6], float2& g) {
        float nu[6], loss, g0, g1;
#pragma unroll
        for (int j = 0; j < 6; ++j) nu[j] = nu_p[j];
        rpo_cart_dev::lagrangian_row(c, a0, a1, nu, loss, dist, g0, g1);
        g = make_float2(scale * g0, scale * g1);
        return loss;
    }
    __device__ static __forceinline__ float complete_bwd(const Consts& c, const float* obs, float g0, float g1) {
        return rpo_cart_dev::complete_bwd_row(c, g0, g1);
    }
};

struct PendActEnv {
    struct Consts { int unused; };
    static constexpr int ROW = RPO_PEND_ROW, S = 5, NI = 1;
    __device__ static __forceinline__ float2 complete(const Consts&, const float* obs, int i, float ap, long long t) {
        rpo_pend_dev::ActArgs a{};
        a.noise_mode = RPO_NOISE_NONE; a.max_steps = 0;
        int k;
        return rpo_pend_dev::pend_explore_project(a, obs, i, ap, 0.0f, t, k);
    }
    __device__ static __forceinline__ float2 complete_noisy(const Consts&, const float* obs, int i, float ap_det, float eps_t,
                                                            const float* noise, float lo, float hi, long long t) {
        rpo_pend_dev::ActArgs a{};
        a.noise_mode = RPO_NOISE_EXPLICIT; a.noise = noise; a.box_lo = lo; a.box_hi = hi; a.max_steps = 0;
        int k;
        return rpo_pend_dev::pend_explore_project(a, obs, i, ap_det, eps_t, t, k);
    }
    __device__ static __forceinline__ float lagr(const Consts&, float a0, float a1, const float* nu_p, float scale,
                                                 float (&dist)[6], float2& g) {
#pragma unroll
        for (int j = 1; j < 6; ++j) dist[j] = 0.0f;
        return rpo_pend_dev::lagrangian_row(a0, a1, nu_p[0], scale, dist[0], g.x, g.y);
    }
    __device__ static __forceinline__ float complete_bwd(const Consts&, const float* obs, float g0, float g1) {
        return rpo_pend_dev::complete_bwd_row(obs, g0, g1);
    }
};

// --------------------------------------------------------------------------------------- actor update, RPODDPG
// The policy step of rpo_ddpg.py:186-205,307-324 on CartSafe in two launches + the weights pass:
//   forward   pi(s) (pre-activations saved) -> exploration noise + clip (take_action) -> Complete -> Q(s, a) (saved) ->
//             nu . relu(g(a)): value, d/d action, d/d nu partials; dq = -1/B
//   backward  critic rows (d/d a of -Q) + d/d a of the Lagrangian -> autograd through Complete -> tanh box / clip ->
//             actor rows; with a shared state embedding the critic's dx0 is added to the actor's, so that ONE
//             first-layer reduction yields the embedding's gradient; workgroup 0 folds the Lagrangian partials.
// == rpo_mlp_forward x 2, rpo_philox_normal, rpo_cartsafe_act_project, rpo_cartsafe_lagrangian, a fill, rpo_mlp_backward
//    rows x 2 (+ the critic's first-layer weights pass), an add, rpo_cartsafe_complete_bwd, rpo_tanh_box_bwd.
struct ActorFwdArgs {
    Mlp actor, critic;
    float scale, base, box_lo, box_hi, eps_start, eps_end, eps_decay;
    const float* batch;           // [B, 24] gathered rows (state = columns 0..5)
    int B;
    const float* noise_in;        // [B] or NULL (tests): N(0,1) draws of take_action
    uint64_t seed;
    uint32_t noise_id_base, noise_salt;
    const long long* ctrl;
    const float* nu;              // [6]
    float* ap_det; float* noise_out; float* actions; float* q_out; float* dq_out; float* g_act;
    float* partial;               // [gridDim.x, 8]: sum nu.dist, sum dist_0..5, sum q
    float* ax0; float* ah1; float* cx0; float* ch1;
};

template <class ENV, int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void ddpg_actor_forward_kernel(ActorFwdArgs p, typename ENV::Consts c) {
    __shared__ TileLds<EIN> lds;
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, B = p.B;
    const long long t = p.ctrl[RPO_CTRL_T];
    if (tid < kRows * ENV::S) {
        const int r = tid / ENV::S, i = tid - r * ENV::S;
        lds.in_s[r * kInS + i] = row0 + r < B ? p.batch[(size_t)(row0 + r) * ENV::ROW + i] : 0.0f;
    }
    mlp_tile_forward<EIN, H>(p.actor, lds, row0, B, p.ax0, p.ah1, 1, p.scale, p.base);
    float vals[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float inv_b = 1.0f / (float)B;
    const bool live = tid < kRows && row0 + tid < B;
    if (tid < kRows) {
        float a0 = 0.0f, a1 = 0.0f;
        if (live) {
            const int i = row0 + tid;
            const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);
            const float ap_det = lds.out[tid * 2];
            float e;
            if (p.noise_in) {
                e = p.noise_in[i];
            } else {
                const rpo_u4 u = rpo_philox(p.seed, p.noise_id_base + (uint32_t)i, (uint32_t)t + p.noise_salt, RPO_STREAM_POLICY,
                                            (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                e = rpo_normal(u.x, u.y);
            }
            p.ap_det[i] = ap_det;
            p.noise_out[i] = e;
            const float2 act = ENV::complete_noisy(c, lds.in_s + tid * kInS, i, ap_det, eps_t, p.noise_out, p.box_lo, p.box_hi, t);
            a0 = act.x; a1 = act.y;
            reinterpret_cast<float2*>(p.actions)[i] = act;
            float dist[6];
            float2 g;
            vals[0] = ENV::lagr(c, a0, a1, p.nu, inv_b, dist, g);
#pragma unroll
            for (int j = 0; j < 6; ++j) vals[1 + j] = dist[j];
            reinterpret_cast<float2*>(p.g_act)[i] = g;
            p.dq_out[i] = -inv_b;                               // d mean(-Q) / dQ
        }
        lds.in_a[tid * kInA] = a0;
        lds.in_a[tid * kInA + 1] = a1;
    }
    mlp_tile_forward<EIN, H>(p.critic, lds, row0, B, p.cx0, p.ch1, 0, 1.0f, 0.0f);
    if (live) {
        vals[7] = lds.out[tid * 2];
        p.q_out[row0 + tid] = vals[7];
    }
    if (tid < 64) {
        rpo_wave_reduce_many<8>(vals, 0u);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (tid == 0) p.partial[blockIdx.x * 8 + k] = vals[k];
    }
}

struct ActorBwdArgs {
    BwdArgs critic, actor;
    const float* g_act; const float* ap_det; const float* noise;
    float eps_start, eps_end, eps_decay, box_lo, box_hi, scale, base;
    const long long* ctrl;
    float* dout;                  // [B] scratch: d loss / d (actor output)
    const float* partial; int n_parts;
    float* lag_out;               // [2]: mean Lagrangian term, mean Q
    float* nu_grad;               // [6] accumulated
    int shared_embedding;
};

template <class ENV, int EIN, int H>
__global__ __launch_bounds__(kThreads) void ddpg_actor_backward_kernel(ActorBwdArgs p, typename ENV::Consts c) {
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, B = p.critic.n;
    mlp_bwd_rows_body<EIN, H>(p.critic);                        // dh, dx0 of the critic and da = d(-Q)/d action
    __syncthreads();
    if (tid < kRows && row0 + tid < B) {
        const int i = row0 + tid;
        const float t = (float)p.ctrl[RPO_CTRL_T];
        const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * t);
        const float2 d = reinterpret_cast<const float2*>(p.critic.da)[i];
        const float2 g = reinterpret_cast<const float2*>(p.g_act)[i];
        const float da0 = d.x + g.x, da1 = d.y + g.y;
        reinterpret_cast<float2*>(p.critic.da)[i] = make_float2(da0, da1);
        const float dap = ENV::complete_bwd(c, p.actor.s + (size_t)i * p.actor.s_stride, da0, da1);
        p.dout[i] = rpo_head_dev::tanh_box_bwd_row(dap, p.ap_det[i], p.noise[i], 1, eps_t, p.box_lo, p.box_hi, p.scale, p.base);
    }
    __syncthreads();
    mlp_bwd_rows_body<EIN, H>(p.actor);                         // reads p.actor.dout == p.dout
    if (p.shared_embedding) {                                   // one first-layer reduction for the shared embedding
        __syncthreads();
        for (int idx = tid; idx < kRows * EIN; idx += kThreads) {
            const int r = idx / EIN, e = idx - r * EIN;
            if (row0 + r < B) p.actor.dx0[(size_t)(row0 + r) * EIN + e] += p.critic.dx0[(size_t)(row0 + r) * EIN + e];
        }
    }
    if (blockIdx.x == 0 && tid < 8) {                           // Lagrangian value / multiplier gradient, fixed order
        float sacc = 0.0f;
        for (int g = 0; g < p.n_parts; ++g) sacc += p.partial[g * 8 + tid];
        const float inv_b = 1.0f / (float)B;
        if (tid == 0) p.lag_out[0] = inv_b * sacc;
        else if (tid == 7) p.lag_out[1] = inv_b * sacc;
        else if (tid - 1 < ENV::NI) p.nu_grad[tid - 1] += inv_b * sacc;
    }
}

// ---------------------------------------------------------------------------------------- actor update, RPOSAC
// rpo_sac.py:191-219,321-339 on either env as forward + backward pipelines (the actor loss only completes the action, it
// does not project it, so nothing couples the rows of a batch):
//   forward   pi(s): mean / log-std heads (saved) -> rsample + box clip + log pi -> Complete -> Q1, Q2 (s, a) (saved) ->
//             nu . relu(g(a)); d(-min(Q1, Q2))/dQ_k with ties split like torch.min's backward
//   backward  both critics' rows -> d/d a -> Complete -> Gaussian head (incl. the alpha log pi term) -> actor rows
struct SacActorFwdArgs {
    Mlp actor, critic1, critic2;
    float scale, base, box_lo, box_hi, alpha;
    const float* batch; int B;
    const float* noise_in;
    uint64_t seed;
    uint32_t noise_id_base, noise_salt;
    const long long* ctrl;
    const float* nu;
    float* raw; float* noise_out; float* logp; float* actions; float* dq1; float* dq2; float* g_act;   // dq_k: Q_k here
    float* partial;               // [gridDim.x, 8]: sum nu.dist, sum dist_0..5 (column 7: the backward kernel's)
    float* ax0; float* ah1; float* c1x0; float* c1h1; float* c2x0; float* c2h1;
};

// gridDim.y = 2 roles per tile: both evaluate pi(s) -> head -> Complete (deterministic given the Philox counters); role k
// continues with critic k on the completed action and leaves Q_k in q_out_k.  Role 0 publishes the policy's outputs and
// the Lagrangian partials.  min(Q1, Q2), its gradient split and the loss term meet in the backward kernel's prologue.
template <class ENV, int EIN, int H>
__global__ __launch_bounds__(kFwdThreads) void sac_actor_forward_kernel(SacActorFwdArgs p, typename ENV::Consts c) {
    __shared__ TileLds<EIN> lds;
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, B = p.B, role = blockIdx.y;
    const long long t = p.ctrl[RPO_CTRL_T];
    if (tid < kRows * ENV::S) {
        const int r = tid / ENV::S, i = tid - r * ENV::S;
        lds.in_s[r * kInS + i] = row0 + r < B ? p.batch[(size_t)(row0 + r) * ENV::ROW + i] : 0.0f;
    }
    mlp_tile_forward<EIN, H>(p.actor, lds, row0, B, role == 0 ? p.ax0 : nullptr, role == 0 ? p.ah1 : nullptr, 0, 1.0f, 0.0f);
    float vals[7] = {0, 0, 0, 0, 0, 0, 0};
    const float inv_b = 1.0f / (float)B;
    const bool live = tid < kRows && row0 + tid < B;
    if (tid < kRows) {
        float2 act = make_float2(0.0f, 0.0f);
        if (live) {
            const int i = row0 + tid;
            const float rm = lds.out[tid * 2], rl = lds.out[tid * 2 + 1];
            float e;
            if (p.noise_in) {
                e = p.noise_in[i];
            } else {
                const rpo_u4 u = rpo_philox(p.seed, p.noise_id_base + (uint32_t)i, (uint32_t)t + p.noise_salt, RPO_STREAM_POLICY,
                                            (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                e = rpo_normal(u.x, u.y);
            }
            float logp = 0.0f;
            const float ap = rpo_head_dev::gauss_head_row(rm, rl, e, p.scale, p.base, p.box_lo, p.box_hi, 0, &logp);
            act = ENV::complete(c, lds.in_s + tid * kInS, i, ap, t);
            if (role == 0) {
                reinterpret_cast<float2*>(p.raw)[i] = make_float2(rm, rl);
                p.noise_out[i] = e;
                p.logp[i] = logp;
                reinterpret_cast<float2*>(p.actions)[i] = act;
                float dist[6];
                float2 g;
                vals[0] = ENV::lagr(c, act.x, act.y, p.nu, inv_b, dist, g);
#pragma unroll
                for (int j = 0; j < 6; ++j) vals[1 + j] = dist[j];
                reinterpret_cast<float2*>(p.g_act)[i] = g;
            }
        }
        lds.in_a[tid * kInA] = act.x;
        lds.in_a[tid * kInA + 1] = act.y;
    }
    if (role == 0) mlp_tile_forward<EIN, H>(p.critic1, lds, row0, B, p.c1x0, p.c1h1, 0, 1.0f, 0.0f);
    else mlp_tile_forward<EIN, H>(p.critic2, lds, row0, B, p.c2x0, p.c2h1, 0, 1.0f, 0.0f);
    if (live) (role == 0 ? p.dq1 : p.dq2)[row0 + tid] = lds.out[tid * 2];        // Q_k(s, a): turned into dLoss/dQ_k later
    if (role == 0 && tid < 64) {
        rpo_wave_reduce_many(vals, 0u);
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (tid == 0) p.partial[blockIdx.x * 8 + k] = vals[k];
    }
}

struct SacActorBwdArgs {
    BwdArgs critic1, critic2, actor;
    const float* g_act; const float* raw; const float* noise;
    float dlogp, box_lo, box_hi, scale, base;   // dlogp: coefficient of log pi in the loss / B (alpha / B)
    float* dout;                  // [B, 2] scratch: d loss / d (mean head, log-std head)
    float* partial; int n_parts;  // [n_parts, 8]: columns 0..6 from the forward kernel; column 7 written here
    const float* logp;            // [B] log pi of the sampled actions
    float* q_dq1; float* q_dq2;   // [B] in: Q_k(s, a) from the forward kernel; out: dLoss/dQ_k (the critics' dout)
    float* lag_out;               // [1]: mean Lagrangian term
    float* nu_grad; int n_ineq;
    int shared_embedding;
};

template <class ENV, int EIN, int H>
__global__ __launch_bounds__(kThreads) void sac_actor_backward_kernel(SacActorBwdArgs p, typename ENV::Consts c) {
    const int row0 = blockIdx.x * kRows, tid = threadIdx.x, B = p.actor.n;
    // ---- where the two critic roles of the forward kernel meet: d(-min(q1, q2))/dq -- the smaller one takes the gradient,
    //      ties are split (torch.min's backward) -- and this tile's share of sum(alpha log pi - min Q)
    if (tid < 64) {
        float v = 0.0f;
        if (tid < kRows && row0 + tid < B) {
            const int i = row0 + tid;
            const float q1 = p.q_dq1[i], q2 = p.q_dq2[i];
            const float w1 = (q1 < q2 ? 1.0f : 0.0f) + (q1 == q2 ? 0.5f : 0.0f);
            const float inv_b = 1.0f / (float)B;
            p.q_dq1[i] = w1 * -inv_b;
            p.q_dq2[i] = (1.0f - w1) * -inv_b;
            v = p.dlogp * (float)B * p.logp[i] - fminf(q1, q2);
        }
        const float sum = rpo_wave_sum(v);
        if (tid == 0) p.partial[blockIdx.x * 8 + 7] = sum;
    }
    __syncthreads();
    mlp_bwd_rows_body<EIN, H>(p.critic1);
    __syncthreads();
    mlp_bwd_rows_body<EIN, H>(p.critic2);
    __syncthreads();
    if (tid < kRows && row0 + tid < B) {
        const int i = row0 + tid;
        const float2 d1 = reinterpret_cast<const float2*>(p.critic1.da)[i], d2 = reinterpret_cast<const float2*>(p.critic2.da)[i];
        const float2 g = reinterpret_cast<const float2*>(p.g_act)[i];
        const float da0 = (d1.x + d2.x) + g.x, da1 = (d1.y + d2.y) + g.y;      // da1.add_(da2).add_(g_act)
        const float dap = ENV::complete_bwd(c, p.actor.s + (size_t)i * p.actor.s_stride, da0, da1);
        const float2 r = reinterpret_cast<const float2*>(p.raw)[i];
        reinterpret_cast<float2*>(p.dout)[i] = rpo_head_dev::gauss_head_bwd_row(r.x, r.y, p.noise[i], dap, p.dlogp, p.scale,
                                                                                p.base, p.box_lo, p.box_hi);
    }
    __syncthreads();
    mlp_bwd_rows_body<EIN, H>(p.actor);
    if (p.shared_embedding) {
        __syncthreads();
        for (int idx = tid; idx < kRows * EIN; idx += kThreads) {
            const int r = idx / EIN, e = idx - r * EIN;
            if (row0 + r < B) {
                const size_t o = (size_t)(row0 + r) * EIN + e;
                p.actor.dx0[o] += p.critic1.dx0[o] + p.critic2.dx0[o];
            }
        }
    }
    if (blockIdx.x == 0 && tid < 7) {                             // (column 7 is being written by the other tiles right now)
        float sacc = 0.0f;
        for (int g = 0; g < p.n_parts; ++g) sacc += p.partial[g * 8 + tid];
        const float inv_b = 1.0f / (float)B;
        if (tid == 0) p.lag_out[0] = inv_b * sacc;
        else if (tid - 1 < p.n_ineq) p.nu_grad[tid - 1] += inv_b * sacc;
    }
}

}  // namespace

extern "C" {

int rpo_cartsafe_rollout(const rpo_mlp* actor_host, int gauss, float scale, float base, int n_envs, float* state,
                         float* action, int* ep_len, float* ep_ret, unsigned* ep_count, float* rows, long long cap_steps,
                         float* stats, int stats_cap, long long* ctrl, int noise_mode, float eps_start, float eps_end,
                         float eps_decay, float box_lo, float box_hi, int max_steps, float corr_lr, float corr_eps,
                         float corr_momentum, const float* consts_host, int partial, int max_episode_steps,
                         int auto_reset, float viol_thresh, unsigned long long seed, unsigned env_id_base, int defer_clock,
                         void* stream) {
    if (!actor_host) return RPO_ERR_NULL;
    if (n_envs <= 0 || max_episode_steps <= 0 || max_steps < 0) return RPO_ERR_ARG;
    if (noise_mode != RPO_NOISE_NONE && noise_mode != RPO_NOISE_PHILOX && noise_mode != RPO_NOISE_CLIP_ONLY) return RPO_ERR_ARG;
    if (!state || !action || !ep_len || !ep_ret || !ep_count || !ctrl) return RPO_ERR_NULL;
    if ((rows && cap_steps <= 0) || (stats && stats_cap <= 0)) return RPO_ERR_ARG;
    RolloutArgs<CartEnv> args{};
    args.actor = to_dev(actor_host);
    if (int e = check_actor(args.actor, 6, gauss)) return e;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    args.scale = scale; args.base = base; args.gauss = gauss; args.defer_clock = defer_clock ? 1 : 0;
    args.act = rpo_cart_dev::ActArgs{n_envs, nullptr, nullptr, action, nullptr, noise_mode, eps_start, eps_end, eps_decay,
                                     box_lo, box_hi, max_steps, corr_lr, corr_eps, corr_momentum, (uint64_t)seed,
                                     (uint32_t)env_id_base, ctrl, stats, stats_cap};
    args.step = rpo_cart_dev::StepArgs{n_envs, state, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, stats_cap,
                                       ctrl, max_episode_steps, auto_reset, viol_thresh, (uint64_t)seed,
                                       (uint32_t)env_id_base, 0};
    return launch_rollout<CartEnv>(args, c, n_envs, stream);
}

int rpo_pendulum_rollout(const rpo_mlp* actor_host, int gauss, float scale, float base, int n_envs, float* internal,
                         float* obs, float* action, int* ep_len, float* ep_ret, unsigned* ep_count, float* rows,
                         long long cap_steps, float* stats, int stats_cap, long long* ctrl, int noise_mode,
                         float eps_start, float eps_end, float eps_decay, float box_lo, float box_hi, int max_steps,
                         float corr_lr, float corr_eps, float corr_momentum, int max_episode_steps, int auto_reset,
                         float viol_thresh, unsigned long long seed, unsigned env_id_base, int defer_clock, void* stream) {
    if (!actor_host) return RPO_ERR_NULL;
    if (n_envs <= 0 || max_episode_steps <= 0 || max_steps < 0) return RPO_ERR_ARG;
    if (noise_mode != RPO_NOISE_NONE && noise_mode != RPO_NOISE_PHILOX && noise_mode != RPO_NOISE_CLIP_ONLY) return RPO_ERR_ARG;
    if (!internal || !action || !ep_len || !ep_ret || !ep_count || !ctrl) return RPO_ERR_NULL;
    if ((rows && cap_steps <= 0) || (stats && stats_cap <= 0)) return RPO_ERR_ARG;
    RolloutArgs<PendEnv> args{};
    args.actor = to_dev(actor_host);
    if (int e = check_actor(args.actor, 5, gauss)) return e;
    args.scale = scale; args.base = base; args.gauss = gauss; args.defer_clock = defer_clock ? 1 : 0;
    args.act = rpo_pend_dev::ActArgs{n_envs, nullptr, 5, nullptr, nullptr, action, nullptr, noise_mode, eps_start, eps_end,
                                     eps_decay, box_lo, box_hi, max_steps, corr_lr, corr_eps, corr_momentum, (uint64_t)seed,
                                     (uint32_t)env_id_base, ctrl, stats, stats_cap};
    args.step = rpo_pend_dev::StepArgs{n_envs, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats,
                                       stats_cap, ctrl, max_episode_steps, auto_reset, viol_thresh, (uint64_t)seed,
                                       (uint32_t)env_id_base};
    return launch_rollout<PendEnv>(args, PendEnv::Consts{0}, n_envs, stream);
}

int rpo_cartsafe_ddpg_critic_forward(const rpo_mlp* actor_target_host, const rpo_mlp* critic_target_host,
                                     const rpo_mlp* critic_host, float scale, float base, const float* rows,
                                     long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                     const long long* idx_in, unsigned long long seed, unsigned sample_salt,
                                     const long long* ctrl, int max_steps, float corr_lr, float corr_eps,
                                     float corr_momentum, float box_lo, float box_hi, const float* consts_host,
                                     int partial, float* q_out, float* qn_out, float* x0_save, float* h1_save,
                                     void* stream) {
    if (!actor_target_host || !critic_target_host || !critic_host) return RPO_ERR_NULL;
    if (batch <= 0 || cap_steps <= 0 || n_envs <= 0 || max_steps < 0) return RPO_ERR_ARG;
    if (!rows || !batch_out || !ctrl || !q_out || !qn_out || !x0_save || !h1_save) return RPO_ERR_NULL;
    CriticFwdArgs a{};
    a.actor_target = to_dev(actor_target_host); a.critic_target = to_dev(critic_target_host); a.critic = to_dev(critic_host);
    const Mlp& at = a.actor_target;
    const Mlp& ct = a.critic_target;
    const Mlp& cr = a.critic;
    if (at.S != 6 || at.A != 0 || at.n_out != 1 || at.cat || ct.S != 6 || ct.A != 2 || ct.cat || cr.S != 6 || cr.A != 2 || cr.cat ||
        at.H != 256 || ct.H != 256 || cr.H != 256 || at.E != ct.E || at.E != cr.E || ct.n_out != 1 || cr.n_out != 1)
        return RPO_ERR_ARG;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    a.scale = scale; a.base = base; a.rows = rows; a.cap_steps = cap_steps; a.n_envs = n_envs; a.batch = batch;
    a.batch_out = batch_out; a.idx_out = idx_out; a.idx_in = idx_in; a.seed = (uint64_t)seed; a.salt = (uint32_t)sample_salt;
    a.ctrl = ctrl; a.max_steps = max_steps; a.corr_lr = corr_lr; a.corr_eps = corr_eps; a.corr_momentum = corr_momentum;
    a.box_lo = box_lo; a.box_hi = box_hi; a.q_out = q_out; a.qn_out = qn_out; a.x0_save = x0_save; a.h1_save = h1_save;
    const int grid = (batch + kRows - 1) / kRows;                   // x: row tile, y: role (target chain | critic)
    if (at.E == 128) {
        hipLaunchKernelGGL((cart_ddpg_critic_forward_kernel<128, 256>), dim3(grid, 2), dim3(kFwdThreads), 0, (hipStream_t)stream, a, c);
    } else if (at.E == 256) {
        hipLaunchKernelGGL((cart_ddpg_critic_forward_kernel<256, 256>), dim3(grid, 2), dim3(kFwdThreads), 0, (hipStream_t)stream, a, c);
    } else {
        return RPO_ERR_ARG;
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_cartsafe_sac_critic_forward(const rpo_mlp* actor_host, const rpo_mlp* critic_target1_host,
                                    const rpo_mlp* critic_target2_host, const rpo_mlp* critic1_host,
                                    const rpo_mlp* critic2_host, float scale, float base, const float* rows,
                                    long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                    const long long* idx_in, const float* eps_in, unsigned long long sample_seed,
                                    unsigned sample_salt, unsigned long long noise_seed, unsigned noise_id_base,
                                    unsigned noise_salt, const long long* ctrl, int max_steps, float corr_lr,
                                    float corr_eps, float corr_momentum, float box_lo, float box_hi,
                                    const float* consts_host, int partial, float* q1_out, float* q2_out,
                                    float* qn1_out, float* qn2_out, float* logp_out, float* x0_save1, float* h1_save1,
                                    float* x0_save2, float* h1_save2, void* stream) {
    if (!actor_host || !critic_target1_host || !critic_target2_host || !critic1_host || !critic2_host) return RPO_ERR_NULL;
    if (batch <= 0 || cap_steps <= 0 || n_envs <= 0 || max_steps < 0) return RPO_ERR_ARG;
    if (!rows || !batch_out || !ctrl || !q1_out || !q2_out || !qn1_out || !qn2_out || !logp_out || !x0_save1 || !h1_save1 ||
        !x0_save2 || !h1_save2)
        return RPO_ERR_NULL;
    SacCriticFwdArgs a{};
    a.actor = to_dev(actor_host);
    a.critic_target1 = to_dev(critic_target1_host); a.critic_target2 = to_dev(critic_target2_host);
    a.critic1 = to_dev(critic1_host); a.critic2 = to_dev(critic2_host);
    if (int e = check_actor(a.actor, 6, 1)) return e;
    const Mlp* qs[4] = {&a.critic_target1, &a.critic_target2, &a.critic1, &a.critic2};
    for (const Mlp* q : qs)
        if (q->S != 6 || q->A != 2 || q->cat || q->H != 256 || q->E != a.actor.E || q->n_out != 1 || q->hd > 1) return RPO_ERR_ARG;
    CartConsts c;
    if (int e = load_consts(c, consts_host, partial)) return e;
    a.scale = scale; a.base = base; a.rows = rows; a.cap_steps = cap_steps; a.n_envs = n_envs; a.batch = batch;
    a.batch_out = batch_out; a.idx_out = idx_out; a.idx_in = idx_in; a.eps_in = eps_in; a.seed = (uint64_t)noise_seed;
    a.sample_seed = (uint64_t)sample_seed; a.salt = (uint32_t)sample_salt; a.noise_salt = (uint32_t)noise_salt;
    a.noise_id_base = (uint32_t)noise_id_base; a.ctrl = ctrl; a.max_steps = max_steps; a.corr_lr = corr_lr;
    a.corr_eps = corr_eps; a.corr_momentum = corr_momentum; a.box_lo = box_lo; a.box_hi = box_hi;
    a.q1_out = q1_out; a.q2_out = q2_out; a.qn1_out = qn1_out; a.qn2_out = qn2_out; a.logp_out = logp_out;
    a.x0_save1 = x0_save1; a.h1_save1 = h1_save1; a.x0_save2 = x0_save2; a.h1_save2 = h1_save2;
    const int grid = (batch + kRows - 1) / kRows;                   // x: row tile, y: role
    if (a.actor.E == 128) {
        hipLaunchKernelGGL((cart_sac_critic_forward_kernel<128, 256>), dim3(grid, 4), dim3(kFwdThreads), 0, (hipStream_t)stream, a, c);
    } else if (a.actor.E == 256) {
        hipLaunchKernelGGL((cart_sac_critic_forward_kernel<256, 256>), dim3(grid, 4), dim3(kFwdThreads), 0, (hipStream_t)stream, a, c);
    } else {
        return RPO_ERR_ARG;
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

static int fill_sac_args(SacCriticFwdArgs& a, int obs_dim, const rpo_mlp* actor_host, const rpo_mlp* ct1, const rpo_mlp* ct2,
                         const rpo_mlp* c1, const rpo_mlp* c2, bool need_actor, bool need_critics) {
    if (need_actor) {
        if (!actor_host) return RPO_ERR_NULL;
        a.actor = to_dev(actor_host);
        if (int e = check_actor(a.actor, obs_dim, 1)) return e;
    }
    if (need_critics) {
        if (!ct1 || !ct2 || !c1 || !c2) return RPO_ERR_NULL;
        a.critic_target1 = to_dev(ct1); a.critic_target2 = to_dev(ct2); a.critic1 = to_dev(c1); a.critic2 = to_dev(c2);
        const Mlp* qs[4] = {&a.critic_target1, &a.critic_target2, &a.critic1, &a.critic2};
        for (const Mlp* q : qs)
            if (q->S != obs_dim || q->A != 2 || q->cat || q->H != 256 || (q->E != 128 && q->E != 256) || q->E != qs[0]->E ||
                q->n_out != 1 || q->hd > 1)
                return RPO_ERR_ARG;
    }
    return 0;
}

int rpo_pendulum_sac_critic_front(const rpo_mlp* actor_host, float scale, float base, float box_lo, float box_hi,
                                  const float* rows, long long cap_steps, int n_envs, int batch, float* batch_out,
                                  long long* idx_out, const long long* idx_in, const float* eps_in,
                                  unsigned long long sample_seed, unsigned sample_salt, unsigned long long noise_seed,
                                  unsigned noise_id_base, unsigned noise_salt, const long long* ctrl, float* ap_out,
                                  float* logp_out, void* stream) {
    if (batch <= 0 || cap_steps <= 0 || n_envs <= 0) return RPO_ERR_ARG;
    if (!rows || !batch_out || !ctrl || !ap_out || !logp_out) return RPO_ERR_NULL;
    SacCriticFwdArgs a{};
    if (int e = fill_sac_args(a, 5, actor_host, nullptr, nullptr, nullptr, nullptr, true, false)) return e;
    a.scale = scale; a.base = base; a.box_lo = box_lo; a.box_hi = box_hi; a.rows = rows; a.cap_steps = cap_steps;
    a.n_envs = n_envs; a.batch = batch; a.batch_out = batch_out; a.idx_out = idx_out; a.idx_in = idx_in; a.eps_in = eps_in;
    a.sample_seed = (uint64_t)sample_seed; a.salt = (uint32_t)sample_salt; a.seed = (uint64_t)noise_seed;
    a.noise_id_base = (uint32_t)noise_id_base; a.noise_salt = (uint32_t)noise_salt; a.ctrl = ctrl; a.ap_out = ap_out;
    a.logp_out = logp_out;
    const int grid = (batch + kRows - 1) / kRows;
    if (a.actor.E == 128) {
        hipLaunchKernelGGL((pend_sac_critic_front_kernel<128, 256>), dim3(grid), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else if (a.actor.E == 256) {
        hipLaunchKernelGGL((pend_sac_critic_front_kernel<256, 256>), dim3(grid), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else {
        return RPO_ERR_ARG;
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_sac_critic_back(const rpo_mlp* critic_target1_host, const rpo_mlp* critic_target2_host,
                                 const rpo_mlp* critic1_host, const rpo_mlp* critic2_host, int batch, float* batch_rows,
                                 const float* next_actions, float* q1_out, float* q2_out, float* qn1_out, float* qn2_out,
                                 float* x0_save1, float* h1_save1, float* x0_save2, float* h1_save2, void* stream) {
    if (batch <= 0) return RPO_ERR_ARG;
    if (!batch_rows || !next_actions || !q1_out || !q2_out || !qn1_out || !qn2_out || !x0_save1 || !h1_save1 || !x0_save2 ||
        !h1_save2)
        return RPO_ERR_NULL;
    SacCriticFwdArgs a{};
    if (int e = fill_sac_args(a, 5, nullptr, critic_target1_host, critic_target2_host, critic1_host, critic2_host, false, true))
        return e;
    a.batch = batch; a.batch_out = batch_rows; a.next_actions = next_actions;
    a.q1_out = q1_out; a.q2_out = q2_out; a.qn1_out = qn1_out; a.qn2_out = qn2_out;
    a.x0_save1 = x0_save1; a.h1_save1 = h1_save1; a.x0_save2 = x0_save2; a.h1_save2 = h1_save2;
    const int grid = (batch + kRows - 1) / kRows;                   // x: row tile, y: role
    if (a.critic1.E == 128) {
        hipLaunchKernelGGL((pend_critic_back_kernel<128, 256, 1>), dim3(grid, 4), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((pend_critic_back_kernel<256, 256, 1>), dim3(grid, 4), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_ddpg_actor_forward(int env, const rpo_mlp* actor_host, const rpo_mlp* critic_host, float scale, float base,
                                    float box_lo, float box_hi, float eps_start, float eps_end, float eps_decay,
                                    const float* batch, int batch_size, const float* noise_in, unsigned long long seed,
                                    unsigned noise_id_base, unsigned noise_salt, const long long* ctrl, const float* nu,
                                    const float* consts_host, int partial, float* ap_det, float* noise_out,
                                    float* actions, float* q_out, float* dq_out, float* g_act, float* partial_out,
                                    float* actor_x0, float* actor_h1, float* critic_x0, float* critic_h1, void* stream) {
    if (!actor_host || !critic_host) return RPO_ERR_NULL;
    if (batch_size <= 0) return RPO_ERR_ARG;
    if (!batch || !ctrl || !nu || !ap_det || !noise_out || !actions || !q_out || !dq_out || !g_act || !partial_out ||
        !actor_x0 || !actor_h1 || !critic_x0 || !critic_h1)
        return RPO_ERR_NULL;
    if (env != 0 && env != 1) return RPO_ERR_ARG;
    const int obs_dim = env == 0 ? 6 : 5;
    ActorFwdArgs a{};
    a.actor = to_dev(actor_host); a.critic = to_dev(critic_host);
    if (int e = check_actor(a.actor, obs_dim, 0)) return e;
    if (a.actor.E != 128 || a.critic.S != obs_dim || a.critic.A != 2 || a.critic.cat || a.critic.H != 256 || a.critic.E != 128 ||
        a.critic.n_out != 1 || a.critic.hd > 1)
        return RPO_ERR_ARG;
    a.scale = scale; a.base = base; a.box_lo = box_lo; a.box_hi = box_hi; a.eps_start = eps_start; a.eps_end = eps_end;
    a.eps_decay = eps_decay; a.batch = batch; a.B = batch_size; a.noise_in = noise_in; a.seed = (uint64_t)seed;
    a.noise_id_base = (uint32_t)noise_id_base; a.noise_salt = (uint32_t)noise_salt; a.ctrl = ctrl; a.nu = nu;
    a.ap_det = ap_det; a.noise_out = noise_out; a.actions = actions; a.q_out = q_out; a.dq_out = dq_out; a.g_act = g_act;
    a.partial = partial_out; a.ax0 = actor_x0; a.ah1 = actor_h1; a.cx0 = critic_x0; a.ch1 = critic_h1;
    const int grid = (batch_size + kRows - 1) / kRows;
    if (env == 0) {
        CartConsts c;
        if (int e = load_consts(c, consts_host, partial)) return e;
        hipLaunchKernelGGL((ddpg_actor_forward_kernel<CartActEnv, 128, 256>), dim3(grid), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, a, c);
    } else {
        hipLaunchKernelGGL((ddpg_actor_forward_kernel<PendActEnv, 128, 256>), dim3(grid), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, a, PendActEnv::Consts{0});
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_ddpg_actor_backward(int env, const rpo_mlp* actor_host, const rpo_mlp_grad* actor_grad_host,
                                     const rpo_mlp* critic_host, int shared_embedding, const float* batch,
                                     int batch_size, const float* actions, const float* g_act, const float* ap_det,
                                     const float* noise, const float* dq, float eps_start, float eps_end,
                                     float eps_decay, float box_lo, float box_hi, float scale, float base,
                                     const long long* ctrl, const float* consts_host, int partial,
                                     const float* actor_x0, const float* actor_h1, const float* critic_x0,
                                     const float* critic_h1, float* actor_dh, float* actor_dx0, float* critic_dh,
                                     float* critic_dx0, float* da, float* dout, const float* partial_in, float* lag_out,
                                     float* nu_grad, float* gradmax, void* stream) {
    if (!actor_host || !actor_grad_host || !critic_host) return RPO_ERR_NULL;
    if (batch_size <= 0) return RPO_ERR_ARG;
    if (!batch || !actions || !g_act || !ap_det || !noise || !dq || !ctrl || !da || !dout || !partial_in || !lag_out || !nu_grad)
        return RPO_ERR_NULL;
    if (env != 0 && env != 1) return RPO_ERR_ARG;
    const int obs_dim = env == 0 ? 6 : 5, row = env == 0 ? RPO_CART_ROW : RPO_PEND_ROW;
    ActorBwdArgs p{};
    Mlp actor = to_dev(actor_host), critic = to_dev(critic_host);
    if (int e = check_actor(actor, obs_dim, 0)) return e;
    if (actor.E != 128 || critic.S != obs_dim || critic.A != 2 || critic.cat || critic.H != 256 || critic.E != 128 ||
        critic.n_out != 1)
        return RPO_ERR_ARG;
    if (!actor_x0 || !actor_h1 || !critic_x0 || !critic_h1 || !actor_dh || !actor_dx0 || !critic_dh || !critic_dx0)
        return RPO_ERR_NULL;
    MlpGrad none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    MlpGrad ag{actor_grad_host->Ws, actor_grad_host->bs, actor_grad_host->Wa, actor_grad_host->ba, actor_grad_host->W0,
               actor_grad_host->b0, actor_grad_host->W1, actor_grad_host->b1, actor_grad_host->W1b, actor_grad_host->b1b};
    if (!ag.Ws || !ag.bs || !ag.W0 || !ag.b0 || !ag.W1 || !ag.b1) return RPO_ERR_NULL;
    // the critic only propagates (d/d action, and dx0 for a shared embedding): no parameter gradients of its own
    p.critic = BwdArgs{critic, none, batch_size, batch, row, actions, 2, critic_x0, critic_h1, dq, critic_dh,
                       critic_dx0, da, 0, 0, nullptr};
    p.actor = BwdArgs{actor, ag, batch_size, batch, row, nullptr, 0, actor_x0, actor_h1, dout, actor_dh, actor_dx0,
                      nullptr, 1, 0, gradmax};
    p.g_act = g_act; p.ap_det = ap_det; p.noise = noise; p.eps_start = eps_start; p.eps_end = eps_end;
    p.eps_decay = eps_decay; p.box_lo = box_lo; p.box_hi = box_hi; p.scale = scale; p.base = base; p.ctrl = ctrl;
    p.dout = dout; p.partial = partial_in; p.n_parts = (batch_size + kRows - 1) / kRows; p.lag_out = lag_out;
    p.nu_grad = nu_grad; p.shared_embedding = shared_embedding;
    const int grid = (batch_size + kRows - 1) / kRows;
    if (env == 0) {
        CartConsts c;
        if (int e = load_consts(c, consts_host, partial)) return e;
        hipLaunchKernelGGL((ddpg_actor_backward_kernel<CartActEnv, 128, 256>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,
                           p, c);
    } else {
        hipLaunchKernelGGL((ddpg_actor_backward_kernel<PendActEnv, 128, 256>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,
                           p, PendActEnv::Consts{0});
    }
    RPO_LAUNCH_CHECK();
    const int fl_outputs = actor.E * (actor.S + 1);
    const int grid_w = (actor.H / 16) * (128 / 64) + actor.H / 64 + mlp_fl_blocks(fl_outputs);
    {
        const SplitK sk = splitk_plan(p.actor, actor_grad_host->splitk_scratch, actor_grad_host->splitk_floats);
        if (sk.Z > 0) return launch_weights_splitk<128, 256>(p.actor, sk, grid_w, (hipStream_t)stream);     // large batches
    }
    hipLaunchKernelGGL((actor_weights_kernel<128, 256>), dim3(grid_w), dim3(kThreads), 0, (hipStream_t)stream, p.actor);
    RPO_LAUNCH_CHECK();
    return 0;
}

static int sac_actor_nets(int obs_dim, const rpo_mlp* actor_host, const rpo_mlp* c1, const rpo_mlp* c2, Mlp& actor, Mlp& q1,
                          Mlp& q2) {
    if (!actor_host || !c1 || !c2) return RPO_ERR_NULL;
    actor = to_dev(actor_host); q1 = to_dev(c1); q2 = to_dev(c2);
    if (int e = check_actor(actor, obs_dim, 1)) return e;
    if (actor.E != 128) return RPO_ERR_ARG;
    const Mlp* qs[2] = {&q1, &q2};
    for (const Mlp* q : qs)
        if (q->S != obs_dim || q->A != 2 || q->cat || q->H != 256 || q->E != 128 || q->n_out != 1 || q->hd > 1) return RPO_ERR_ARG;
    return 0;
}

int rpo_sac_actor_forward(int env, const rpo_mlp* actor_host, const rpo_mlp* critic1_host, const rpo_mlp* critic2_host,
                          float scale, float base, float box_lo, float box_hi, float alpha, const float* batch,
                          int batch_size, const float* noise_in, unsigned long long seed, unsigned noise_id_base,
                          unsigned noise_salt, const long long* ctrl, const float* nu, const float* consts_host,
                          int partial, float* raw, float* noise_out, float* logp, float* actions, float* dq1, float* dq2,
                          float* g_act, float* partial_out, float* actor_x0, float* actor_h1, float* critic1_x0,
                          float* critic1_h1, float* critic2_x0, float* critic2_h1, void* stream) {
    if (env != 0 && env != 1) return RPO_ERR_ARG;
    if (batch_size <= 0) return RPO_ERR_ARG;
    if (!batch || !ctrl || !nu || !raw || !noise_out || !logp || !actions || !dq1 || !dq2 || !g_act || !partial_out ||
        !actor_x0 || !actor_h1 || !critic1_x0 || !critic1_h1 || !critic2_x0 || !critic2_h1)
        return RPO_ERR_NULL;
    SacActorFwdArgs a{};
    if (int e = sac_actor_nets(env == 0 ? 6 : 5, actor_host, critic1_host, critic2_host, a.actor, a.critic1, a.critic2)) return e;
    a.scale = scale; a.base = base; a.box_lo = box_lo; a.box_hi = box_hi; a.alpha = alpha; a.batch = batch; a.B = batch_size;
    a.noise_in = noise_in; a.seed = (uint64_t)seed; a.noise_id_base = (uint32_t)noise_id_base;
    a.noise_salt = (uint32_t)noise_salt; a.ctrl = ctrl; a.nu = nu; a.raw = raw; a.noise_out = noise_out; a.logp = logp;
    a.actions = actions; a.dq1 = dq1; a.dq2 = dq2; a.g_act = g_act; a.partial = partial_out; a.ax0 = actor_x0;
    a.ah1 = actor_h1; a.c1x0 = critic1_x0; a.c1h1 = critic1_h1; a.c2x0 = critic2_x0; a.c2h1 = critic2_h1;
    const int grid = (batch_size + kRows - 1) / kRows;
    if (env == 0) {
        CartConsts c;
        if (int e = load_consts(c, consts_host, partial)) return e;
        hipLaunchKernelGGL((sac_actor_forward_kernel<CartActEnv, 128, 256>), dim3(grid, 2), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, a, c);
    } else {
        hipLaunchKernelGGL((sac_actor_forward_kernel<PendActEnv, 128, 256>), dim3(grid, 2), dim3(kFwdThreads), 0,
                           (hipStream_t)stream, a, PendActEnv::Consts{0});
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_sac_actor_backward(int env, const rpo_mlp* actor_host, const rpo_mlp_grad* actor_grad_host,
                           const rpo_mlp* critic1_host, const rpo_mlp* critic2_host, int shared_embedding,
                           const float* batch, int batch_size, const float* actions, const float* g_act, const float* raw,
                           const float* noise, const float* logp, float* dq1, float* dq2, float dlogp, float box_lo,
                           float box_hi, float scale, float base, const float* consts_host, int partial,
                           const float* actor_x0, const float* actor_h1, const float* critic1_x0, const float* critic1_h1,
                           const float* critic2_x0, const float* critic2_h1, float* actor_dh, float* actor_dx0,
                           float* critic1_dh, float* critic1_dx0, float* critic2_dh, float* critic2_dx0, float* da1,
                           float* da2, float* dout, float* partial_in, float* lag_out, float* nu_grad,
                           float* gradmax, void* stream) {
    if (env != 0 && env != 1) return RPO_ERR_ARG;
    if (batch_size <= 0) return RPO_ERR_ARG;
    if (!actor_grad_host || !batch || !actions || !g_act || !raw || !noise || !logp || !dq1 || !dq2 || !da1 || !da2 || !dout ||
        !partial_in || !lag_out || !nu_grad || !actor_x0 || !actor_h1 || !critic1_x0 || !critic1_h1 || !critic2_x0 ||
        !critic2_h1 || !actor_dh || !actor_dx0 || !critic1_dh || !critic1_dx0 || !critic2_dh || !critic2_dx0)
        return RPO_ERR_NULL;
    SacActorBwdArgs p{};
    Mlp actor, q1, q2;
    const int obs_dim = env == 0 ? 6 : 5, row = env == 0 ? RPO_CART_ROW : RPO_PEND_ROW;
    if (int e = sac_actor_nets(obs_dim, actor_host, critic1_host, critic2_host, actor, q1, q2)) return e;
    MlpGrad none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    MlpGrad ag{actor_grad_host->Ws, actor_grad_host->bs, actor_grad_host->Wa, actor_grad_host->ba, actor_grad_host->W0,
               actor_grad_host->b0, actor_grad_host->W1, actor_grad_host->b1, actor_grad_host->W1b, actor_grad_host->b1b};
    if (!ag.Ws || !ag.bs || !ag.W0 || !ag.b0 || !ag.W1 || !ag.b1 || !ag.W1b || !ag.b1b) return RPO_ERR_NULL;
    p.critic1 = BwdArgs{q1, none, batch_size, batch, row, actions, 2, critic1_x0, critic1_h1, dq1, critic1_dh, critic1_dx0, da1,
                        0, 0, nullptr};
    p.critic2 = BwdArgs{q2, none, batch_size, batch, row, actions, 2, critic2_x0, critic2_h1, dq2, critic2_dh, critic2_dx0, da2,
                        0, 0, nullptr};
    p.actor = BwdArgs{actor, ag, batch_size, batch, row, nullptr, 0, actor_x0, actor_h1, dout, actor_dh, actor_dx0, nullptr, 1, 0,
                      gradmax};
    p.logp = logp; p.q_dq1 = dq1; p.q_dq2 = dq2;
    p.g_act = g_act; p.raw = raw; p.noise = noise; p.dlogp = dlogp; p.box_lo = box_lo; p.box_hi = box_hi; p.scale = scale;
    p.base = base; p.dout = dout; p.partial = partial_in; p.n_parts = (batch_size + kRows - 1) / kRows; p.lag_out = lag_out;
    p.nu_grad = nu_grad; p.n_ineq = env == 0 ? 6 : 1; p.shared_embedding = shared_embedding;
    const int grid = (batch_size + kRows - 1) / kRows;
    if (env == 0) {
        CartConsts c;
        if (int e = load_consts(c, consts_host, partial)) return e;
        hipLaunchKernelGGL((sac_actor_backward_kernel<CartActEnv, 128, 256>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,
                           p, c);
    } else {
        hipLaunchKernelGGL((sac_actor_backward_kernel<PendActEnv, 128, 256>), dim3(grid), dim3(kThreads), 0, (hipStream_t)stream,
                           p, PendActEnv::Consts{0});
    }
    RPO_LAUNCH_CHECK();
    const int fl_outputs = actor.E * (actor.S + 1);
    const int grid_w = (actor.H / 16) * (128 / 64) + actor.H / 64 + mlp_fl_blocks(fl_outputs);
    {
        const SplitK sk = splitk_plan(p.actor, actor_grad_host->splitk_scratch, actor_grad_host->splitk_floats);
        if (sk.Z > 0) return launch_weights_splitk<128, 256>(p.actor, sk, grid_w, (hipStream_t)stream);     // large batches
    }
    hipLaunchKernelGGL((actor_weights_kernel<128, 256>), dim3(grid_w), dim3(kThreads), 0, (hipStream_t)stream, p.actor);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_ddpg_critic_front(const rpo_mlp* actor_target_host, float scale, float base, const float* rows,
                                   long long cap_steps, int n_envs, int batch, float* batch_out, long long* idx_out,
                                   const long long* idx_in, unsigned long long sample_seed, unsigned sample_salt,
                                   const long long* ctrl, float* ap_out, void* stream) {
    if (!actor_target_host) return RPO_ERR_NULL;
    if (batch <= 0 || cap_steps <= 0 || n_envs <= 0) return RPO_ERR_ARG;
    if (!rows || !batch_out || !ctrl || !ap_out) return RPO_ERR_NULL;
    SacCriticFwdArgs a{};
    a.actor = to_dev(actor_target_host);
    if (int e = check_actor(a.actor, 5, 0)) return e;
    a.scale = scale; a.base = base; a.rows = rows; a.cap_steps = cap_steps; a.n_envs = n_envs; a.batch = batch;
    a.batch_out = batch_out; a.idx_out = idx_out; a.idx_in = idx_in; a.sample_seed = (uint64_t)sample_seed;
    a.salt = (uint32_t)sample_salt; a.ctrl = ctrl; a.ap_out = ap_out;
    const int grid = (batch + kRows - 1) / kRows;
    if (a.actor.E == 128) {
        hipLaunchKernelGGL((pend_ddpg_critic_front_kernel<128, 256>), dim3(grid), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else if (a.actor.E == 256) {
        hipLaunchKernelGGL((pend_ddpg_critic_front_kernel<256, 256>), dim3(grid), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else {
        return RPO_ERR_ARG;
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_ddpg_critic_back(const rpo_mlp* critic_target_host, const rpo_mlp* critic_host, int batch,
                                  float* batch_rows, const float* next_actions, float* q_out, float* qn_out,
                                  float* x0_save, float* h1_save, void* stream) {
    if (!critic_target_host || !critic_host) return RPO_ERR_NULL;
    if (batch <= 0) return RPO_ERR_ARG;
    if (!batch_rows || !next_actions || !q_out || !qn_out || !x0_save || !h1_save) return RPO_ERR_NULL;
    SacCriticFwdArgs a{};
    a.critic_target1 = to_dev(critic_target_host); a.critic1 = to_dev(critic_host);
    const Mlp* qs[2] = {&a.critic_target1, &a.critic1};
    for (const Mlp* q : qs)
        if (q->S != 5 || q->A != 2 || q->cat || q->H != 256 || (q->E != 128 && q->E != 256) || q->E != qs[0]->E || q->n_out != 1 ||
            q->hd > 1)
            return RPO_ERR_ARG;
    a.batch = batch; a.batch_out = batch_rows; a.next_actions = next_actions; a.q1_out = q_out; a.qn1_out = qn_out;
    a.x0_save1 = x0_save; a.h1_save1 = h1_save;
    const int grid = (batch + kRows - 1) / kRows;                   // x: row tile, y: role (Q_targ | Q)
    if (a.critic1.E == 128) {
        hipLaunchKernelGGL((pend_critic_back_kernel<128, 256, 0>), dim3(grid, 2), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL((pend_critic_back_kernel<256, 256, 0>), dim3(grid, 2), dim3(kFwdThreads), 0, (hipStream_t)stream, a);
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
