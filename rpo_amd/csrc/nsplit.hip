// Column-split update kernels (MI355X): the batch-256 update of RPODDPG / RPOSAC as short launches that use the width
// of the chip -- 128 workgroups per network evaluation instead of 16 -- cut at the seams where a value needs every
// hidden column (the heads).  nsplit_dev.h has the slab function and the argument for why the results are bitwise
// those of the row-tile kernels.
#include <stdlib.h>

#include "cartsafe_dev.h"
#include "heads_dev.h"
#include "mlp_bwd.h"
#include "nsplit_dev.h"
#include "pendulum_dev.h"
#include "rollout_env.h"

namespace {

using namespace rpo_mlp_dev;

Mlp to_dev(const rpo_mlp* h) {
    return Mlp{h->Ws, h->bs, h->Wa, h->ba, h->W0, h->b0, h->W1, h->b1, h->W1b, h->b1b, h->S, h->A, h->E, h->H, h->n_out, h->cat, h->head_dim};
}

bool split_ok(const Mlp& m) {
    return m.E == 128 && m.H == 256 && !m.cat && m.hd <= 1 && m.S <= 6 && m.A <= 4 && m.n_out >= 1 && m.n_out <= 2 && m.Ws && m.W0 &&
           m.W1 && (m.A == 0 || m.Wa) && (m.n_out == 1 || m.W1b);
}

// (row tile, column group) of a slab workgroup.  The launch grid is (8 groups, T tiles, planes); inside a plane the
// hardware numbers workgroups x-fastest and workgroup b lands on XCD b % 8.  Here the ROW TILE is the fastest index of that
// numbering: all 8 column groups of a tile -- and every later launch's workgroups for that tile -- run on the XCD
// tile % 8, so what one launch leaves for the next (partials, saved activations) is produced and consumed by the same L2.
struct NsBlock { int tile, g; };
__device__ __forceinline__ NsBlock ns_block() {
    const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, T = (int)gridDim.y;
    return NsBlock{lin % T, lin / T};
}

// ------------------------------------------------------------------------------------------- stand-alone split forward
struct SplitFwdArgs {
    Mlp net;
    int n;
    const float* s; int s_stride;
    const float* a; int a_stride;
    float* part;           // [8, n, 2] head partials of the column groups
    float* x0_save; float* h1_save;
};

// grid (row tiles, 8 column groups, networks)
struct SplitFwdArgs4 { SplitFwdArgs net[4]; };

__global__ __launch_bounds__(kNsThreads) void mlp_forward_split_kernel(SplitFwdArgs4 all) {
    __shared__ NsLds<128> lds;
    const SplitFwdArgs& p = all.net[blockIdx.z];
    const NsBlock nb = ns_block();
    const int row0 = nb.tile * kRows, g = nb.g, tid = threadIdx.x;
    NsWeights<128> w;
    ns_load_weights<128, 256>(p.net, g, w);
    if (tid < kRows * 8) {
        const int r = tid >> 3, i = tid & 7;
        const bool live = row0 + r < p.n;
        lds.in_s[tid] = (live && i < p.net.S) ? p.s[(size_t)(row0 + r) * p.s_stride + i] : 0.0f;
        lds.in_a[tid] = (live && i < p.net.A) ? p.a[(size_t)(row0 + r) * p.a_stride + i] : 0.0f;
    }
    ns_hidden<128, 256>(p.net, w, lds, g, row0, p.n, p.part, p.x0_save, p.h1_save);
}

__global__ __launch_bounds__(RPO_BLOCK) void mlp_split_head_kernel(Mlp net, int n, const float* __restrict__ part,
                                                                   float* __restrict__ out, int out_mode, float scale,
                                                                   float base) {
    const int idx = blockIdx.x * RPO_BLOCK + threadIdx.x;
    if (idx >= n * net.n_out) return;
    const int r = idx / net.n_out, o = idx - r * net.n_out;
    float v = ns_head(part, n, r, o, o == 0 ? net.b1[0] : net.b1b[0]);
    if (out_mode == 1 && o == 0) v = scale * tanhf(v) + base;
    out[idx] = v;
}


// ====================================================================================== column-split critic update
using rpo_cart_dev::CartConsts;

struct CartRow { typedef CartEnv Env; static constexpr int ROW = RPO_CART_ROW, CH = 6, RCH = RPO_CART_RING / 4, S = 6, A_OFF = 6, NS_OFF = 8, R_OFF = 14; };
struct PendRow { typedef PendEnv Env; static constexpr int ROW = RPO_PEND_ROW, CH = 4, RCH = RPO_PEND_RING / 4, S = 5, A_OFF = 5, NS_OFF = 7, R_OFF = 12; };

// Device view of rpo_split_update (host network descriptors resolved to device pointer sets).
struct SplitArgs {
    Mlp actor, actor_target, critic[2], critic_target[2];
    MlpGrad critic_grad[2], actor_grad;
    int twin, B;
    const float* rows; long long cap_steps; int n_envs;
    float* batch_out; long long* idx_out; const long long* idx_in;
    uint64_t sample_seed; uint32_t sample_salt;
    const float* eps_in; uint64_t noise_seed; uint32_t noise_id_base, noise_salt;
    const long long* ctrl;
    float scale, base, box_lo, box_hi;
    int max_steps; float corr_lr, corr_eps, corr_momentum;
    float alpha, gamma, eps_start, eps_end, eps_decay;
    float *part_pi, *part_q[2], *part_qn[2];
    float *x0[2], *h1[2], *x0_a, *h1_a;
    float *logp, *next_actions; int* proj_iters;
    float *dq[2], *loss_partial, *dx0[2], *dx0_a, *gradmax;
    const float* nu; float* nu_grad;
    float *ap_det, *noise_out, *raw, *actions, *g_act, *lag_partial, *lag_out, *da_part, *dout;
    int shared_embedding;
    long long* rollout_ctrl; float* rollout_stats; int rollout_stats_cap;
    int* prep_step; float prep_beta1, prep_beta2; long long* clock_out; float* gradmax_reset;
    int* prep2_step[3]; float prep2_beta1[3], prep2_beta2[3]; float* gradmax_reset2;
    long long* updates_out;
    float* part_pol;              // policy step: head partials of pi(s) (pol_a -> pol_b); NULL: part_pi is reused
    unsigned* tile_sync;          // fused front launch: per-tile arrival words (ns_tile_arrive / ns_tile_wait)
    int debug;                    // tests: bit 0 = the policy workgroup (tile 0, group 0) of a fused front withholds its hand-over
};

// ---- Hand-over between workgroups of ONE launch (the fused front of the critic update, rpo_split_critic_front).
// What a row tile's stages pass on -- head partials of the 8 column groups, the gathered rows, saved activations -- is
// produced and consumed by the workgroups of that tile only, so the stages need no launch boundary between them, only a
// per-tile arrival word.  Consumers are workgroups with HIGHER block ids (dispatched after every producer, so a waiting
// workgroup can never keep a producer off the chip); they request their weights first and then poll the word.
//
// Scope.  With the row tile fastest in the block numbering (ns_block) and T % 8 == 0 every workgroup of a tile, in every
// plane, runs on XCD tile % 8: producer and consumer share ONE L2.  The hand-over therefore stays inside that L2 -- stores
// are complete there once vmcnt is 0 (the vector L1 writes through), the arrival is an atomic executed in that L2, the
// consumer polls it with a returning atomic (never served by its L1) and drops its L1 before reading the data.  Device-scope
// fences instead (buffer_wbl2 / buffer_inv sc1: write back and invalidate the WHOLE L2, by 256 waves of the launch) made the
// fused launch 8 us slower than the two launches it replaces.  The placement is a property of the dispatcher, so the host
// checks it once per process with a probe launch of the same grid (rpo_xcc_probe; rpo_amd/ops.py front_launch_ok) and keeps
// the separate launches if any tile's workgroups are spread over XCDs.
//
// tile_sync layout: word [(stage * T + tile) * kNsSyncStride], stage 0 / 1 = arrival counts, stage 2 = consumers that have passed their waits
// (the last one zeroes the tile's words: the buffer is all zeros again when the launch ends); word [3 * T * kNsSyncStride] = 1 if a wait
// ever gave up (kNsSpinMax polls: a lost producer must not hang the device; the tests check the word).
constexpr int kNsSpinMax = 1 << 18;
// one 128-byte line per arrival word: with the 16 tiles' words in ONE line the polls and arrivals of the whole launch queued on
// one L2 channel per XCD and the waits took 7.8 us (measured) instead of the ~3 us the producers need
constexpr int kNsSyncStride = 32;

// (a word holds two 16-bit counts: `inc` = 1 for the low one, 1 << 16 for the high one)
__device__ __forceinline__ void ns_tile_arrive(unsigned* word, unsigned inc = 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every wave: its stores have reached the XCD's L2 ...
    __syncthreads();                                             // ... before the workgroup's one arrival
    if (threadIdx.x == 0) __hip_atomic_fetch_add(word, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ unsigned ns_poll(unsigned* word) {     // returning atomic: executed in the L2, not a cached load
    unsigned v;
    asm volatile("global_atomic_or %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(word), "v"(0u) : "memory");
    return v;
}

__device__ __forceinline__ void ns_tile_wait(unsigned* word, unsigned need, unsigned* gave_up, int shift = 0) {
    if (threadIdx.x == 0) {
        int polls = 0;
        while (((ns_poll(word) >> shift) & 0xffffu) < need) {
            if (++polls >= kNsSpinMax) { *gave_up = 1u; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    asm volatile("buffer_inv sc0" ::: "memory");                 // the vector L1 of this CU holds nothing older than the arrival
}

// a consumer of the tile (`n_consumers` in all) has passed its waits: the last one clears the tile's words
__device__ __forceinline__ void ns_tile_passed(unsigned* sync, int T, int tile, unsigned n_consumers) {
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(sync + (2 * T + tile) * kNsSyncStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ==
            n_consumers - 1u) {
        sync[tile * kNsSyncStride] = 0u; sync[(T + tile) * kNsSyncStride] = 0u; sync[(2 * T + tile) * kNsSyncStride] = 0u;
    }
}

// XCD of every workgroup of a grid (see above): out[linear block id] = XCC_ID
__global__ void xcc_probe_kernel(int* out) {
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[(size_t)blockIdx.x + (size_t)gridDim.x * ((size_t)blockIdx.y + (size_t)gridDim.y * blockIdx.z)] = (int)(id & 15u);
    }
}

// Diagnostic: where the dispatcher puts the workgroups of a grid that is resident all at once.  Every workgroup records
// (XCC_ID << 16) | HW_ID[15:0] of its first wave (HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) and then
// idles for `spin` s_sleep rounds so that the grid's workgroups coexist like those of a 5-10 us launch.
__global__ void hw_probe_kernel(int* out, int spin) {
    extern __shared__ float hw_probe_lds[];
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[(size_t)blockIdx.x + (size_t)gridDim.x * ((size_t)blockIdx.y + (size_t)gridDim.y * blockIdx.z)] =
            (int)(((xcc & 15u) << 16) | (hw & 0xffffu));
        hw_probe_lds[0] = 0.0f;
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

// ReplayBuffer.sample (buffer.py:31-34): counter-based draw + gather of the tile's 16 rows (same draw in every workgroup
// of the tile; `publish` stores the gathered rows / indices once).
template <class L>
__device__ __forceinline__ void ns_sample(const SplitArgs& p, float4* tile, int row0, long long t, bool publish) {
    const int tid = threadIdx.x, B = p.B;
    if (tid < kRows * L::CH) {
        const int r = tid / L::CH, ch = tid - r * L::CH;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (row0 + r < B) {
            long long row;
            if (p.idx_in) {
                row = p.idx_in[row0 + r];
            } else {
                const unsigned long long n_valid = (unsigned long long)((t < p.cap_steps ? t : p.cap_steps) * (long long)p.n_envs);
                const rpo_u4 u = rpo_philox(p.sample_seed, (uint32_t)(row0 + r), (uint32_t)t + p.sample_salt, RPO_STREAM_SAMPLE,
                                            (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                row = (long long)__umul64hi(((unsigned long long)u.x << 32) | u.y, n_valid);
            }
            v = reinterpret_cast<const float4*>(p.rows)[row * L::RCH + ch];
            if (publish) {
                reinterpret_cast<float4*>(p.batch_out)[(size_t)(row0 + r) * L::CH + ch] = v;
                if (p.idx_out && ch == 0) p.idx_out[row0 + r] = row;
            }
        }
        tile[tid] = v;
    }
}

// The tile's 16 gathered rows from batch_out (written by an earlier launch)
template <class L>
__device__ __forceinline__ void ns_load_tile(const SplitArgs& p, float4* tile, int row0) {
    const int tid = threadIdx.x;
    if (tid < kRows * L::CH) {
        const int r = tid / L::CH;
        tile[tid] = row0 + r < p.B ? reinterpret_cast<const float4*>(p.batch_out)[(size_t)row0 * L::CH + tid]
                                   : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

template <class L>
__device__ __forceinline__ void ns_stage(NsLds<128>& lds, const float* tf, bool next, bool with_action) {
    const int tid = threadIdx.x;
    if (tid < kRows * 8) {
        const int r = tid >> 3, i = tid & 7;
        lds.in_s[tid] = i < L::S ? tf[r * L::ROW + (next ? L::NS_OFF : 0) + i] : 0.0f;
        lds.in_a[tid] = (with_action && i < 2) ? tf[r * L::ROW + L::A_OFF + i] : 0.0f;
    }
}

// ---- fwd_a: grid (row tiles, 8 column groups, roles).  Role 0 = the policy on s' (pi_targ for RPODDPG, pi for RPOSAC),
//      roles 1.. = the critics on the stored (s, a), pre-activations saved for the backward pass.
template <class L>
__device__ __forceinline__ void fwd_a_role(const SplitArgs& p, NsLds<128>& lds, float4* tile, int row0, int g, int role,
                                           float (*hp)[4] = nullptr, unsigned* rows_word = nullptr) {
    if (p.rollout_ctrl && role == 0 && g == 0 && row0 == 0 && threadIdx.x < RPO_WAVE) {
        // the rollout before this update left its clock to us (defer_clock): every workgroup of it has finished, so one
        // wave advances the step counter and clears the statistics row of the next step -- no arrival counting
        const long long tr = p.rollout_ctrl[RPO_CTRL_T];
        if (p.rollout_stats && p.rollout_stats_cap > 1) {
            float* nxt = p.rollout_stats + ((tr + 1) % p.rollout_stats_cap) * RPO_STATS_SUB * RPO_STATS_LEN;
            for (int k = threadIdx.x; k < RPO_STATS_SUB * RPO_STATS_LEN; k += RPO_WAVE) nxt[k] = 0.0f;
        }
        if (threadIdx.x == 0) p.rollout_ctrl[RPO_CTRL_T] = tr + 1;
    }
    if (role == 0 && g == 0 && row0 == 0 && threadIdx.x < RPO_GRADMAX_SLOTS) {   // on behalf of earlier (prepared) Adam launches
        if (p.gradmax_reset) p.gradmax_reset[threadIdx.x * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS)] = 0.0f;
        if (p.gradmax_reset2) p.gradmax_reset2[threadIdx.x * (RPO_GRADMAX_LEN / RPO_GRADMAX_SLOTS)] = 0.0f;
    }
    const Mlp& net = role == 0 ? (p.twin ? p.actor : p.actor_target) : p.critic[role - 1];
    NsWeights<128> w;
    ns_load_weights<128, 256>(net, g, w);
    const long long t = p.ctrl[RPO_CTRL_T];
    ns_sample<L>(p, tile, row0, t, role == 0 && g == 0);
    if (rows_word) ns_tile_arrive(rows_word);                    // "the tile's gathered rows are published" (it synchronises)
    else __syncthreads();
    ns_stage<L>(lds, reinterpret_cast<const float*>(tile), role == 0, role != 0);
    if (role == 0) ns_hidden<128, 256>(net, w, lds, g, row0, p.B, p.part_pi, nullptr, nullptr, nullptr, hp);
    else ns_hidden<128, 256>(net, w, lds, g, row0, p.B, p.part_q[role - 1], p.x0[role - 1], p.h1[role - 1]);
}

template <class L>
__global__ __launch_bounds__(kNsThreads) void split_critic_fwd_a_kernel(SplitArgs p) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    // row tile fastest (ns_block): every XCD's L2 pulls each network's whole W0 per launch (more fetched bytes than with the
    // column group fastest, where an XCD fetched only its 1/8 slice), but what the chain hands from launch to launch for a
    // tile stays in the L2 that produced it -- the chain is latency-bound, not byte-bound: -7 % per iteration (A/B, DESIGN A.3)
    const NsBlock nb = ns_block();
    fwd_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g, blockIdx.z);
}

// The policy head of row i from the slab partials: tanh box (RPODDPG, model/policy.py:30-31) or rsample of the squashed
// Gaussian + clip (RPOSAC, model/policy.py:53-66, agent/sac_pa.py:111; the draw of the update step)
__device__ __forceinline__ float ns_policy_head(const SplitArgs& p, int i, long long t, float* logp) {
    if (!p.twin) {
        const float v = ns_head(p.part_pi, p.B, i, 0, p.actor_target.b1[0]);
        return p.scale * tanhf(v) + p.base;
    }
    const float rm = ns_head(p.part_pi, p.B, i, 0, p.actor.b1[0]);
    const float rl = ns_head(p.part_pi, p.B, i, 1, p.actor.b1b[0]);
    float e;
    if (p.eps_in) {
        e = p.eps_in[i];
    } else {                                                   // == rpo_philox_normal(id_base, salt, RPO_STREAM_POLICY)
        const rpo_u4 u = rpo_philox(p.noise_seed, p.noise_id_base + (uint32_t)i, (uint32_t)t + p.noise_salt, RPO_STREAM_POLICY,
                                    (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
        e = rpo_normal(u.x, u.y);
    }
    return rpo_head_dev::gauss_head_row(rm, rl, e, p.scale, p.base, p.box_lo, p.box_hi, 0, logp);
}

// ---- fwd_b: grid (row tiles, 8 column groups, target critics).  PROJ = 1 (CartSafe): head -> Complete + Proj per row
//      (== the reference's batched call for this env) in the prologue; PROJ = 0 (SpringPendulum): the projected actions
//      come from rpo_split_pend_head_project.
template <class L, int PROJ>
__device__ __forceinline__ void fwd_b_role(const SplitArgs& p, const CartConsts& c, NsLds<128>& lds, float4* tile, int row0, int g,
                                           int k, unsigned* wait_word = nullptr, unsigned wait_need = 0u) {
    const int tid = threadIdx.x;
    const Mlp& net = p.critic_target[k];
    NsWeights<128> w;
    ns_load_weights<128, 256>(net, g, w);
    const long long t = p.ctrl[RPO_CTRL_T];
    // fused front: the rows are gathered HERE too (same draw, same rows as fwd_a's workgroups of this launch, which publish
    // them) -- the gather and the staging run while the tile's policy slabs are still being computed; only the head needs them
    if (wait_word) ns_sample<L>(p, tile, row0, t, false);
    else ns_load_tile<L>(p, tile, row0);
    __syncthreads();
    ns_stage<L>(lds, reinterpret_cast<const float*>(tile), true, false);
    __syncthreads();                                             // (ns_stage zeroed in_a: order it before the writes below)
    float pre[kRows];
    if (wait_word) {                                             // the tile's policy slabs of THIS launch
        ns_layer1_state<128>(net, w, lds, pre);                  // (the state half of layer 1 does not need them either)
        const int T = (p.B + kRows - 1) / kRows;
        ns_tile_wait(wait_word, wait_need, p.tile_sync + 3 * T * kNsSyncStride);
    }
    if (tid < kRows) {
        const int i = row0 + tid;
        float2 act = make_float2(0.0f, 0.0f);
        if (i < p.B) {
            if (PROJ) {
                float logp = 0.0f;
                const float ap = ns_policy_head(p, i, t, &logp);
                rpo_cart_dev::ActArgs a{};
                a.noise_mode = RPO_NOISE_NONE;
                a.max_steps = p.max_steps; a.corr_lr = p.corr_lr; a.corr_eps = p.corr_eps; a.corr_momentum = p.corr_momentum;
                a.box_lo = p.box_lo; a.box_hi = p.box_hi;
                int it;
                act = rpo_cart_dev::cart_explore_project(a, c, i, ap, 0.0f, t, it);
                if (p.twin && g == 0 && k == 0) p.logp[i] = logp;
            } else {
                act = reinterpret_cast<const float2*>(p.next_actions)[i];
            }
        }
        lds.in_a[tid * 8] = act.x;
        lds.in_a[tid * 8 + 1] = act.y;
    }
    ns_hidden<128, 256>(net, w, lds, g, row0, p.B, p.part_qn[k], nullptr, nullptr, wait_word ? pre : nullptr);
}

template <class L, int PROJ>
__global__ __launch_bounds__(kNsThreads) void split_critic_fwd_b_kernel(SplitArgs p, CartConsts c) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    const NsBlock nb = ns_block();
    fwd_b_role<L, PROJ>(p, c, lds, tile, nb.tile * kRows, nb.g, blockIdx.z);
}

// ============================================================================== rollout stages riding on update launches
// Vector step t+1 needs nothing the critic update of step t produces (no shared state embedding, no policy step in
// between) and the update reads its own clock, so the step is given to workgroups the update launches leave idle:
//   * the actor forward of the lanes (column-split: 16 lanes x 32 hidden columns per workgroup, head partials to `part`)
//     touches neither the ring nor anything the update writes: it rides on fwd_a AND fwd_b, a lane range each -- 2048
//     extra workgroups behind one launch cost more than the launch (13.3 vs 8.8 us), ~1000 ride almost free;
//   * explore + project + env step + replay scatter + statistics (one thread per lane) write the ring, so they run behind
//     fwd_a's gather: they ride on bwd_b, the longest of the remaining launches.
// Same arithmetic as rollout_kernel (fused.hip): the slab forward is bitwise the row-tile forward (nsplit_dev.h), the lane
// functions are the same inlined code.
template <class ENV>
struct RideArgs {
    Mlp actor;
    float scale, base;
    int gauss, n;
    int lane0, lane1;             // forward stages: the lanes of this launch (multiples of 16, or n)
    int defer_clock;              // step stage: the next fwd_a advances ctrl[T] (rpo_split_update.rollout_ctrl)
    typename ENV::ActArgs act;
    typename ENV::StepArgs step;
    float* part;                  // [8, n, 2]
};

__device__ __forceinline__ const CartConsts& env_consts(const CartConsts& c, CartEnv*) { return c; }
__device__ __forceinline__ PendEnv::Consts env_consts(const CartConsts&, PendEnv*) { return PendEnv::Consts{0}; }

// forward workgroup `wg` of the launch: column group g of lanes [lane0 + 16 wg, +16)
template <class ENV>
__device__ __forceinline__ void ride_forward(const RideArgs<ENV>& r, NsLds<128>& lds, int wg, int g) {
    const int row0 = r.lane0 + wg * kRows;
    if (row0 >= r.lane1) return;
    NsWeights<128> w;
    ns_load_weights<128, 256>(r.actor, g, w);
    ENV::stage_obs(r.step, row0, kRows, lds.in_s, 8);           // (S <= 6 entries per lane; the slab reads only those)
    ns_hidden<128, 256>(r.actor, w, lds, g, row0, r.n, r.part, nullptr, nullptr);
}

// the same for a 256-thread workgroup: waves 0-1 take lane tile 2 wg, waves 2-3 lane tile 2 wg + 1 (two independent slab units
// with an NsLds each -- half the workgroups to dispatch for the same lanes, and no waves that leave at once; both halves run
// the same code, so the workgroup barriers inside ns_hidden line up; a half whose tile is past the range leaves as a whole)
template <class ENV>
__device__ __forceinline__ void ride_forward_pair(const RideArgs<ENV>& r, NsLds<128>* lds2, int wg, int g) {
    const int half = (int)threadIdx.x >> 7, row0 = r.lane0 + (2 * wg + half) * kRows;
    if (row0 >= r.lane1) return;
    NsLds<128>& lds = lds2[half];
    NsWeights<128> w;
    ns_load_weights<128, 256>(r.actor, g, w);
    ENV::stage_obs(r.step, row0, kRows, lds.in_s, 8, (int)threadIdx.x & (kNsThreads - 1));
    ns_hidden<128, 256>(r.actor, w, lds, g, row0, r.n, r.part, nullptr, nullptr);
}

// blocks `blk` of `nblk` (kThreads lanes each); smem: 4 * 11 floats
template <class ENV>
__device__ __forceinline__ void ride_tail(const RideArgs<ENV>& r, const CartConsts& cc, float* smem, unsigned blk, unsigned nblk) {
    constexpr int kStats = 10;
    const int tid = threadIdx.x, i = (int)blk * kThreads + tid;
    const long long t = r.step.ctrl[RPO_CTRL_T];
    float st[kStats + 1];
#pragma unroll
    for (int k = 0; k <= kStats; ++k) st[k] = 0.0f;
    if (i < r.n) {
        // everything the lane reads is requested up front: the step below is one dependent chain
        const RpoEpisode ep = ENV::episode(r.step, i);
        float obs[8];
        ENV::lane_obs(r.step, i, obs);
        float ap = ns_head(r.part, r.n, i, 0, r.actor.b1[0]);
        const float eps_t = fmaxf(r.act.eps_end, r.act.eps_start - r.act.eps_decay * (float)t);
        if (r.gauss) {
            const float rl = ns_head(r.part, r.n, i, 1, r.actor.b1b[0]);
            const rpo_u4 u = rpo_philox(r.act.seed, r.act.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_POLICY,
                                        (uint32_t)r.step.ctrl[RPO_CTRL_UPDATES]);
            ap = rpo_head_dev::gauss_head_row(ap, rl, rpo_normal(u.x, u.y), r.scale, r.base, r.act.box_lo, r.act.box_hi, 0, nullptr);
        } else {
            ap = r.scale * tanhf(ap) + r.base;
        }
        int k;
        const auto c = env_consts(cc, (ENV*)nullptr);
        const float2 a = ENV::project(r.act, c, obs, i, ap, eps_t, t, k);
        reinterpret_cast<float2*>(r.act.action)[i] = a;
        const long long ring_base = r.step.rows ? (t % r.step.cap_steps) * (long long)r.n : 0;
        float lane_st[kStats];
#pragma unroll
        for (int q = 0; q < kStats; ++q) lane_st[q] = 0.0f;
        ENV::lane(r.step, c, i, obs, a, ep, ring_base, lane_st);
#pragma unroll
        for (int q = 0; q < kStats; ++q) st[q] = lane_st[q];
        st[kStats] = (float)k;
    }
    if (r.step.stats) {
        // sums 0..7, maxima 8..9 (rollout_kernel's slots), then the projection-iteration sum; every block owns a sub-row
        float* srow = rpo_stats_row_at(r.step.stats, r.step.stats_cap, t, blk);
        const int lane = tid & (RPO_WAVE - 1), wave = tid / RPO_WAVE;
#pragma unroll
        for (int k = 0; k <= kStats; ++k) {                          // DPP inside the rows, v_readlane across them: no LDS round trips
            const bool mx = k == 8 || k == 9;
            float x = st[k];
            if (mx) {
                x = rpo_wave_max_nonneg(x);
            } else {
                x = rpo_row16_sum_desc_lane0(x);
                const int b = __float_as_int(x);
                x = (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
                    (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
            }
            if (lane == 0) smem[wave * (kStats + 1) + k] = x;
        }
        __syncthreads();
        if (tid <= kStats) {
            const int slot[kStats + 1] = {RPO_STAT_REWARD_SUM, RPO_STAT_EPISODES, RPO_STAT_RETURN_SUM, RPO_STAT_LENGTH_SUM,
                                          RPO_STAT_MAX_INEQ_SUM, RPO_STAT_MAX_EQ_SUM, RPO_STAT_VIOL_COUNT, RPO_STAT_TERMINATED,
                                          RPO_STAT_MAX_INEQ_MAX, RPO_STAT_MAX_EQ_MAX, RPO_STAT_PROJ_ITERS};
            const bool mx = tid == 8 || tid == 9;
            float v = smem[tid];
            for (int w = 1; w < kThreads / RPO_WAVE; ++w)
                v = mx ? fmaxf(v, smem[w * (kStats + 1) + tid]) : v + smem[w * (kStats + 1) + tid];
            if (mx) { if (v > 0.0f) rpo_atomic_max_nonneg(srow + slot[tid], v); }
            else if (v != 0.0f) atomicAdd(srow + slot[tid], v);
        }
    }
    if (!r.defer_clock) rpo_step_epilogue_at(r.step.ctrl, t, r.step.stats, r.step.stats_cap, blk, nblk);
}

// fwd_a / fwd_b + the actor forward of a lane range: grid (8 column groups, row tiles of the batch, roles + planes of lane
// tiles); the update's own workgroups come first in dispatch order
template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_fwd_a_ride_kernel(SplitArgs p, RideArgs<typename L::Env> r) {
    // 256-thread workgroups: the riders work in pairs (ride_forward_pair); the update's own roles use two of the four waves
    __shared__ NsLds<128> lds[2];
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    const int roles = (p.twin ? 2 : 1) + 1;
    const NsBlock nb = ns_block();
    if ((int)blockIdx.z < roles) {
        if (threadIdx.x < kNsThreads) fwd_a_role<L>(p, lds[0], tile, nb.tile * kRows, nb.g, blockIdx.z);
    } else {
        ride_forward_pair<typename L::Env>(r, lds, ((int)blockIdx.z - roles) * (int)gridDim.y + nb.tile, nb.g);
    }
}

template <class L, int PROJ>
__global__ __launch_bounds__(kNsThreads) void split_critic_fwd_b_ride_kernel(SplitArgs p, CartConsts c, RideArgs<typename L::Env> r) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    const int K = p.twin ? 2 : 1;
    const NsBlock nb = ns_block();
    if ((int)blockIdx.z < K) fwd_b_role<L, PROJ>(p, c, lds, tile, nb.tile * kRows, nb.g, blockIdx.z);
    else ride_forward<typename L::Env>(r, lds, ((int)blockIdx.z - K) * (int)gridDim.y + nb.tile, nb.g);
}

// ---- SpringPendulum: head of the policy + the reference's batch-coupled projection (pendulum.py:337-339), one workgroup
template <int LPS>
__global__ __launch_bounds__(1024) void split_pend_head_project_kernel(SplitArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int i = threadIdx.x / LPS;
    float ap = 0.0f;
    if (i < p.B) {
        float logp = 0.0f;
        ap = ns_policy_head(p, i, p.ctrl[RPO_CTRL_T], &logp);
        if (p.twin && threadIdx.x % LPS == 0) p.logp[i] = logp;
    }
    rpo_pend_dev::project_batchref_body<LPS>(p.B, i < p.B ? p.batch_out + (size_t)i * RPO_PEND_ROW + PendRow::NS_OFF : nullptr,
                                             ap, p.next_actions, p.proj_iters, p.max_steps, p.corr_lr, p.corr_eps,
                                             p.corr_momentum, lds);
}

// B <= 256: thread i < B takes the head of row i, then the register-tiled projection (project_batchref_wide)
__global__ __launch_bounds__(1024) void split_pend_head_project_wide_kernel(SplitArgs p) {
    __shared__ __attribute__((aligned(16))) float lds[rpo_pend_dev::kWideLds];
    const int i = threadIdx.x;
    float ap = 0.0f;
    if (i < p.B) {
        float logp = 0.0f;
        ap = ns_policy_head(p, i, p.ctrl[RPO_CTRL_T], &logp);
        if (p.twin) p.logp[i] = logp;
    }
    rpo_pend_dev::project_batchref_wide(p.B, p.batch_out + PendRow::NS_OFF, RPO_PEND_ROW, ap, p.next_actions, p.proj_iters,
                                        p.max_steps, p.corr_lr, p.corr_eps, p.corr_momentum, lds);
}

// ---- The same on ONE WORKGROUP PER ROW TILE (B <= 256, max_steps <= kPmMaxSteps): workgroup w owns the samples [16 w, 16 w + 16)
// -- 16 lanes per sample, the values j = 64 q + 4 c + m of project_batchref_wide, so the bits are its bits -- and the n^2
// predicate sum of a GRG iteration is 1/16 per compute unit (the one-workgroup form is bound by the vector ALU of ONE CU:
// 2.3 us per iteration).  Every iteration is an all-gather of the 256 dgp values (+ the rows' stop bits) between the
// workgroups, done with 8-byte {tag, value} granules in `ws`: the data IS the flag (a reader takes a granule only when it
// carries the tag of this launch and iteration), so no fences, no arrival counter, and nothing stale can be consumed
// whatever the placement.  Granule stores are agent-scope (write-through `sc1`) unless the workgroups find themselves on
// ONE XCD (they exchange HW_REG_XCC_ID through such granules first): then plain stores, which stay in that XCD's L2 where
// the `sc1` polling loads are served -- a speed choice only, taken inside the launch it applies to.
// ws (u64 words, zero before the first launch): [2][256] dgp granules (by iteration parity) | [16] XCC granules | epoch |
// gave-up flag | readers-done count | (fused front, below) [8][256][2] head-partial granules | [256][3] action granules.
// Tags are epoch * 32 + 1 (XCC, partials, actions) / + 2 + k (iteration k).  The epoch is advanced when every workgroup
// that reads it is done with it: by workgroup 0 here (it has then seen every other workgroup's last granule), by the last of
// the counted readers in the fused front.
constexpr int kPmRows = 16, kPmMaxSteps = 30, kPmSpinMax = 1 << 16;
constexpr int kPmXcc = 512, kPmEpoch = 528, kPmGaveUp = 529, kPmDone = 530, kPmPart = 544, kPmAct = kPmPart + 8 * 256 * 2;
constexpr int kPmWords = kPmAct + 256 * 3;
constexpr int kPmSmem = 6 * kPmRows + 2 * 256 + 8;   // floats: the rows' state | dgp (two buffers) | flags
#ifdef RPO_PM_TRACE   // development aid: 100 MHz timestamps of the fused front's phases behind the workspace (tools/probe_pfront.py)
#define PM_STAMP(ws, cond, slot) do { if ((cond) && threadIdx.x == 0) (ws)[kPmWords + (slot)] = wall_clock64(); } while (0)
#else
#define PM_STAMP(ws, cond, slot) do { } while (0)
#endif

__device__ __forceinline__ void pm_store(unsigned long long* g, unsigned long long v, bool local) {
    if (local) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(g), "v"(v) : "memory");
    else __hip_atomic_store(g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pm_load(unsigned long long* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pm_tag0(unsigned long long* ws) {
    return (unsigned long long)((unsigned)pm_load(ws + kPmEpoch) & 0x3ffffffu) * 32ull + 1ull;   // (31-bit tags)
}
// Poll `count` (<= blockDim) granules g[0 .. count) until each carries `tag` (bit 31 of the tag word is payload); returns the
// granule of this thread (0 beyond count).  `failed` (LDS) is raised when the wait was given up.
__device__ __forceinline__ unsigned long long pm_gather(unsigned long long* g, int count, unsigned long long tag, int* failed) {
    unsigned long long x = 0;
    const bool mine = (int)threadIdx.x < count;
    if ((int)(threadIdx.x & ~63u) < count) {                     // whole waves that hold at least one granule
        for (int spins = 0;; ++spins) {
            bool ok = true;
            if (mine) { x = pm_load(g + threadIdx.x); ok = ((x >> 32) & 0x7fffffffull) == tag; }
            if (__all(ok)) break;
            if (spins >= kPmSpinMax) { *failed = 1; x = 0; break; }
        }
    }
    return x;
}

// every projection workgroup announces its XCD as early as it can; pm_project reads the announcements
__device__ __forceinline__ void pm_publish_xcc(unsigned long long* ws, unsigned long long tag0, int wg) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pm_store(ws + kPmXcc + wg, (tag0 << 32) | (xcc & 15u), false);
    }
}

// The projection of the row tile `wg` (of T) by a 256-thread workgroup.  `ax0` / the observation of row r < 16 come from the
// caller through st[] (the rows' state: a_x | a_y | C_p | C_o | 1 / C_o | b, 16 each); lds: kPmSmem floats.  act_gran: NULL, or
// where the tile's projected actions (+ log pi) go as granules for consumers of the same launch.
__device__ __forceinline__ void pm_project(const SplitArgs& p, unsigned long long* ws, unsigned long long tag0, int store_mode, int wg,
                                           int T, float* lds, const float* logp_row, unsigned long long* act_gran) {
    using namespace rpo_pend_dev;
    const int tid = threadIdx.x, c = tid & 15, sl = tid >> 4, n = p.B;
    float* st = lds;
    float* dbuf = lds + 6 * kPmRows;
    int* flags = reinterpret_cast<int*>(lds + 6 * kPmRows + 2 * 256);   // [0] stop word, [1] plain stores allowed, [2] gave up
    if (tid >= 64 && tid < 128) {                                // one wave: are the workgroups on one XCD? (pm_publish_xcc)
        const int l = tid - 64;
        unsigned long long x = 0;
        bool ok = true;
        for (int spins = 0;; ++spins) {
            if (l < T) { x = pm_load(ws + kPmXcc + l); ok = (x >> 32) == tag0; }
            if (__all(ok)) break;
            if (spins >= kPmSpinMax) { if (l == 0) flags[2] = 1; break; }
        }
        const unsigned mine = (unsigned)x & 15u, first = (unsigned)__builtin_amdgcn_readfirstlane((int)mine);
        const bool same = __all(l >= T || (ok && mine == first));
        if (l == 0) flags[1] = (store_mode == 2 || (store_mode == 1 && same)) ? 1 : 0;
    }
    __syncthreads();
    PM_STAMP(ws, wg == 0, 4);
    const bool local = flags[1] != 0;
    PbSample sm;
    pb_init(sm, st[sl], st[kPmRows + sl], st[2 * kPmRows + sl], st[3 * kPmRows + sl], st[4 * kPmRows + sl], st[5 * kPmRows + sl]);
    const int me = wg * kPmRows + sl;                            // this thread's sample (16 lanes each)
    int k = 0;
    for (; k < p.max_steps; ++k) {
        const unsigned long long tag = tag0 + 1ull + (unsigned long long)k;
        unsigned long long* gk = ws + (k & 1) * 256;
        float dg;
        const bool viol = pb_pre(sm, p.corr_eps, dg);
        if (c == 0)                                              // granule: stop bit | 31-bit tag | dgp
            pm_store(gk + me, ((unsigned long long)(viol ? 1u : 0u) << 63) | (tag << 32) |
                                  (unsigned long long)__float_as_uint(me < n ? dg : 0.0f), local);
        float* dk = dbuf + (k & 1) * 256;
        {                                                        // gather the granules of this iteration (one per thread)
            const unsigned long long x = pm_gather(gk, T * kPmRows, tag, flags + 2);
            dk[tid] = __uint_as_float((unsigned)x);
            if (__any((x >> 63) != 0ull)) flags[0] = k + 1;      // (every writer of this iteration stores the same value)
        }
        __syncthreads();
        if ((k > 0 && flags[0] < k + 1) || flags[2]) break;      // batch-global stop test, rpo_ddpg.py:271-272
        float dv[16];
        pb_load16(dk, c, dv);
        const float grad = rpo_row16_allsum(pb_partial(sm.ax, pb_bgp(sm), dv));
        pb_step(sm, grad, p.corr_lr, p.corr_momentum);
        PM_STAMP(ws, wg == 0, 5 + k);
    }
    PM_STAMP(ws, wg == 0, 36);
    if (c == 0 && me < n) reinterpret_cast<float2*>(p.next_actions)[me] = make_float2(sm.ax, sm.ay);
    if (act_gran && c < 3) {
        const float v = c == 0 ? sm.ax : c == 1 ? sm.ay : logp_row[sl];
        pm_store(act_gran + me * 3 + c, (tag0 << 32) | (unsigned long long)__float_as_uint(v), false);
    }
    if (wg == 0 && tid == 0 && p.proj_iters) *p.proj_iters = k;
    if (tid == 0 && flags[2]) __hip_atomic_store(ws + kPmGaveUp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the rows' state for pm_project: row r of the tile has the basic action ax and the observation o
__device__ __forceinline__ void pm_stage_row(float* st, int r, bool live, float ax, const float* o) {
    using namespace rpo_pend_dev;
    Eq e = {0.0f, 1.0f, 1.0f, 0.0f};
    float ay = 0.0f;
    if (live) {
        e = set_eq(o[0], o[1], o[2], o[3], o[4]);
        ay = pb_complete(e, ax);
    } else {
        ax = 0.0f;
    }
    st[r] = ax; st[kPmRows + r] = ay; st[2 * kPmRows + r] = e.C_p; st[3 * kPmRows + r] = e.C_o;
    st[4 * kPmRows + r] = e.C_o_inv; st[5 * kPmRows + r] = e.b;
}

// stand-alone launch: grid 8 T blocks, block b runs on XCD b % 8 -> the T workers are blocks 0, 8, ..
// (ap_in != NULL: the basic actions are given and the observations are rows of `obs` -- rpo_pendulum_project_batchref_ws)
__global__ __launch_bounds__(kThreads) void split_pend_head_project_multi_kernel(SplitArgs p, unsigned long long* ws, int store_mode,
                                                                                 const float* ap_in, const float* obs, int obs_stride) {
    if (blockIdx.x & 7) return;
    const int wg = blockIdx.x >> 3, T = gridDim.x >> 3, tid = threadIdx.x;
    __shared__ __attribute__((aligned(16))) float lds[kPmSmem];
    int* flags = reinterpret_cast<int*>(lds + 6 * kPmRows + 2 * 256);
    if (tid == 0) { flags[0] = 0; flags[2] = 0; }
    const unsigned long long tag0 = pm_tag0(ws);
    pm_publish_xcc(ws, tag0, wg);
    if (tid < kPmRows) {                                         // head of the policy for the workgroup's rows, Complete
        const int i = wg * kPmRows + tid;
        float ax = 0.0f;
        const float* o = nullptr;
        if (i < p.B) {
            if (ap_in) {
                ax = ap_in[i];
                o = obs + (size_t)i * obs_stride;
            } else {
                float logp = 0.0f;
                ax = ns_policy_head(p, i, p.ctrl[RPO_CTRL_T], &logp);
                if (p.twin) p.logp[i] = logp;
                o = p.batch_out + (size_t)i * RPO_PEND_ROW + PendRow::NS_OFF;
            }
        }
        pm_stage_row(lds, tid, i < p.B, ax, o);
    }
    pm_project(p, ws, tag0, store_mode, wg, T, lds, nullptr, nullptr);
    if (wg == 0 && tid == 0)                                     // (workgroup 0 has seen every other workgroup's last granule)
        __hip_atomic_store(ws + kPmEpoch, ((tag0 - 1ull) >> 5) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- TD target + Huber for row i of critic k from the slab partials (rpo_ddpg.py:331-335, rpo_sac.py:346-353)
template <class L>
__device__ __forceinline__ float ns_td_row(const SplitArgs& p, int k, int i, float* hub) {
    const float q = ns_head(p.part_q[k], p.B, i, 0, p.critic[k].b1[0]);
    const float qn1 = ns_head(p.part_qn[0], p.B, i, 0, p.critic_target[0].b1[0]);
    const float qn2 = p.twin ? ns_head(p.part_qn[1], p.B, i, 0, p.critic_target[1].b1[0]) : 0.0f;
    const float qn = rpo_head_dev::td_next_value(qn1, qn2, p.twin, p.twin ? p.logp[i] : 0.0f, p.twin, p.alpha);
    const float* row = p.batch_out + (size_t)i * L::ROW;
    const float y = rpo_head_dev::td_target(row[L::R_OFF], row[L::R_OFF + 1], p.gamma, qn);
    return rpo_head_dev::td_huber_row(q, y, 1.0f / (float)p.B, hub);
}

constexpr int kBwdMaxB = 1024;

// dh[b][j] = d loss / d h1 of row b, hidden column j, from the rows' head gradients (d0 [, d1]): formed where it is needed
__device__ __forceinline__ float ns_dh(float h, float d0, float d1, float w1a, float w1b) {
    return (h > 0.0f) ? fmaf(d1, w1b, d0 * w1a) : 0.0f;
}

// ---- Weight roles of a backward pass.  A workgroup's operand stream comes from L2 at ~50-100 GB/s, so the tiles are
//      kept small: the 16 x 64 dW0 tiles of the row-tile kernels need 80 KB each (~6 us measured for the role), 16 x 16
//      tiles 32 KB.  Summation orders are those of mlp_bwd_weights_body (batch split in 4 ranges, each a sequential chain,
//      the 4 sums added in order), so the gradients are bitwise the same.
constexpr int kW0Tiles = (256 / 16) * (128 / 16);   // 128 dW0 tiles of 16 hidden rows x 16 input columns
constexpr int kHvBlocks = 256 / 16;                 // 16 blocks of 16 hidden columns: db0, dW1 (, dW1b), block 0 also db1
constexpr int kWeightBlocks = kW0Tiles + kHvBlocks;

// dW0 tile: wave w sums the batch range [w, w + 1) * B / 4 (k-steps of 4 rows), dh formed on the fly.
__device__ __forceinline__ float ns_dw0_tile(const Mlp& net, float* gW0, const float* __restrict__ h1,
                                             const float* __restrict__ x0, const float* dglob, int dstride, bool two, int B,
                                             int blk, float* smem) {
    constexpr int EIN = 128, H = 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    // block -> tile so that block b (XCD b % 8) works on hidden columns [32 (b % 8), +32): the two column tiles that share
    // h1's 128-byte lines; an XCD's L2 then fetches 1/8 of h1 (32 KB) + all of x0 (131 KB) instead of all of h1 + 1/8 of x0
    const int xq = blk & 7, yq = blk >> 3, jt = 2 * xq + (yq >> 3), et = yq & 7;
    static_assert(EIN / 16 == 8 && H / 16 == 16, "tile decode");
    const int j = jt * 16 + li, e = et * 16 + li;
    const int nk = (B + 3) / 4, ks_lo = (nk * wave) / 4, ks_hi = (nk * (wave + 1)) / 4, last = B - 1;
    // everything is requested up front, straight-line (rows past the wave's range are clamped and zeroed below)
    float hv[16], xv[16], d0[16], d1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int bb = (ks_lo + u) * 4 + lg, bc = bb < last ? bb : last;
        hv[u] = h1[(size_t)bc * H + j];
        xv[u] = x0[(size_t)bc * EIN + e];
        d0[u] = dglob[(size_t)bc * dstride];
        d1[u] = two ? dglob[(size_t)bc * dstride + 1] : 0.0f;
    }
    const float w1a = net.W1[j], w1b = two ? net.W1b[j] : 0.0f;
    float* dst = &gW0[(size_t)(jt * 16 + (tid >> 4)) * EIN + et * 16 + (tid & 15)];
    const float cur = *dst;
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int bb = (ks_lo + u) * 4 + lg;
        float av = ns_dh(hv[u], d0[u], d1[u], w1a, w1b);
        if (bb > last || ks_lo + u >= ks_hi) av = 0.0f;
        if (ks_lo + u < ks_hi || u == 0) acc = mfma4(av, fmaxf(xv[u], 0.0f), acc);
    }
    for (int ks = ks_lo + 16; ks < ks_hi; ++ks) {                  // batches beyond 256 rows
        const int bb = ks * 4 + lg, bc = bb < last ? bb : last;
        float av = ns_dh(h1[(size_t)bc * H + j], dglob[(size_t)bc * dstride], two ? dglob[(size_t)bc * dstride + 1] : 0.0f, w1a, w1b);
        if (bb > last) av = 0.0f;
        acc = mfma4(av, fmaxf(x0[(size_t)bc * EIN + e], 0.0f), acc);
    }
    // acc[i] = this wave's share of dW0[j = 16 jt + 4 lg + i][e = 16 et + li]
#pragma unroll
    for (int i = 0; i < 4; ++i) smem[(wave * 16 + lg * 4 + i) * 16 + li] = acc[i];
    __syncthreads();
    const int r = tid >> 4, c = tid & 15;
    const float nv = cur + (((smem[(0 * 16 + r) * 16 + c] + smem[(1 * 16 + r) * 16 + c]) + smem[(2 * 16 + r) * 16 + c]) +
                            smem[(3 * 16 + r) * 16 + c]);
    *dst = nv;
    return fabsf(nv);
}

// Hidden-layer vectors of 16 columns: thread = (column, batch range); 64 threads of the workgroup work, each a sequential
// chain over its B / 4 rows (the order of mlp_bwd_weights_body); block 0 also sums d_k over the batch for db1_k.
__device__ __forceinline__ float ns_hv_block(const Mlp& net, const MlpGrad& gr, const float* __restrict__ h1, const float* dglob,
                                             int dstride, bool two, int B, int rb, float* smem) {
    constexpr int H = 256;
    const int tid = threadIdx.x;
    float gmax = 0.0f;
    float (*partial)[3][16] = reinterpret_cast<float (*)[3][16]>(smem);       // [4][3][16]
    // what the gradients hold so far (+=) is requested NOW: read where it is added, at the end of the role's chain, it was one
    // more exposed round trip on the longest role of the launch (round 5)
    float cur_b0 = 0.0f, cur_w1 = 0.0f, cur_w1b = 0.0f, cur_b1 = 0.0f, cur_b1b = 0.0f;
    if (tid < 16) {
        cur_b0 = gr.b0[rb * 16 + tid];
        cur_w1 = gr.W1[rb * 16 + tid];
        if (two) cur_w1b = gr.W1b[rb * 16 + tid];
    }
    if (rb == 0 && tid == 0) {
        cur_b1 = gr.b1[0];
        if (two) cur_b1b = gr.b1b[0];
    }
    if (tid < 64) {
        const int o = tid & 15, part = tid >> 4;
        const int b_lo = (int)(((long long)B * part) / 4), b_hi = (int)(((long long)B * (part + 1)) / 4);
        const int j = rb * 16 + o;
        const float w1a = net.W1[j], w1b = two ? net.W1b[j] : 0.0f;
        float gb0 = 0.0f, gw1a = 0.0f, gw1b = 0.0f;
        // scalar heads (the critics): 64 rows of loads in flight -- the whole range of a part at batch 256 in ONE round trip
        // (these 16 blocks are the longest role of bwd_b: leaving them out made the launch 1.3 us shorter, the 128 dW0 tiles
        // 0.1 us); same sequential chain over the rows
        for (int c0 = b_lo; !two && c0 < b_hi; c0 += 64) {
            float hv[64], d0[64];
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const int bb = c0 + u < b_hi ? c0 + u : b_hi - 1;
                hv[u] = h1[(size_t)bb * H + j];
                d0[u] = dglob[(size_t)bb * dstride];
            }
#pragma unroll
            for (int u = 0; u < 64; ++u) {
                const bool ok = c0 + u < b_hi;
                const float h = ok ? hv[u] : 0.0f, a0 = ok ? d0[u] : 0.0f;
                gb0 += ns_dh(h, a0, 0.0f, w1a, w1b);
                const float hr = fmaxf(h, 0.0f);
                gw1a = fmaf(a0, hr, gw1a);
                gw1b = fmaf(0.0f, hr, gw1b);
            }
        }
        for (int c0 = b_lo; two && c0 < b_hi; c0 += 32) {          // 32 rows of loads in flight, straight-line inside
            float hv[32], d0[32], d1[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int bb = c0 + u < b_hi ? c0 + u : b_hi - 1;
                hv[u] = h1[(size_t)bb * H + j];
                d0[u] = dglob[(size_t)bb * dstride];
                d1[u] = two ? dglob[(size_t)bb * dstride + 1] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const bool ok = c0 + u < b_hi;
                const float h = ok ? hv[u] : 0.0f, a0 = ok ? d0[u] : 0.0f, a1 = ok ? d1[u] : 0.0f;
                gb0 += ns_dh(h, a0, a1, w1a, w1b);
                const float hr = fmaxf(h, 0.0f);
                gw1a = fmaf(a0, hr, gw1a);
                gw1b = fmaf(a1, hr, gw1b);
            }
        }
        partial[part][0][o] = gb0; partial[part][1][o] = gw1a; partial[part][2][o] = gw1b;
    }
    __syncthreads();
    if (tid < 16) {
        const int o = tid, j = rb * 16 + o;
        const float nb0 = cur_b0 + (((partial[0][0][o] + partial[1][0][o]) + partial[2][0][o]) + partial[3][0][o]);
        const float nw1 = cur_w1 + (((partial[0][1][o] + partial[1][1][o]) + partial[2][1][o]) + partial[3][1][o]);
        gr.b0[j] = nb0;
        gr.W1[j] = nw1;
        gmax = fmaxf(fabsf(nb0), fabsf(nw1));
        if (two) {
            const float nw1b = cur_w1b + (((partial[0][2][o] + partial[1][2][o]) + partial[2][2][o]) + partial[3][2][o]);
            gr.W1b[j] = nw1b;
            gmax = fmaxf(gmax, fabsf(nw1b));
        }
    }
    if (rb == 0) {
        // db1_k = sum_b d_k[b]: strided per-thread partials, wave reduction, 4 wave partials added in order
        __syncthreads();
        float s0 = 0.0f, s1 = 0.0f;
        for (int b2 = tid; b2 < B; b2 += kThreads) {
            s0 += dglob[(size_t)b2 * dstride];
            if (two) s1 += dglob[(size_t)b2 * dstride + 1];
        }
        { float ss[2] = {s0, s1}; rpo_wave_reduce_many(ss, 0u); s0 = ss[0]; s1 = ss[1]; }   // (both butterflies step by step)
        float* red = smem + 256;
        if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s0; red[(tid >> 6) * 2 + 1] = s1; }
        __syncthreads();
        if (tid == 0) {
            const float nb1 = cur_b1 + (((red[0] + red[2]) + red[4]) + red[6]);
            gr.b1[0] = nb1;
            gmax = fmaxf(gmax, fabsf(nb1));
            if (two) {
                const float nb1b = cur_b1b + (((red[1] + red[3]) + red[5]) + red[7]);
                gr.b1b[0] = nb1b;
                gmax = fmaxf(gmax, fabsf(nb1b));
            }
        }
    }
    return gmax;
}

// block w < kW0Tiles: dW0 tile, else hidden-vector block.  The rows' head gradients d0 [, d1] were left in global memory by
// the dx0 launch before this one (row stride `dstride` floats).
__device__ __forceinline__ float ns_weight_role(const Mlp& net, const MlpGrad& gr, const float* h1, const float* x0,
                                                const float* dglob, int dstride, bool two, int B, int w, float* smem) {
    if (w < kW0Tiles) return ns_dw0_tile(net, gr.W0, h1, x0, dglob, dstride, two, B, w, smem);
    return ns_hv_block(net, gr, h1, dglob, dstride, two, B, w - kW0Tiles, smem);
}

// ---- bwd_a: TD target + Huber (from the slab partials) -> dh = dLoss/dQ * W1 * 1[h1 > 0] (on the fly) -> dx0 of one
//      (critic k, row tile, 16 first-layer columns); the column-group-0 workgroup of a tile leaves dLoss/dQ and the loss share.
//      blocks (critic, column group, row tile), 256 threads -- the ROW TILE is the fastest index here: block b lands on XCD
//      b % 8, and what the 8 column-group blocks of a tile share is the tile's h1 rows (16 KB each, the launch's largest
//      operand): with the tile fastest an XCD's L2 fetches the rows of its own tiles once instead of every XCD fetching all.
constexpr int kBwdASmem = kRows * (256 + 4) + 16 + 4 * 16 * 16;

template <class L>
__device__ __forceinline__ void bwd_a_role(const SplitArgs& p, float* smem, int b, unsigned fused_consumers = 0u,
                                           unsigned fused_need = 0u, unsigned fused_need_critics = 0u) {
    constexpr int EIN = 128, H = 256, LDH = H + 4;
    const int T = (p.B + kRows - 1) / kRows, B = p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    {
        const int k = b / (T * kNsGroups), rem = b - k * T * kNsGroups, g = rem / T, tile = rem - g * T;   // row tile fastest: see bwd_a
        const int row0 = tile * kRows;
        const Mlp& net = p.critic[k];
        float* dh_s = smem;                                       // [16][LDH]
        float* dq_s = smem + kRows * LDH;                         // [16]
        float* wpart = dq_s + 16;                                 // [4][16][16]
        // operands that depend on nothing are requested first: this wave's W0 column slice and the tile's h1 column
        const int jw = wave * (H / 4), e = g * 16 + li;
        float wv[H / 16], hv[kRows];
#pragma unroll
        for (int ks = 0; ks < H / 16; ++ks) wv[ks] = net.W0[(size_t)(jw + ks * 4 + lg) * EIN + e];
        const float w1a = net.W1[tid];                            // 256 threads = 256 hidden columns
        // fused front: the tile's critic slabs of THIS launch (fwd_a's critic planes: the high count of word 1) -- their saved
        // activations are requested while fwd_b's workgroups are still at work
        if (fused_consumers && fused_need_critics)
            ns_tile_wait(p.tile_sync + (T + tile) * kNsSyncStride, fused_need_critics, p.tile_sync + 3 * T * kNsSyncStride, 16);
#pragma unroll
        for (int r = 0; r < kRows; ++r) hv[r] = row0 + r < B ? p.h1[k][(size_t)(row0 + r) * H + tid] : 0.0f;
        const size_t xo = (size_t)(row0 + (tid >> 4)) * EIN + g * 16 + (tid & 15);
        const float x0v = row0 + (tid >> 4) < B ? p.x0[k][xo] : 0.0f;   // mask of this thread's output, requested early
        if (fused_consumers) {
            // fused front / mid: the tile's target slabs (fwd_b) of THIS launch, 8 arrivals per network (the low count)
            ns_tile_wait(p.tile_sync + (T + tile) * kNsSyncStride, fused_need, p.tile_sync + 3 * T * kNsSyncStride);
            ns_tile_passed(p.tile_sync, T, tile, fused_consumers);
        }
        if (tid < 64) {
            float dq = 0.0f, hub = 0.0f;
            if (tid < kRows && row0 + tid < B) {
                dq = ns_td_row<L>(p, k, row0 + tid, &hub);
                if (g == 0) p.dq[k][row0 + tid] = dq;
            }
            if (tid < kRows) dq_s[tid] = dq;
            const float sum = rpo_row16_sum_desc_lane0(hub);     // (== rpo_wave_sum: only lanes 0..15 hold terms)
            if (tid == 0 && g == 0) p.loss_partial[k * T + tile] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRows; ++r) dh_s[r * LDH + tid] = (hv[r] > 0.0f) ? dq_s[r] * w1a : 0.0f;
        __syncthreads();
        f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < H / 16; ++ks) acc = mfma4(dh_s[li * LDH + jw + ks * 4 + lg], wv[ks], acc);
        // acc[i] = partial dx0[row = 4 lg + i][e = 16 g + li] of this wave's quarter of the hidden columns
#pragma unroll
        for (int i = 0; i < 4; ++i) wpart[(wave * 16 + lg * 4 + i) * 16 + li] = acc[i];
        __syncthreads();
        {
            const int r = tid >> 4, ee = tid & 15;
            float v = wpart[(0 * 16 + r) * 16 + ee];               // waves added in the order of mlp_bwd_rows_body
            v = wpart[(1 * 16 + r) * 16 + ee] + v;
            v = wpart[(2 * 16 + r) * 16 + ee] + v;
            v = wpart[(3 * 16 + r) * 16 + ee] + v;
            if (row0 + r < B) p.dx0[k][xo] = x0v > 0.0f ? v : 0.0f;
        }
    }
}

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_bwd_a_kernel(SplitArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[kBwdASmem];
    bwd_a_role<L>(p, smem, blockIdx.x);
}

// ---- bwd_b: every parameter gradient of critic k = blockIdx.y.  blocks [0, 36): dW0 tiles and hidden-layer vectors from
//      dLoss/dQ (left by bwd_a) and the saved activations, dh formed on the fly; then the first-layer gradients dWs / dbs /
//      dWa / dba from dx0 (batch reduction, one owner per output, fixed order).  Leaves the inf-norm of what it wrote.
// Bookkeeping for the optimiser launch that follows bwd_b ("prepared" rpo_adam_step): one thread advances the step counter
// and leaves the bias corrections of that step -- nothing in this launch reads them -- and, when the critic step is the
// last optimiser launch of the iteration, the update clock (bwd_a / bwd_b do not read it).  The Adam launch then has no
// bookkeeping left and needs no last-workgroup detection (two atomic round trips, 2.3 us).
__device__ __forceinline__ void adam_prepare(int* step_dev, float beta1, float beta2) {
    const int s = step_dev[0] + 1;
    double* cache = reinterpret_cast<double*>(step_dev + 4);
    cache[0] = 1.0 - pow((double)beta1, (double)s);
    cache[1] = sqrt(1.0 - pow((double)beta2, (double)s));
    step_dev[1] = s;
    step_dev[0] = s;
}

__device__ __forceinline__ void bwd_b_bookkeeping(const SplitArgs& p) {
    if (threadIdx.x != 0) return;
    if (p.prep_step) adam_prepare(p.prep_step, p.prep_beta1, p.prep_beta2);
    if (p.clock_out) p.clock_out[0] += 1;
    if (p.updates_out) p.updates_out[RPO_CTRL_UPDATES] += 1;
}

// own blocks: [0, kWeightBlocks) weight roles | first-layer blocks | (k == 0, when asked for) one bookkeeping block
template <class L>
__device__ __forceinline__ void bwd_b_role(const SplitArgs& p, float* smem, int bx, int k, int own_blocks) {
    if ((p.prep_step || p.clock_out || p.updates_out) && bx == own_blocks - 1) {
        if (k == 0) bwd_b_bookkeeping(p);
        return;
    }
    if (bx < kWeightBlocks) {
        gradmax_flush(p.gradmax, ns_weight_role(p.critic[k], p.critic_grad[k], p.h1[k], p.x0[k], p.dq[k], 1, false, p.B, bx, smem));
        return;
    }
    BwdArgs a{};
    a.net = p.critic[k];
    a.g = p.critic_grad[k];
    a.n = p.B;
    a.s = p.batch_out; a.s_stride = L::ROW;
    a.a = p.batch_out + L::A_OFF; a.a_stride = L::ROW;
    a.dx0 = p.dx0[k];
    a.param_grads = 1;
    a.first_layer_state_only = 0;
    gradmax_flush(p.gradmax, mlp_bwd_first_layer<128>(a, bx - kWeightBlocks));
}

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_bwd_b_kernel(SplitArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 16];
    bwd_b_role<L>(p, smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// bwd_b + explore / project / step / scatter of the lanes: grid (own_blocks + lane blocks, K); the extra x-blocks of plane 0
// step 256 lanes each (those of plane 1 leave at once)
template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_bwd_b_ride_kernel(SplitArgs p, CartConsts c, RideArgs<typename L::Env> r,
                                                                           int own_blocks) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 16];
    if ((int)blockIdx.x < own_blocks) bwd_b_role<L>(p, smem, blockIdx.x, blockIdx.y, own_blocks);
    else if (blockIdx.y == 0) ride_tail<typename L::Env>(r, c, smem, blockIdx.x - own_blocks, gridDim.x - own_blocks);
}

}  // namespace

extern "C" {

int rpo_mlp_split_supported(const rpo_mlp* net) { return net && split_ok(to_dev(net)) ? 1 : 0; }

int rpo_mlp_forward_split(int k, const rpo_mlp* const* nets, int n, const float* const* s, const int* s_stride,
                          const float* const* a, const int* a_stride, float* const* part, float* const* x0_save,
                          float* const* h1_save, void* stream) {
    if (k < 1 || k > 4 || n <= 0 || !nets || !s || !part) return RPO_ERR_ARG;
    SplitFwdArgs4 all{};
    for (int i = 0; i < k; ++i) {
        if (!nets[i] || !s[i] || !part[i]) return RPO_ERR_NULL;
        const Mlp m = to_dev(nets[i]);
        if (!split_ok(m) || (m.A > 0 && (!a || !a[i]))) return RPO_ERR_ARG;
        all.net[i] = SplitFwdArgs{m, n, s[i], s_stride[i], m.A > 0 ? a[i] : nullptr, m.A > 0 ? a_stride[i] : 0, part[i],
                                  x0_save ? x0_save[i] : nullptr, h1_save ? h1_save[i] : nullptr};
    }
    hipLaunchKernelGGL(mlp_forward_split_kernel, dim3(kNsGroups, (n + kRows - 1) / kRows, k), dim3(kNsThreads), 0,
                       (hipStream_t)stream, all);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_mlp_split_head(const rpo_mlp* net, int n, const float* part, float* out, int out_mode, float scale, float base,
                       void* stream) {
    if (!net || !part || !out || n <= 0) return RPO_ERR_NULL;
    const Mlp m = to_dev(net);
    if (!split_ok(m)) return RPO_ERR_ARG;
    hipLaunchKernelGGL(mlp_split_head_kernel, dim3((n * m.n_out + RPO_BLOCK - 1) / RPO_BLOCK), dim3(RPO_BLOCK), 0,
                       (hipStream_t)stream, m, n, part, out, out_mode, scale, base);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

namespace {

// ======================================================================================== column-split policy step
// Env policies of the actor loss: Complete, the Lagrangian row and the backward of Complete (rpo_ddpg.py:307-324).
struct CartPol {
    typedef CartRow L;
    static constexpr int NI = 6;
    __device__ static __forceinline__ float2 complete(const CartConsts& c, const float* obs, int i, float ap, long long t) {
        rpo_cart_dev::ActArgs a{};
        a.noise_mode = RPO_NOISE_NONE; a.max_steps = 0;
        int k;
        return rpo_cart_dev::cart_explore_project(a, c, i, ap, 0.0f, t, k);
    }
    __device__ static __forceinline__ float lagr(const CartConsts& c, float a0, float a1, const float* nu_p, float scale,
                                                 float (&dist)[6], float2& g) {
        float nu[6], loss, g0, g1;
#pragma unroll
        for (int j = 0; j < 6; ++j) nu[j] = nu_p[j];
        rpo_cart_dev::lagrangian_row(c, a0, a1, nu, loss, dist, g0, g1);
        g = make_float2(scale * g0, scale * g1);
        return loss;
    }
    __device__ static __forceinline__ float complete_bwd(const CartConsts& c, const float* obs, float g0, float g1) {
        return rpo_cart_dev::complete_bwd_row(c, g0, g1);
    }
};
struct PendPol {
    typedef PendRow L;
    static constexpr int NI = 1;
    __device__ static __forceinline__ float2 complete(const CartConsts&, const float* obs, int i, float ap, long long t) {
        rpo_pend_dev::ActArgs a{};
        a.noise_mode = RPO_NOISE_NONE; a.max_steps = 0;
        int k;
        return rpo_pend_dev::pend_explore_project(a, obs, i, ap, 0.0f, t, k);
    }
    __device__ static __forceinline__ float lagr(const CartConsts&, float a0, float a1, const float* nu_p, float scale,
                                                 float (&dist)[6], float2& g) {
#pragma unroll
        for (int j = 1; j < 6; ++j) dist[j] = 0.0f;
        return rpo_pend_dev::lagrangian_row(a0, a1, nu_p[0], scale, dist[0], g.x, g.y);
    }
    __device__ static __forceinline__ float complete_bwd(const CartConsts&, const float* obs, float g0, float g1) {
        return rpo_pend_dev::complete_bwd_row(obs, g0, g1);
    }
};

// take_action's exploration (agent/ddpg_pa.py:108-110): the arithmetic of cart/pend_explore_project's RPO_NOISE_EXPLICIT
__device__ __forceinline__ float explore_clip(float ap_det, float eps_t, float e, float lo, float hi) {
    RPO_FP_STRICT
    return rpo_clamp(ap_det + eps_t * e, lo, hi);
}

// ---- pol_a: pi hidden slabs on the batch states, pre-activations saved.  grid (row tiles, 8)
__device__ __forceinline__ float* pol_part(const SplitArgs& p) { return p.part_pol ? p.part_pol : p.part_pi; }

template <class L>
__device__ __forceinline__ void pol_a_role(const SplitArgs& p, NsLds<128>& lds, float4* tile, int row0, int g,
                                           unsigned* wait_word = nullptr, unsigned wait_need = 0u) {
    NsWeights<128> w;
    ns_load_weights<128, 256>(p.actor, g, w);
    if (wait_word) {                                             // fused front: the tile's gathered rows of THIS launch
        const int T = (p.B + kRows - 1) / kRows;
        ns_tile_wait(wait_word, wait_need, p.tile_sync + 3 * T * kNsSyncStride);
    }
    ns_load_tile<L>(p, tile, row0);
    __syncthreads();
    ns_stage<L>(lds, reinterpret_cast<const float*>(tile), false, false);
    ns_hidden<128, 256>(p.actor, w, lds, g, row0, p.B, pol_part(p), p.x0_a, p.h1_a);
}

template <class L>
__global__ __launch_bounds__(kNsThreads) void split_policy_a_kernel(SplitArgs p) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    const NsBlock nb = ns_block();
    pol_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g);
}

// fwd_b + pol_a: on a policy iteration of a configuration WITHOUT a shared state embedding the policy slabs on the batch
// states need nothing the critic update produces (the batch was gathered by fwd_a, the critic step does not touch the
// actor), so they are an extra plane of fwd_b's grid instead of a launch of their own behind the critic step.  Their head
// partials go to part_pol: fwd_b's own prologue still reads part_pi.
template <class L, int PROJ>
__global__ __launch_bounds__(kNsThreads) void split_critic_fwd_b_pol_kernel(SplitArgs p, CartConsts c) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * L::CH];
    const int K = p.twin ? 2 : 1;
    const NsBlock nb = ns_block();
    if ((int)blockIdx.z < K) fwd_b_role<L, PROJ>(p, c, lds, tile, nb.tile * kRows, nb.g, blockIdx.z);
    else pol_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g);
}

// ---- fused front of the critic update (CartSafe): fwd_a, fwd_b and bwd_a in ONE launch -- the three stages whose hand-overs
//      stay inside a row tile.  grid (8, T, planes) of 256-thread workgroups (bwd_a's shape; the forward roles use two of the
//      four waves, the other two leave at once):
//        planes [0, 1 + K)          fwd_a's roles; the policy arrives at the tile's word 0, the critics at word 1 (high count)
//        planes [1 + K, 1 + 2 K)    fwd_b's target critics: wait for word 0 == 8, arrive at word 1
//        planes [1 + 2 K, 1 + 3 K)  bwd_a of critic k: behind its W0 requests waits for the critics' 8 K arrivals, requests the
//                                   saved activations, then waits for fwd_b's 8 K arrivals (word 1, low count)
//        plane  1 + 3 K (pol = 1)   pol_a of a policy iteration without a shared embedding (see fwd_b_pol): waits for word 0
//      Same code, same values as the launches it replaces.  Returns false for planes beyond its own.
template <class L>
__device__ __forceinline__ bool front_role(const SplitArgs& p, const CartConsts& c, float* smem, int pol) {
    const int K = p.twin ? 2 : 1, T = (int)gridDim.y, z = (int)blockIdx.z;
    if (z > 3 * K + pol) return false;
    const unsigned consumers = (unsigned)(kNsGroups * (2 * K + pol));
    if (z >= 1 + 2 * K && z < 1 + 3 * K) {
        bwd_a_role<L>(p, smem, (z - 1 - 2 * K) * T * kNsGroups + (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, consumers,
                      (unsigned)(kNsGroups * K), (unsigned)(kNsGroups * K));
        return true;
    }
    if (threadIdx.x >= kNsThreads) return true;                  // (whole waves: the barriers below count the two that stay)
    NsLds<128>& lds = *reinterpret_cast<NsLds<128>*>(smem);
    float4* tile = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + sizeof(NsLds<128>));
    const NsBlock nb = ns_block();
    unsigned* sync = p.tile_sync;
    if (z < 1 + K) {
        fwd_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g, z);
        if (z == 0) { if (!((p.debug & 1) && nb.tile == 0 && nb.g == 0)) ns_tile_arrive(sync + nb.tile * kNsSyncStride); }
        else ns_tile_arrive(sync + (T + nb.tile) * kNsSyncStride, 1u << 16);
    } else if (z < 1 + 2 * K) {
        fwd_b_role<L, 1>(p, c, lds, tile, nb.tile * kRows, nb.g, z - 1 - K, sync + nb.tile * kNsSyncStride, (unsigned)kNsGroups);
        ns_tile_arrive(sync + (T + nb.tile) * kNsSyncStride);
        ns_tile_passed(sync, T, nb.tile, consumers);
    } else {
        pol_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g, sync + nb.tile * kNsSyncStride, (unsigned)kNsGroups);
        ns_tile_passed(sync, T, nb.tile, consumers);
    }
    return true;
}

constexpr int kFrontSmem = kBwdASmem > (int)(2 * sizeof(NsLds<128>) / sizeof(float)) ? kBwdASmem : (int)(2 * sizeof(NsLds<128>) / sizeof(float));
static_assert(sizeof(NsLds<128>) + sizeof(float4) * kRows * 6 <= sizeof(float) * kFrontSmem, "forward roles fit");
static_assert(sizeof(NsLds<128>) % 16 == 0, "tile alignment");

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_front_kernel(SplitArgs p, CartConsts c, int pol) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    front_role<L>(p, c, smem, pol);
}

// front + the actor forward of lanes [lane0, lane1) of the NEXT vector step (ride_forward) in the planes behind its own
template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_front_ride_kernel(SplitArgs p, CartConsts c, RideArgs<typename L::Env> r) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    if (front_role<L>(p, c, smem, 0)) return;
    const NsBlock nb = ns_block();
    const int own = 1 + 3 * (p.twin ? 2 : 1);
    ride_forward_pair<typename L::Env>(r, reinterpret_cast<NsLds<128>*>(smem), ((int)blockIdx.z - own) * (int)gridDim.y + nb.tile, nb.g);
}

// ---- SpringPendulum: the batch-coupled projection (one workgroup, every row) sits between fwd_a and fwd_b, so only fwd_b and
//      bwd_a share a launch ("mid"): planes [0, K) fwd_b's target critics (arrive at word 1), [K, 2 K) bwd_a (waits for
//      word 1 == 8 K), plane 2 K (pol = 1) pol_a (the gathered rows are an earlier launch's: no wait).
template <class L>
__device__ __forceinline__ bool mid_role(const SplitArgs& p, const CartConsts& c, float* smem, int pol) {
    const int K = p.twin ? 2 : 1, T = (int)gridDim.y, z = (int)blockIdx.z;
    if (z >= 2 * K + pol) return false;
    if (z >= K && z < 2 * K) {
        bwd_a_role<L>(p, smem, (z - K) * T * kNsGroups + (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y,
                      (unsigned)(kNsGroups * K), (unsigned)(kNsGroups * K));
        return true;
    }
    if (threadIdx.x >= kNsThreads) return true;
    NsLds<128>& lds = *reinterpret_cast<NsLds<128>*>(smem);
    float4* tile = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + sizeof(NsLds<128>));
    const NsBlock nb = ns_block();
    if (z < K) {
        fwd_b_role<L, 0>(p, c, lds, tile, nb.tile * kRows, nb.g, z);
        ns_tile_arrive(p.tile_sync + (T + nb.tile) * kNsSyncStride);
    } else {
        pol_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g);
    }
    return true;
}

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_mid_kernel(SplitArgs p, CartConsts c, int pol) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    mid_role<L>(p, c, smem, pol);
}

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_mid_ride_kernel(SplitArgs p, CartConsts c, RideArgs<typename L::Env> r) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    if (mid_role<L>(p, c, smem, 0)) return;
    const NsBlock nb = ns_block();
    ride_forward_pair<typename L::Env>(r, reinterpret_cast<NsLds<128>*>(smem), ((int)blockIdx.z - 2 * (p.twin ? 2 : 1)) * (int)gridDim.y + nb.tile,
                                       nb.g);
}

// ---- fused front of the critic update (SpringPendulum): fwd_a, the batch-coupled projection, fwd_b and bwd_a in ONE launch.
//      The projection needs every row, so its hand-overs cross XCDs -- they are tagged granules (pm_* above: the data is the
//      flag, agent-scope stores and loads, nothing depends on placement); the row-tile-local hand-overs (critic slabs, saved
//      activations, fwd_b -> bwd_a) stay on the tile's XCD as in the CartSafe front.  grid (8, T, planes) x 256 threads:
//        planes [0, 1 + K)          fwd_a's roles: the policy on s' arrives at the tile's word 0 (its gathered rows are
//                                   published) and then leaves its head partials as granules; the critics arrive at word 1 (high)
//        plane  1 + K               blocks (0, y): the projection of row tile y -- gathers its own rows, polls the tile's 8 x 16
//                                   (x 2) partial granules, head -> Complete -> the GRG iterations with the other tiles' workgroups
//                                   (all T of them on XCD 0) -> the projected actions (+ log pi) as granules
//        planes [2 + K, 2 + 2 K)    fwd_b's target critics: own gather, state half of layer 1, poll the tile's action granules,
//                                   arrive at word 1 (low)
//        planes [2 + 2 K, 2 + 3 K)  bwd_a of critic k (waits for word 1: critics, then fwd_b)
//        plane  2 + 3 K (pol = 1)   pol_a of a policy iteration (waits for word 0)
//      The epoch of the tags is advanced by the last of its readers (policy role, projection, fwd_b: a count in ws).
__device__ __forceinline__ void pm_reader_done(unsigned long long* ws, unsigned long long tag0, unsigned readers) {
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(ws + kPmDone, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)readers - 1ull) {
        __hip_atomic_store(ws + kPmDone, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ws + kPmEpoch, ((tag0 - 1ull) >> 5) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the policy head of row i from the tile's partials staged in LDS as [group][row][output] (== ns_policy_head)
__device__ __forceinline__ float pm_policy_draw(const SplitArgs& p, int i, long long t) {     // RPOSAC: the rsample draw of row i
    if (p.eps_in) return p.eps_in[i];
    const rpo_u4 u = rpo_philox(p.noise_seed, p.noise_id_base + (uint32_t)i, (uint32_t)t + p.noise_salt, RPO_STREAM_POLICY,
                                (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
    return rpo_normal(u.x, u.y);
}
__device__ __forceinline__ float pm_policy_head_lds(const SplitArgs& p, const float* lpart, int r, float e, float* logp) {
    float v0 = p.twin ? p.actor.b1[0] : p.actor_target.b1[0], v1 = p.twin ? p.actor.b1b[0] : 0.0f;
#pragma unroll
    for (int g = 0; g < kNsGroups; ++g) { v0 += lpart[g * 32 + r * 2]; v1 += lpart[g * 32 + r * 2 + 1]; }
    if (!p.twin) return p.scale * tanhf(v0) + p.base;
    return rpo_head_dev::gauss_head_row(v0, v1, e, p.scale, p.base, p.box_lo, p.box_hi, 0, logp);
}

template <class L>
__device__ __forceinline__ bool pfront_role(const SplitArgs& p, const CartConsts& c, float* smem, int pol, unsigned long long* ws,
                                            int store_mode) {
    const int K = p.twin ? 2 : 1, T = (int)gridDim.y, z = (int)blockIdx.z, tid = threadIdx.x, B = p.B;
    if (z >= 2 + 3 * K + pol) return false;
    const unsigned consumers = (unsigned)(kNsGroups * (K + pol));          // of the tile words: bwd_a, pol_a
    const unsigned readers = (unsigned)(kNsGroups * T * (1 + K) + T);      // of the epoch: policy role, fwd_b, projection
    unsigned* sync = p.tile_sync;
    if (z >= 2 + 2 * K && z < 2 + 3 * K) {
        PM_STAMP(ws, z == 2 + 2 * K && blockIdx.x == 0 && blockIdx.y == 0, 52);
        bwd_a_role<L>(p, smem, (z - 2 - 2 * K) * T * kNsGroups + (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y, consumers,
                      (unsigned)(kNsGroups * K), (unsigned)(kNsGroups * K));
        PM_STAMP(ws, z == 2 + 2 * K && blockIdx.x == 0 && blockIdx.y == 0, 53);
        return true;
    }
    if (z == 1 + K) {                                            // ---- the projection of row tile blockIdx.y
        if (blockIdx.x != 0) return true;
        const int wg = blockIdx.y, row0 = wg * kPmRows;
        float* lds = smem;                                       // kPmSmem floats
        float* lpart = smem + kPmSmem;                           // [8][16][2] head partials
        float* logp_row = lpart + 256;                           // [16]
        float4* tile = reinterpret_cast<float4*>(logp_row + 16);  // the tile's gathered rows
        static_assert((kPmSmem + 256 + 16) % 4 == 0, "tile alignment");
        int* flags = reinterpret_cast<int*>(lds + 6 * kPmRows + 2 * 256);
        if (tid == 0) { flags[0] = 0; flags[2] = 0; }
        PM_STAMP(ws, wg == 0, 0);
        const unsigned long long tag0 = pm_tag0(ws);
        pm_publish_xcc(ws, tag0, wg);                            // (the others read it after the head: long since there)
        const long long t = p.ctrl[RPO_CTRL_T];
        ns_sample<L>(p, tile, row0, t, false);                   // (same draw, same rows as fwd_a's workgroups of this tile)
        float e_draw = 0.0f;                                     // the row's policy draw does not need the partials either
        if (tid < kPmRows && row0 + tid < B && p.twin) e_draw = pm_policy_draw(p, row0 + tid, t);
        __syncthreads();
        PM_STAMP(ws, wg == 0, 1);
        {
            const int g = tid >> 5, r = (tid >> 1) & 15, o = tid & 1;
            const bool mine = o < (p.twin ? 2 : 1) && row0 + r < B;
            unsigned long long* src = ws + kPmPart + ((size_t)g * B + row0 + r) * 2 + o;
            unsigned long long x = 0;
            for (int spins = 0;; ++spins) {
                bool ok = true;
                if (mine) { x = pm_load(src); ok = (x >> 32) == tag0; }
                if (__all(ok)) break;
                if (spins >= kPmSpinMax) { flags[2] = 1; x = 0; break; }
            }
            lpart[tid] = mine ? __uint_as_float((unsigned)x) : 0.0f;
        }
        __syncthreads();
        PM_STAMP(ws, wg == 0, 2);
        if (tid < kPmRows) {
            const int i = row0 + tid;
            float ax = 0.0f, logp = 0.0f;
            if (i < B) ax = pm_policy_head_lds(p, lpart, tid, e_draw, &logp);
            logp_row[tid] = logp;
            pm_stage_row(lds, tid, i < B, ax, reinterpret_cast<const float*>(tile) + tid * L::ROW + L::NS_OFF);
        }
        PM_STAMP(ws, wg == 0, 3);
        pm_project(p, ws, tag0, store_mode, wg, T, lds, logp_row, ws + kPmAct);
        pm_reader_done(ws, tag0, readers);
        return true;
    }
    if (tid >= kNsThreads) return true;                          // (whole waves: the barriers below count the two that stay)
    NsLds<128>& lds = *reinterpret_cast<NsLds<128>*>(smem);
    float4* tile = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + sizeof(NsLds<128>));
    const NsBlock nb = ns_block();
    const int row0 = nb.tile * kRows, g = nb.g;
    if (z == 0) {                                                // the policy on s'
        PM_STAMP(ws, nb.tile == 0 && g == 0, 40);
        const unsigned long long tag0 = pm_tag0(ws);
        float hp[2][4];
        // word 0 says "the tile's gathered rows are published" (pol_a waits for it): the arrival sits right behind the gather,
        // long before the granules -- nothing of this workgroup may follow the launch's last consumer, which clears the words
        fwd_a_role<L>(p, lds, tile, row0, g, 0, hp, sync + nb.tile * kNsSyncStride);
        PM_STAMP(ws, nb.tile == 0 && g == 0, 41);
        const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
        if (tid >= 64 && li == 0 && !((p.debug & 1) && nb.tile == 0 && g == 0)) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = row0 + lg * 4 + i;
                    if (row < B && o < (p.twin ? 2 : 1))
                        pm_store(ws + kPmPart + ((size_t)g * B + row) * 2 + o, (tag0 << 32) | (unsigned long long)__float_as_uint(hp[o][i]),
                                 false);
                }
        }
        PM_STAMP(ws, nb.tile == 0 && g == 0, 42);
        pm_reader_done(ws, tag0, readers);
    } else if (z < 1 + K) {
        PM_STAMP(ws, nb.tile == 0 && g == 0 && z == 1, 44);
        fwd_a_role<L>(p, lds, tile, row0, g, z);
        ns_tile_arrive(sync + (T + nb.tile) * kNsSyncStride, 1u << 16);
        PM_STAMP(ws, nb.tile == 0 && g == 0 && z == 1, 45);
    } else if (z < 2 + 2 * K) {                                  // fwd_b of target critic k on the projected actions
        const int k = z - 2 - K;
        PM_STAMP(ws, nb.tile == 0 && g == 0 && k == 0, 48);
        const unsigned long long tag0 = pm_tag0(ws);
        const Mlp& net = p.critic_target[k];
        NsWeights<128> w;
        ns_load_weights<128, 256>(net, g, w);
        ns_sample<L>(p, tile, row0, p.ctrl[RPO_CTRL_T], false);
        __syncthreads();
        ns_stage<L>(lds, reinterpret_cast<const float*>(tile), true, false);
        __syncthreads();
        float pre[kRows];
        ns_layer1_state<128>(net, w, lds, pre);
        PM_STAMP(ws, nb.tile == 0 && g == 0 && k == 0, 49);
        if (tid < 64) {                                          // one wave polls the tile's 16 x (a_x, a_y, log pi) granules
            const int r = tid / 3, q = tid - 3 * r;
            const bool mine = tid < 48 && row0 + r < B;
            unsigned long long x = 0;
            for (int spins = 0;; ++spins) {
                bool ok = true;
                if (mine) { x = pm_load(ws + kPmAct + (size_t)(row0 + r) * 3 + q); ok = (x >> 32) == tag0; }
                if (__all(ok)) break;
                if (spins >= kPmSpinMax) { x = 0; __hip_atomic_store(ws + kPmGaveUp, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            if (mine) {
                const float v = __uint_as_float((unsigned)x);
                if (q < 2) lds.in_a[r * 8 + q] = v;
                else if (p.twin && g == 0 && k == 0) p.logp[row0 + r] = v;
            }
        }
        PM_STAMP(ws, nb.tile == 0 && g == 0 && k == 0, 50);
        pm_reader_done(ws, tag0, readers);
        ns_hidden<128, 256>(net, w, lds, g, row0, B, p.part_qn[k], nullptr, nullptr, pre);
        ns_tile_arrive(sync + (T + nb.tile) * kNsSyncStride);
        PM_STAMP(ws, nb.tile == 0 && g == 0 && k == 0, 51);
    } else {
        pol_a_role<L>(p, lds, tile, row0, g, sync + nb.tile * kNsSyncStride, (unsigned)kNsGroups);
        ns_tile_passed(sync, T, nb.tile, consumers);
    }
    return true;
}

static_assert(kPmSmem + 256 + 16 + 4 * kRows * 4 <= kFrontSmem, "projection role fits");

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_pfront_kernel(SplitArgs p, CartConsts c, int pol, unsigned long long* ws,
                                                                      int store_mode) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    pfront_role<L>(p, c, smem, pol, ws, store_mode);
}

template <class L>
__global__ __launch_bounds__(kThreads) void split_critic_pfront_ride_kernel(SplitArgs p, CartConsts c, RideArgs<typename L::Env> r,
                                                                           unsigned long long* ws, int store_mode) {
    __shared__ __attribute__((aligned(16))) float smem[kFrontSmem];
    if (pfront_role<L>(p, c, smem, 0, ws, store_mode)) return;
    const NsBlock nb = ns_block();
    const int own = 2 + 3 * (p.twin ? 2 : 1);
    ride_forward_pair<typename L::Env>(r, reinterpret_cast<NsLds<128>*>(smem), ((int)blockIdx.z - own) * (int)gridDim.y + nb.tile, nb.g);
}

// ---- pol_b: head -> exploration noise + clip (RPODDPG) / rsample + clip + log pi (RPOSAC) -> Complete -> Lagrangian row
//      terms -> Q_k hidden slabs on (s, a_pi), pre-activations saved.  grid (row tiles, 8, critics); workgroup
//      (tile, 0, 0) publishes the per-row outputs and the tile's Lagrangian partial sums.
template <class ENV>
__device__ __forceinline__ void pol_b_role(const SplitArgs& p, const CartConsts& c, NsLds<128>& lds, float4* tile, int tile_i, int g,
                                           int k, unsigned* wait_word = nullptr, unsigned wait_need = 0u) {
    typedef typename ENV::L L;
    const int row0 = tile_i * kRows, tid = threadIdx.x, B = p.B;
    const Mlp& net = p.critic[k];
    NsWeights<128> w;
    ns_load_weights<128, 256>(net, g, w);
    ns_load_tile<L>(p, tile, row0);                              // (gathered by the critic update: an earlier launch)
    __syncthreads();
    ns_stage<L>(lds, reinterpret_cast<const float*>(tile), false, false);
    __syncthreads();
    float pre[kRows];
    if (wait_word) {                                             // fused policy front: the tile's policy slabs of THIS launch
        ns_layer1_state<128>(net, w, lds, pre);
        const int T = (B + kRows - 1) / kRows;
        ns_tile_wait(wait_word, wait_need, p.tile_sync + 3 * T * kNsSyncStride);
    }
    const bool writer = g == 0 && k == 0;
    float vals[7] = {0, 0, 0, 0, 0, 0, 0};
    if (tid < kRows) {
        const int i = row0 + tid;
        float2 act = make_float2(0.0f, 0.0f);
        if (i < B) {
            const long long t = p.ctrl[RPO_CTRL_T];
            const float inv_b = 1.0f / (float)B;
            float e;
            if (p.eps_in) {
                e = p.eps_in[i];
            } else {
                const rpo_u4 u = rpo_philox(p.noise_seed, p.noise_id_base + (uint32_t)i, (uint32_t)t + p.noise_salt,
                                            RPO_STREAM_POLICY, (uint32_t)p.ctrl[RPO_CTRL_UPDATES]);
                e = rpo_normal(u.x, u.y);
            }
            float ap, logp = 0.0f, rm = 0.0f, rl = 0.0f, ap_det = 0.0f;
            if (p.twin) {
                rm = ns_head(pol_part(p), B, i, 0, p.actor.b1[0]);
                rl = ns_head(pol_part(p), B, i, 1, p.actor.b1b[0]);
                ap = rpo_head_dev::gauss_head_row(rm, rl, e, p.scale, p.base, p.box_lo, p.box_hi, 0, &logp);
            } else {
                const float v = ns_head(pol_part(p), B, i, 0, p.actor.b1[0]);
                ap_det = p.scale * tanhf(v) + p.base;
                const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * (float)t);
                ap = explore_clip(ap_det, eps_t, e, p.box_lo, p.box_hi);
            }
            act = ENV::complete(c, lds.in_s + tid * 8, i, ap, t);
            if (writer) {
                if (p.twin) {
                    reinterpret_cast<float2*>(p.raw)[i] = make_float2(rm, rl);
                    p.logp[i] = logp;
                } else {
                    p.ap_det[i] = ap_det;
                }
                p.noise_out[i] = e;
                reinterpret_cast<float2*>(p.actions)[i] = act;
                float dist[6];
                float2 gg;
                vals[0] = ENV::lagr(c, act.x, act.y, p.nu, inv_b, dist, gg);
#pragma unroll
                for (int j = 0; j < 6; ++j) vals[1 + j] = dist[j];
                reinterpret_cast<float2*>(p.g_act)[i] = gg;
            }
        }
        lds.in_a[tid * 8] = act.x;
        lds.in_a[tid * 8 + 1] = act.y;
    }
    if (writer && tid < 64) {
#pragma unroll
        for (int q = 0; q < 7; ++q) {                                // (== rpo_wave_sum: only lanes 0..15 hold terms)
            const float sum = rpo_row16_sum_desc_lane0(vals[q]);
            if (tid == 0) p.lag_partial[tile_i * 8 + q] = sum;
        }
    }
    ns_hidden<128, 256>(net, w, lds, g, row0, B, p.part_q[k], p.x0[k], p.h1[k], wait_word ? pre : nullptr);
}

template <class ENV>
__global__ __launch_bounds__(kNsThreads) void split_policy_b_kernel(SplitArgs p, CartConsts c) {
    __shared__ NsLds<128> lds;
    __shared__ __attribute__((aligned(16))) float4 tile[kRows * ENV::L::CH];
    const NsBlock nb = ns_block();
    pol_b_role<ENV>(p, c, lds, tile, nb.tile, nb.g, blockIdx.z);
}

// dLoss/dQ_k of row i in the actor loss: -1/B (RPODDPG, rpo_ddpg.py:317) or the split of d(-min(Q1, Q2)) with ties
// shared (torch.min's backward, rpo_sac.py:331); *term = the row's share of column 7 of the loss partials
// (Q for RPODDPG, alpha log pi - min Q for RPOSAC).
__device__ __forceinline__ float ns_policy_dq(const SplitArgs& p, int k, int i, float* term) {
    const float inv_b = 1.0f / (float)p.B;
    const float q1 = ns_head(p.part_q[0], p.B, i, 0, p.critic[0].b1[0]);
    if (!p.twin) {
        *term = q1;
        return -inv_b;
    }
    const float q2 = ns_head(p.part_q[1], p.B, i, 0, p.critic[1].b1[0]);
    const float w1 = (q1 < q2 ? 1.0f : 0.0f) + (q1 == q2 ? 0.5f : 0.0f);
    *term = p.alpha * p.logp[i] - fminf(q1, q2);
    return k == 0 ? w1 * -inv_b : (1.0f - w1) * -inv_b;
}

// ---- pol_c: the critics' rows pass inside the actor loss: dx0 column groups (no parameter gradients of their own) and,
//      per group, its share of d(-Q)/d action = dx0 Wa.  blocks (critic, column group, row tile), 256 threads (as bwd_a).
__device__ __forceinline__ void pol_c_role(const SplitArgs& p, float* smem, int b, unsigned fused_consumers = 0u,
                                           unsigned fused_need = 0u) {
    constexpr int EIN = 128, H = 256, LDH = H + 4;
    float* dh_s = smem;                                           // [16][LDH]
    float* dq_s = smem + kRows * LDH;                             // [16]
    float* wpart = dq_s + 16;                                     // [4][16][16]
    const int T = (p.B + kRows - 1) / kRows, B = p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int k = b / (T * kNsGroups), rem = b - k * T * kNsGroups, g = rem / T, tile = rem - g * T;   // row tile fastest: see bwd_a
    const int row0 = tile * kRows;
    const Mlp& net = p.critic[k];
    const int jw = wave * (H / 4), e = g * 16 + li;
    float wv[H / 16], hv[kRows];
#pragma unroll
    for (int ks = 0; ks < H / 16; ++ks) wv[ks] = net.W0[(size_t)(jw + ks * 4 + lg) * EIN + e];
    const float w1a = net.W1[tid];
    const float2 wa2 = *reinterpret_cast<const float2*>(&net.Wa[(g * 16 + (tid & 15)) * 2]);
    if (fused_consumers) {                                       // fused policy front: the tile's Q slabs (pol_b) of THIS launch
        ns_tile_wait(p.tile_sync + (T + tile) * kNsSyncStride, fused_need, p.tile_sync + 3 * T * kNsSyncStride);
        ns_tile_passed(p.tile_sync, T, tile, fused_consumers);
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) hv[r] = row0 + r < B ? p.h1[k][(size_t)(row0 + r) * H + tid] : 0.0f;
    const size_t xo = (size_t)(row0 + (tid >> 4)) * EIN + g * 16 + (tid & 15);
    const float x0v = row0 + (tid >> 4) < B ? p.x0[k][xo] : 0.0f;
    if (tid < 64) {
        float dq = 0.0f, term = 0.0f;
        if (tid < kRows && row0 + tid < B) dq = ns_policy_dq(p, k, row0 + tid, &term);
        if (tid < kRows) dq_s[tid] = dq;
        const float sum = rpo_row16_sum_desc_lane0(term);        // (== rpo_wave_sum: only lanes 0..15 hold terms)
        if (tid == 0 && g == 0 && k == 0) p.lag_partial[tile * 8 + 7] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRows; ++r) dh_s[r * LDH + tid] = (hv[r] > 0.0f) ? dq_s[r] * w1a : 0.0f;
    __syncthreads();
    f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ks = 0; ks < H / 16; ++ks) acc = mfma4(dh_s[li * LDH + jw + ks * 4 + lg], wv[ks], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) wpart[(wave * 16 + lg * 4 + i) * 16 + li] = acc[i];
    __syncthreads();
    {
        const int r = tid >> 4, ee = tid & 15;
        float v = wpart[(0 * 16 + r) * 16 + ee];
        v = wpart[(1 * 16 + r) * 16 + ee] + v;
        v = wpart[(2 * 16 + r) * 16 + ee] + v;
        v = wpart[(3 * 16 + r) * 16 + ee] + v;
        const bool live = row0 + r < B;
        v = (live && x0v > 0.0f) ? v : 0.0f;
        if (live) p.dx0[k][xo] = v;
        // this group's share of d/d action: sum over its 16 columns (fixed butterfly), 2 action components
        float d0 = v * wa2.x, d1 = v * wa2.y;
        d0 = rpo_row16_sum_lane0(d0); d1 = rpo_row16_sum_lane0(d1);   // (xor-butterfly association at ee == 0)
        if (ee == 0 && live)
            reinterpret_cast<float2*>(p.da_part)[((size_t)k * kNsGroups + g) * B + row0 + r] = make_float2(d0, d1);
    }
}

__global__ __launch_bounds__(kThreads) void split_policy_c_kernel(SplitArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[kBwdASmem];
    pol_c_role(p, smem, blockIdx.x);
}


// d loss / d (actor head outputs) of row i: sum of the critics' d/d action shares + the Lagrangian's -> autograd through
// Complete -> head backward (rpo_ddpg.py:307-324, rpo_sac.py:321-339)
template <class ENV>
__device__ __forceinline__ float2 ns_policy_dout(const SplitArgs& p, const CartConsts& c, int i) {
    typedef typename ENV::L L;
    const int K = p.twin ? 2 : 1;
    float da0 = 0.0f, da1 = 0.0f;
    for (int k = 0; k < K; ++k) {
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int g = 0; g < kNsGroups; ++g) {
            const float2 d = reinterpret_cast<const float2*>(p.da_part)[((size_t)k * kNsGroups + g) * p.B + i];
            s0 += d.x; s1 += d.y;
        }
        da0 = k == 0 ? s0 : da0 + s0;
        da1 = k == 0 ? s1 : da1 + s1;
    }
    const float2 gg = reinterpret_cast<const float2*>(p.g_act)[i];
    da0 += gg.x; da1 += gg.y;
    const float dap = ENV::complete_bwd(c, p.batch_out + (size_t)i * L::ROW, da0, da1);
    if (p.twin) {
        const float2 r = reinterpret_cast<const float2*>(p.raw)[i];
        return rpo_head_dev::gauss_head_bwd_row(r.x, r.y, p.noise_out[i], dap, p.alpha / (float)p.B, p.scale, p.base, p.box_lo,
                                                p.box_hi);
    }
    const float t = (float)p.ctrl[RPO_CTRL_T];
    const float eps_t = fmaxf(p.eps_end, p.eps_start - p.eps_decay * t);
    return make_float2(rpo_head_dev::tanh_box_bwd_row(dap, p.ap_det[i], p.noise_out[i], 1, eps_t, p.box_lo, p.box_hi, p.scale,
                                                      p.base), 0.0f);
}

// ---- pol_d: d loss / d (actor head outputs) of the tile's rows -> actor dx0 of one (row tile, 16 first-layer columns);
//      the column-group-0 workgroup of a tile leaves the rows' head gradients for pol_e.  blocks (row tile, group).
template <class ENV>
__device__ __forceinline__ void pol_d_role(const SplitArgs& p, const CartConsts& c, float* smem, int b, unsigned fused_consumers = 0u,
                                           unsigned fused_need_a = 0u, unsigned fused_need_c = 0u) {
    constexpr int EIN = 128, H = 256, LDH = H + 4;
    const int B = p.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
    const Mlp& net = p.actor;
    {
        const int T = (B + kRows - 1) / kRows, g = b / T, tile = b - g * T, row0 = tile * kRows;   // row tile fastest: see bwd_a
        float* dh_s = smem;
        float* do_s = smem + kRows * LDH;                          // [16][2]
        float* wpart = do_s + 32;
        const int jw = wave * (H / 4), e = g * 16 + li;
        float wv[H / 16], hv[kRows];
#pragma unroll
        for (int ks = 0; ks < H / 16; ++ks) wv[ks] = net.W0[(size_t)(jw + ks * 4 + lg) * EIN + e];
        const float w1a = net.W1[tid], w1b = net.n_out > 1 ? net.W1b[tid] : 0.0f;
        unsigned* gave_up = p.tile_sync ? p.tile_sync + 3 * T * kNsSyncStride : nullptr;
        // fused policy front: the policy's saved activations are pol_a's (word 0, low count; an earlier launch when pol_a ran
        // early) -- requested before the wait for pol_c's workgroups of the tile (word 0, high count)
        if (fused_consumers && fused_need_a) ns_tile_wait(p.tile_sync + tile * kNsSyncStride, fused_need_a, gave_up);
#pragma unroll
        for (int r = 0; r < kRows; ++r) hv[r] = row0 + r < B ? p.h1_a[(size_t)(row0 + r) * H + tid] : 0.0f;
        const size_t xo = (size_t)(row0 + (tid >> 4)) * EIN + g * 16 + (tid & 15);
        const bool xlive = row0 + (tid >> 4) < B;
        const float x0v = xlive ? p.x0_a[xo] : 0.0f;
        if (fused_consumers) {
            ns_tile_wait(p.tile_sync + tile * kNsSyncStride, fused_need_c, gave_up, 16);
            ns_tile_passed(p.tile_sync, T, tile, fused_consumers);
        }
        float cdx = 0.0f;                                          // the critics' dx0 under a shared state embedding
        if (xlive && p.shared_embedding) cdx = p.twin ? p.dx0[0][xo] + p.dx0[1][xo] : p.dx0[0][xo];
        if (tid < kRows) {
            float2 d = make_float2(0.0f, 0.0f);
            if (row0 + tid < B) {
                d = ns_policy_dout<ENV>(p, c, row0 + tid);
                if (g == 0) reinterpret_cast<float2*>(p.dout)[row0 + tid] = d;      // [B, 2] for both head types
            }
            do_s[tid * 2] = d.x; do_s[tid * 2 + 1] = d.y;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRows; ++r)
            dh_s[r * LDH + tid] = (hv[r] > 0.0f) ? fmaf(do_s[r * 2 + 1], w1b, do_s[r * 2] * w1a) : 0.0f;
        __syncthreads();
        f32x4 acc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < H / 16; ++ks) acc = mfma4(dh_s[li * LDH + jw + ks * 4 + lg], wv[ks], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) wpart[(wave * 16 + lg * 4 + i) * 16 + li] = acc[i];
        __syncthreads();
        {
            const int r = tid >> 4, ee = tid & 15;
            float v = wpart[(0 * 16 + r) * 16 + ee];
            v = wpart[(1 * 16 + r) * 16 + ee] + v;
            v = wpart[(2 * 16 + r) * 16 + ee] + v;
            v = wpart[(3 * 16 + r) * 16 + ee] + v;
            if (xlive) {
                v = x0v > 0.0f ? v : 0.0f;
                // shared state embedding: ONE first-layer reduction yields its gradient from both losses' dx0
                if (p.shared_embedding) v += cdx;
                p.dx0_a[xo] = v;
            }
        }
    }
}

constexpr int kPolDSmem = kRows * (256 + 4) + 32 + 4 * 16 * 16;

template <class ENV>
__global__ __launch_bounds__(kThreads) void split_policy_d_kernel(SplitArgs p, CartConsts c) {
    __shared__ __attribute__((aligned(16))) float smem[kPolDSmem];
    pol_d_role<ENV>(p, c, smem, blockIdx.x);
}

// ---- fused front of the policy step: pol_a, pol_b, pol_c and pol_d in ONE launch (the hand-overs between them stay inside a
//      row tile, like those of the critic update's front).  grid (8, T, with_a + 2 K + 1) of 256-thread workgroups:
//        plane  0 (with_a = 1)              pol_a: the policy's slabs on the batch states; arrives at the tile's word 0 (low count)
//        planes [with_a, with_a + K)        pol_b of critic k: waits for word 0 low == 8 (with_a), arrives at word 1
//        planes [with_a + K, with_a + 2 K)  pol_c of critic k: waits for word 1 == 8 K, arrives at word 0 (high count)
//        plane  with_a + 2 K                pol_d: waits for pol_a (with_a), requests its activations, waits for word 0 high == 8 K
//      with_a = 0: pol_a ran inside the critic update's launch (front_pol / fwd_b_pol / mid_pol).
constexpr int kPolFrontSmem = kPolDSmem > kFrontSmem ? kPolDSmem : kFrontSmem;

template <class ENV>
__global__ __launch_bounds__(kThreads) void split_policy_front_kernel(SplitArgs p, CartConsts c, int with_a) {
    typedef typename ENV::L L;
    __shared__ __attribute__((aligned(16))) float smem[kPolFrontSmem];
    const int K = p.twin ? 2 : 1, T = (int)gridDim.y, z = (int)blockIdx.z;
    const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.y;
    const unsigned consumers = (unsigned)(kNsGroups * (K * (1 + with_a) + 1));   // pol_c's + pol_d's workgroups (+ pol_b's when they wait)
    if (z == with_a + 2 * K) {
        pol_d_role<ENV>(p, c, smem, lin, consumers, with_a ? (unsigned)kNsGroups : 0u, (unsigned)(kNsGroups * K));
        return;
    }
    if (z >= with_a + K) {
        pol_c_role(p, smem, (z - with_a - K) * T * kNsGroups + lin, consumers, (unsigned)(kNsGroups * K));
        ns_tile_arrive(p.tile_sync + (lin % T) * kNsSyncStride, 1u << 16);
        return;
    }
    if (threadIdx.x >= kNsThreads) return;
    NsLds<128>& lds = *reinterpret_cast<NsLds<128>*>(smem);
    float4* tile = reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + sizeof(NsLds<128>));
    const NsBlock nb = ns_block();
    unsigned* sync = p.tile_sync;
    if (z < with_a) {
        pol_a_role<L>(p, lds, tile, nb.tile * kRows, nb.g);
        ns_tile_arrive(sync + nb.tile * kNsSyncStride);
    } else {
        pol_b_role<ENV>(p, c, lds, tile, nb.tile, nb.g, z - with_a, with_a ? sync + nb.tile * kNsSyncStride : nullptr,
                        (unsigned)kNsGroups);
        ns_tile_arrive(sync + (T + nb.tile) * kNsSyncStride);
        if (with_a) ns_tile_passed(sync, T, nb.tile, consumers);
    }
}

// ---- pol_e: every parameter gradient of the actor.  blocks [0, 36): dW0 tiles and hidden-layer vectors from the rows' head
//      gradients (left by pol_d) and the saved activations; then the first-layer gradients from dx0_a; one more workgroup
//      folds the Lagrangian partials (value, d/d nu).
template <class ENV>
__global__ __launch_bounds__(kThreads) void split_policy_e_kernel(SplitArgs p, int fl_blocks) {
    typedef typename ENV::L L;
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * 16];
    if (blockIdx.x < kWeightBlocks) {
        gradmax_flush(p.gradmax, ns_weight_role(p.actor, p.actor_grad, p.h1_a, p.x0_a, p.dout, 2, p.actor.n_out > 1, p.B,
                                                blockIdx.x, smem));
        return;
    }
    const int fb = blockIdx.x - kWeightBlocks;
    if (fb == fl_blocks) {
        const int tid = threadIdx.x, T = (p.B + kRows - 1) / kRows;
        if (tid < 8) {
            float sacc = 0.0f;
            for (int g = 0; g < T; ++g) sacc += p.lag_partial[g * 8 + tid];
            const float inv_b = 1.0f / (float)p.B;
            if (tid == 0) p.lag_out[0] = inv_b * sacc;
            else if (tid == 7) { if (!p.twin) p.lag_out[1] = inv_b * sacc; }
            else if (tid - 1 < ENV::NI) p.nu_grad[tid - 1] += inv_b * sacc;
        }
        // bookkeeping for the rpo_adam_step_multi(prepared) launch behind the policy step (nothing in this launch reads it)
        if (tid >= 64 && tid < 67 && p.prep2_step[tid - 64]) adam_prepare(p.prep2_step[tid - 64], p.prep2_beta1[tid - 64], p.prep2_beta2[tid - 64]);
        if (tid == 128 && p.clock_out) p.clock_out[0] += 1;
        if (tid == 192 && p.updates_out) p.updates_out[RPO_CTRL_UPDATES] += 1;
        return;
    }
    BwdArgs a{};
    a.net = p.actor;
    a.g = p.actor_grad;
    a.n = p.B;
    a.s = p.batch_out; a.s_stride = L::ROW;
    a.dx0 = p.dx0_a;
    a.param_grads = 1;
    gradmax_flush(p.gradmax, mlp_bwd_first_layer<128>(a, fb));
}

}  // namespace

namespace {

MlpGrad grad_dev(const rpo_mlp_grad* g) {
    if (!g) return MlpGrad{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return MlpGrad{g->Ws, g->bs, g->Wa, g->ba, g->W0, g->b0, g->W1, g->b1, g->W1b, g->b1b};
}

// Resolve and validate the parts of rpo_split_update a stage needs.  need: bit 0 policy net, 1 critics, 2 target
// critics, 3 critic gradients, 4 actor gradients, 5 online actor (policy step).
int to_args(const rpo_split_update* u, unsigned need, SplitArgs& a, CartConsts& c) {
    if (!u) return RPO_ERR_NULL;
    if (u->batch <= 0 || u->batch > kBwdMaxB || (u->env != 0 && u->env != 1)) return RPO_ERR_ARG;
    const int S = u->env == 0 ? 6 : 5, K = u->twin ? 2 : 1;
    a = SplitArgs{};
    a.twin = u->twin ? 1 : 0; a.B = u->batch;
    auto ok_actor = [&](const rpo_mlp* h, Mlp& m) {
        if (!h) return (int)RPO_ERR_NULL;
        m = to_dev(h);
        return (split_ok(m) && m.S == S && m.A == 0 && m.n_out == (u->twin ? 2 : 1)) ? 0 : (int)RPO_ERR_ARG;
    };
    auto ok_critic = [&](const rpo_mlp* h, Mlp& m) {
        if (!h) return (int)RPO_ERR_NULL;
        m = to_dev(h);
        return (split_ok(m) && m.S == S && m.A == 2 && m.n_out == 1) ? 0 : (int)RPO_ERR_ARG;
    };
    if (need & 1u) {
        if (int e = u->twin ? ok_actor(u->actor, a.actor) : ok_actor(u->actor_target, a.actor_target)) return e;
    }
    if (need & 32u) {
        if (int e = ok_actor(u->actor, a.actor)) return e;
    }
    if (need & 2u) {
        if (int e = ok_critic(u->critic1, a.critic[0])) return e;
        if (K == 2) if (int e = ok_critic(u->critic2, a.critic[1])) return e;
    }
    if (need & 4u) {
        if (int e = ok_critic(u->critic_target1, a.critic_target[0])) return e;
        if (K == 2) if (int e = ok_critic(u->critic_target2, a.critic_target[1])) return e;
    }
    if (need & 8u) {
        if (!u->critic1_grad || (K == 2 && !u->critic2_grad)) return RPO_ERR_NULL;
        a.critic_grad[0] = grad_dev(u->critic1_grad);
        a.critic_grad[1] = grad_dev(u->critic2_grad);
        for (int k = 0; k < K; ++k) {
            const MlpGrad& g = a.critic_grad[k];
            if (!g.Ws || !g.bs || !g.Wa || !g.ba || !g.W0 || !g.b0 || !g.W1 || !g.b1) return RPO_ERR_NULL;
        }
    }
    if (need & 16u) {
        if (!u->actor_grad) return RPO_ERR_NULL;
        a.actor_grad = grad_dev(u->actor_grad);
        const MlpGrad& g = a.actor_grad;
        if (!g.Ws || !g.bs || !g.W0 || !g.b0 || !g.W1 || !g.b1 || (u->twin && (!g.W1b || !g.b1b))) return RPO_ERR_NULL;
    }
    c = CartConsts{};
    if (u->env == 0) {
        if (int e = rpo_cart_dev::load_consts(c, u->consts_host, u->partial)) return e;
    }
    a.rows = u->rows; a.cap_steps = u->cap_steps; a.n_envs = u->n_envs; a.batch_out = u->batch_out; a.idx_out = u->idx_out;
    a.idx_in = u->idx_in; a.sample_seed = (uint64_t)u->sample_seed; a.sample_salt = (uint32_t)u->sample_salt;
    a.eps_in = u->eps_in; a.noise_seed = (uint64_t)u->noise_seed; a.noise_id_base = (uint32_t)u->noise_id_base;
    a.noise_salt = (uint32_t)u->noise_salt; a.ctrl = u->ctrl;
    a.scale = u->scale; a.base = u->base; a.box_lo = u->box_lo; a.box_hi = u->box_hi;
    a.max_steps = u->max_steps; a.corr_lr = u->corr_lr; a.corr_eps = u->corr_eps; a.corr_momentum = u->corr_momentum;
    a.alpha = u->alpha; a.gamma = u->gamma; a.eps_start = u->eps_start; a.eps_end = u->eps_end; a.eps_decay = u->eps_decay;
    a.part_pi = u->part_pi; a.part_q[0] = u->part_q1; a.part_q[1] = u->part_q2; a.part_qn[0] = u->part_qn1; a.part_qn[1] = u->part_qn2;
    a.x0[0] = u->x0_1; a.h1[0] = u->h1_1; a.x0[1] = u->x0_2; a.h1[1] = u->h1_2; a.x0_a = u->x0_a; a.h1_a = u->h1_a;
    a.logp = u->logp; a.next_actions = u->next_actions; a.proj_iters = u->proj_iters;
    a.dq[0] = u->dq1; a.dq[1] = u->dq2; a.loss_partial = u->loss_partial; a.dx0[0] = u->dx0_1; a.dx0[1] = u->dx0_2;
    a.dx0_a = u->dx0_a; a.gradmax = u->gradmax; a.nu = u->nu; a.nu_grad = u->nu_grad;
    a.ap_det = u->ap_det; a.noise_out = u->noise_out; a.raw = u->raw; a.actions = u->actions; a.g_act = u->g_act;
    a.lag_partial = u->lag_partial; a.lag_out = u->lag_out; a.da_part = u->da_part; a.dout = u->dout;
    a.shared_embedding = u->shared_embedding;
    a.rollout_ctrl = u->rollout_ctrl; a.rollout_stats = u->rollout_stats; a.rollout_stats_cap = u->rollout_stats_cap;
    a.prep_step = u->prep_step; a.prep_beta1 = u->prep_beta1; a.prep_beta2 = u->prep_beta2; a.clock_out = u->clock_out;
    a.gradmax_reset = u->gradmax_reset; a.gradmax_reset2 = u->gradmax_reset2; a.updates_out = u->updates_out;
    a.part_pol = u->part_pol; a.tile_sync = u->tile_sync; a.debug = u->debug;
    for (int j = 0; j < 3; ++j) { a.prep2_step[j] = u->prep2_step[j]; a.prep2_beta1[j] = u->prep2_beta1[j]; a.prep2_beta2[j] = u->prep2_beta2[j]; }
    return 0;
}

}  // namespace

extern "C" {

int rpo_split_critic_fwd_a(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 1u | 2u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.rows || !a.batch_out || !a.ctrl || !a.part_pi || a.cap_steps <= 0 || a.n_envs <= 0) return RPO_ERR_NULL;
    if (a.rollout_ctrl && a.rollout_stats && a.rollout_stats_cap <= 0) return RPO_ERR_ARG;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k]) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, 1 + K);
    if (u->env == 0) hipLaunchKernelGGL(split_critic_fwd_a_kernel<CartRow>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(split_critic_fwd_a_kernel<PendRow>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_fwd_b(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, (u && u->env == 0 ? 1u : 0u) | 4u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out || !a.ctrl) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_qn[k]) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, K);
    if (u->env == 0) {
        if (!a.part_pi || (a.twin && !a.logp) || a.max_steps < 0) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_kernel<CartRow, 1>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    } else {
        if (!a.next_actions) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_kernel<PendRow, 0>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_xcc_probe(int gx, int gy, int gz, int threads, int* out, void* stream) {
    if (gx <= 0 || gy <= 0 || gz <= 0 || threads <= 0 || threads > 1024) return RPO_ERR_ARG;
    if (!out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(gx, gy, gz), dim3(threads), 0, (hipStream_t)stream, out);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_hw_probe(int gx, int gy, int gz, int threads, int lds_bytes, int spin, int* out, void* stream) {
    if (gx <= 0 || gy <= 0 || gz <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 4 || lds_bytes > 65536 || spin < 0) return RPO_ERR_ARG;
    if (!out) return RPO_ERR_NULL;
    hipLaunchKernelGGL(hw_probe_kernel, dim3(gx, gy, gz), dim3(threads), lds_bytes, (hipStream_t)stream, out, spin);
    RPO_LAUNCH_CHECK();
    return 0;
}

static int front_args(const rpo_split_update* u, unsigned need, SplitArgs& a, CartConsts& c) {
    if (!u) return RPO_ERR_NULL;
    if (u->env != 0) return RPO_ERR_ARG;
    if (int e = to_args(u, 1u | 2u | 4u | need, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.rows || !a.batch_out || !a.ctrl || !a.part_pi || !a.tile_sync || a.cap_steps <= 0 || a.n_envs <= 0) return RPO_ERR_NULL;
    if (a.rollout_ctrl && a.rollout_stats && a.rollout_stats_cap <= 0) return RPO_ERR_ARG;
    if ((a.twin && !a.logp) || a.max_steps < 0 || !a.loss_partial) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k] || !a.part_qn[k] || !a.dq[k] || !a.dx0[k]) return RPO_ERR_NULL;
    return 0;
}

static int front_launch(const rpo_split_update* u, int pol, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = front_args(u, pol ? 32u : 0u, a, c)) return e;
    if (pol) {
        if (u->shared_embedding) return RPO_ERR_ARG;              // the critic step would change the policy's first layer
        if (!a.part_pol || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    }
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, 1 + 3 * (a.twin ? 2 : 1) + pol);
    hipLaunchKernelGGL(split_critic_front_kernel<CartRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c, pol);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_front(const rpo_split_update* u, void* stream) { return front_launch(u, 0, stream); }
int rpo_split_critic_front_pol(const rpo_split_update* u, void* stream) { return front_launch(u, 1, stream); }

static int mid_args(const rpo_split_update* u, unsigned need, SplitArgs& a, CartConsts& c) {
    if (!u) return RPO_ERR_NULL;
    if (u->env != 1) return RPO_ERR_ARG;
    if (int e = to_args(u, 2u | 4u | need, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out || !a.ctrl || !a.next_actions || !a.tile_sync || !a.loss_partial || (a.twin && !a.logp)) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k] || !a.part_qn[k] || !a.dq[k] || !a.dx0[k]) return RPO_ERR_NULL;
    return 0;
}

static int mid_launch(const rpo_split_update* u, int pol, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = mid_args(u, pol ? 32u : 0u, a, c)) return e;
    if (pol) {
        if (u->shared_embedding) return RPO_ERR_ARG;
        if (!a.part_pol || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    }
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, 2 * (a.twin ? 2 : 1) + pol);
    hipLaunchKernelGGL(split_critic_mid_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c, pol);
    RPO_LAUNCH_CHECK();
    return 0;
}

// SpringPendulum: fwd_a + projection + fwd_b + bwd_a as one launch (pfront_role)
static int pfront_args(const rpo_split_update* u, unsigned need, SplitArgs& a, CartConsts& c) {
    if (!u) return RPO_ERR_NULL;
    if (u->env != 1) return RPO_ERR_ARG;
    if (int e = to_args(u, 1u | 2u | 4u | need, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.rows || !a.batch_out || !a.ctrl || !a.part_pi || !a.next_actions || !a.tile_sync || !a.loss_partial || !u->proj_ws ||
        (a.twin && !a.logp))
        return RPO_ERR_NULL;
    if (a.cap_steps <= 0 || a.n_envs <= 0 || a.B > 256 || a.max_steps < 0 || a.max_steps > kPmMaxSteps) return RPO_ERR_ARG;
    if (((uintptr_t)u->proj_ws & 127u) || u->proj_store_mode < 0 || u->proj_store_mode > 2) return RPO_ERR_ARG;
    if (a.rollout_ctrl && a.rollout_stats && a.rollout_stats_cap <= 0) return RPO_ERR_ARG;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k] || !a.part_qn[k] || !a.dq[k] || !a.dx0[k]) return RPO_ERR_NULL;
    return 0;
}

static int pfront_launch(const rpo_split_update* u, int pol, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = pfront_args(u, pol ? 32u : 0u, a, c)) return e;
    if (pol) {
        if (u->shared_embedding) return RPO_ERR_ARG;
        if (!a.part_pol || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    }
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, 2 + 3 * (a.twin ? 2 : 1) + pol);
    hipLaunchKernelGGL(split_critic_pfront_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c, pol, u->proj_ws,
                       u->proj_store_mode);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_pfront(const rpo_split_update* u, void* stream) { return pfront_launch(u, 0, stream); }
int rpo_split_critic_pfront_pol(const rpo_split_update* u, void* stream) { return pfront_launch(u, 1, stream); }

int rpo_split_critic_mid(const rpo_split_update* u, void* stream) { return mid_launch(u, 0, stream); }
int rpo_split_critic_mid_pol(const rpo_split_update* u, void* stream) { return mid_launch(u, 1, stream); }

int rpo_split_critic_fwd_b_pol(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, (u && u->env == 0 ? 1u : 0u) | 4u | 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (u->shared_embedding) return RPO_ERR_ARG;                  // the critic step would change the policy's first layer
    if (!a.batch_out || !a.ctrl || !a.part_pol || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_qn[k]) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, K + 1);
    if (u->env == 0) {
        if (!a.part_pi || (a.twin && !a.logp) || a.max_steps < 0) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_pol_kernel<CartRow, 1>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    } else {
        if (!a.next_actions) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_pol_kernel<PendRow, 0>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_pend_head_project(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 1u, a, c)) return e;
    if (u->env != 1 || a.max_steps < 0) return RPO_ERR_ARG;
    if (!a.batch_out || !a.ctrl || !a.part_pi || !a.next_actions || (a.twin && !a.logp)) return RPO_ERR_NULL;
    const size_t lds = ((size_t)a.B + 4) * sizeof(float);
    static_assert(kPmWords == RPO_PROJ_WS_WORDS && kPmGaveUp == RPO_PROJ_WS_GAVE_UP, "workspace layout");
    if (u->proj_ws && a.B <= 256 && a.max_steps <= kPmMaxSteps) {
        if (((uintptr_t)u->proj_ws & 127u) || u->proj_store_mode < 0 || u->proj_store_mode > 2) return RPO_ERR_ARG;
        hipLaunchKernelGGL(split_pend_head_project_multi_kernel, dim3(8 * ((a.B + kPmRows - 1) / kPmRows)), dim3(kThreads), 0,
                           (hipStream_t)stream, a, u->proj_ws, u->proj_store_mode, (const float*)nullptr, (const float*)nullptr, 0);
    } else if (a.B <= 256) {
        hipLaunchKernelGGL(split_pend_head_project_wide_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    } else {
        const int threads = (a.B + RPO_WAVE - 1) / RPO_WAVE * RPO_WAVE;
        hipLaunchKernelGGL(split_pend_head_project_kernel<1>, dim3(1), dim3(threads), lds, (hipStream_t)stream, a);
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_pendulum_project_batchref_ws(int n, const float* obs, int obs_stride, const float* ap, float* action, int* iters_out,
                                     int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                                     unsigned long long* ws, int store_mode, void* stream) {
    if (n <= 0 || n > 256 || max_steps < 0 || max_steps > kPmMaxSteps || obs_stride < RPO_PEND_OBS_DIM) return RPO_ERR_ARG;
    if (!obs || !ap || !action || !ws) return RPO_ERR_NULL;
    if (((uintptr_t)ws & 127u) || store_mode < 0 || store_mode > 2) return RPO_ERR_ARG;
    SplitArgs a{};
    a.B = n; a.next_actions = action; a.proj_iters = iters_out; a.max_steps = max_steps; a.corr_lr = corr_lr;
    a.corr_eps = corr_eps; a.corr_momentum = corr_momentum;
    hipLaunchKernelGGL(split_pend_head_project_multi_kernel, dim3(8 * ((n + kPmRows - 1) / kPmRows)), dim3(kThreads), 0, (hipStream_t)stream,
                       a, ws, store_mode, ap, obs, obs_stride);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_bwd_a(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 2u | 4u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    if (!a.batch_out || !a.loss_partial || (a.twin && !a.logp)) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.part_qn[k] || !a.x0[k] || !a.h1[k] || !a.dq[k] || !a.dx0[k]) return RPO_ERR_NULL;
    const int grid = K * T * kNsGroups;
    if (u->env == 0) hipLaunchKernelGGL(split_critic_bwd_a_kernel<CartRow>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(split_critic_bwd_a_kernel<PendRow>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_bwd_b(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 2u | 8u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.dx0[k] || !a.dq[k] || !a.x0[k] || !a.h1[k]) return RPO_ERR_NULL;
    const Mlp& m = a.critic[0];
    const int blocks = kWeightBlocks + mlp_fl_blocks(m.E * (m.S + 1 + m.A + 1)) + ((a.prep_step || a.clock_out || a.updates_out) ? 1 : 0);
    if (u->env == 0) hipLaunchKernelGGL(split_critic_bwd_b_kernel<CartRow>, dim3(blocks, K), dim3(kThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(split_critic_bwd_b_kernel<PendRow>, dim3(blocks, K), dim3(kThreads), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ---- riding rollout stages
namespace {

int ride_check(const rpo_split_update* u, const rpo_rollout_rider* r) {
    if (!u || !r) return RPO_ERR_NULL;
    if (r->n_envs <= 0 || r->max_episode_steps <= 0 || r->max_steps < 0) return RPO_ERR_ARG;
    if (r->noise_mode != RPO_NOISE_NONE && r->noise_mode != RPO_NOISE_PHILOX && r->noise_mode != RPO_NOISE_CLIP_ONLY) return RPO_ERR_ARG;
    if (!r->state || !r->action || !r->ep_len || !r->ep_ret || !r->ep_count || !r->ctrl || !r->part) return RPO_ERR_NULL;
    if ((r->rows && r->cap_steps <= 0) || (r->stats && r->stats_cap <= 0)) return RPO_ERR_ARG;
    if ((r->gauss != 0) != (u->twin != 0)) return RPO_ERR_ARG;          // u->actor is the rollout policy
    if (u->shared_embedding) return RPO_ERR_ARG;                         // the critic step would change the policy under the rollout
    return 0;
}

void ride_env(const rpo_rollout_rider* r, RideArgs<CartEnv>& o) {
    o.act = rpo_cart_dev::ActArgs{r->n_envs, nullptr, nullptr, r->action, nullptr, r->noise_mode, r->eps_start, r->eps_end,
                                  r->eps_decay, r->box_lo, r->box_hi, r->max_steps, r->corr_lr, r->corr_eps, r->corr_momentum,
                                  (uint64_t)r->seed, (uint32_t)r->env_id_base, r->ctrl, r->stats, r->stats_cap};
    o.step = rpo_cart_dev::StepArgs{r->n_envs, r->state, r->action, r->ep_len, r->ep_ret, r->ep_count, r->rows, r->cap_steps,
                                    r->stats, r->stats_cap, r->ctrl, r->max_episode_steps, r->auto_reset, r->viol_thresh,
                                    (uint64_t)r->seed, (uint32_t)r->env_id_base, 0};
}

void ride_env(const rpo_rollout_rider* r, RideArgs<PendEnv>& o) {
    o.act = rpo_pend_dev::ActArgs{r->n_envs, nullptr, 5, nullptr, nullptr, r->action, nullptr, r->noise_mode, r->eps_start,
                                  r->eps_end, r->eps_decay, r->box_lo, r->box_hi, r->max_steps, r->corr_lr, r->corr_eps,
                                  r->corr_momentum, (uint64_t)r->seed, (uint32_t)r->env_id_base, r->ctrl, r->stats, r->stats_cap};
    o.step = rpo_pend_dev::StepArgs{r->n_envs, r->state, r->obs, r->action, r->ep_len, r->ep_ret, r->ep_count, r->rows,
                                    r->cap_steps, r->stats, r->stats_cap, r->ctrl, r->max_episode_steps, r->auto_reset,
                                    r->viol_thresh, (uint64_t)r->seed, (uint32_t)r->env_id_base};
}

template <class ENV>
RideArgs<ENV> ride_args(const SplitArgs& a, const rpo_rollout_rider* r) {
    RideArgs<ENV> o{};
    o.actor = a.actor; o.scale = r->scale; o.base = r->base; o.gauss = r->gauss ? 1 : 0; o.n = r->n_envs; o.part = r->part;
    o.lane0 = r->lane_begin; o.lane1 = r->lane_end; o.defer_clock = r->defer_clock ? 1 : 0;
    ride_env(r, o);
    return o;
}

}  // namespace

extern "C" {

// forward stages: lanes [lane_begin, lane_end) of this launch
static int ride_range(const rpo_rollout_rider* r) {
    if (r->lane_begin < 0 || r->lane_begin % kRows || r->lane_end < r->lane_begin || r->lane_end > r->n_envs ||
        (r->lane_end % kRows && r->lane_end != r->n_envs))
        return RPO_ERR_ARG;
    return 0;
}

int rpo_split_critic_fwd_a_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    if (int e = ride_range(r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 1u | 2u | 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    if (!a.rows || !a.batch_out || !a.ctrl || !a.part_pi || a.cap_steps <= 0 || a.n_envs <= 0) return RPO_ERR_NULL;
    if (a.rollout_ctrl && a.rollout_stats && a.rollout_stats_cap <= 0) return RPO_ERR_ARG;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k]) return RPO_ERR_NULL;
    const int lane_wgs = ((r->lane_end - r->lane_begin + kRows - 1) / kRows + 1) / 2;      // two lane tiles per workgroup
    const dim3 grid(kNsGroups, T, 1 + K + (lane_wgs + T - 1) / T);
    if (u->env == 0)
        hipLaunchKernelGGL(split_critic_fwd_a_ride_kernel<CartRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, ride_args<CartEnv>(a, r));
    else
        hipLaunchKernelGGL(split_critic_fwd_a_ride_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, ride_args<PendEnv>(a, r));
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_fwd_b_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    if (int e = ride_range(r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, (u->env == 0 ? 1u : 0u) | 4u | 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    if (!a.batch_out || !a.ctrl) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_qn[k]) return RPO_ERR_NULL;
    const int lane_wgs = (r->lane_end - r->lane_begin + kRows - 1) / kRows;
    const dim3 grid(kNsGroups, T, K + (lane_wgs + T - 1) / T);
    if (u->env == 0) {
        if (!a.part_pi || (a.twin && !a.logp) || a.max_steps < 0) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_ride_kernel<CartRow, 1>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c,
                           ride_args<CartEnv>(a, r));
    } else {
        if (!a.next_actions) return RPO_ERR_NULL;
        hipLaunchKernelGGL((split_critic_fwd_b_ride_kernel<PendRow, 0>), grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c,
                           ride_args<PendEnv>(a, r));
    }
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_front_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    if (int e = ride_range(r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = front_args(u, 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    const int lane_wgs = ((r->lane_end - r->lane_begin + kRows - 1) / kRows + 1) / 2;      // two lane tiles per workgroup
    const dim3 grid(kNsGroups, T, 1 + 3 * K + (lane_wgs + T - 1) / T);
    hipLaunchKernelGGL(split_critic_front_ride_kernel<CartRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c,
                       ride_args<CartEnv>(a, r));
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_pfront_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    if (int e = ride_range(r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = pfront_args(u, 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    const int lane_wgs = ((r->lane_end - r->lane_begin + kRows - 1) / kRows + 1) / 2;      // two lane tiles per workgroup
    const dim3 grid(kNsGroups, T, 2 + 3 * K + (lane_wgs + T - 1) / T);
    hipLaunchKernelGGL(split_critic_pfront_ride_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c,
                       ride_args<PendEnv>(a, r), u->proj_ws, u->proj_store_mode);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_mid_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    if (int e = ride_range(r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = mid_args(u, 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    const int lane_wgs = ((r->lane_end - r->lane_begin + kRows - 1) / kRows + 1) / 2;      // two lane tiles per workgroup
    const dim3 grid(kNsGroups, T, 2 * K + (lane_wgs + T - 1) / T);
    hipLaunchKernelGGL(split_critic_mid_ride_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c,
                       ride_args<PendEnv>(a, r));
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_critic_bwd_b_ride(const rpo_split_update* u, const rpo_rollout_rider* r, void* stream) {
    if (int e = ride_check(u, r)) return e;
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 2u | 8u | 32u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.dx0[k] || !a.dq[k] || !a.x0[k] || !a.h1[k]) return RPO_ERR_NULL;
    const Mlp& m = a.critic[0];
    const int own = kWeightBlocks + mlp_fl_blocks(m.E * (m.S + 1 + m.A + 1)) + ((a.prep_step || a.clock_out || a.updates_out) ? 1 : 0);
    const dim3 grid(own + (r->n_envs + kThreads - 1) / kThreads, K);
    if (u->env == 0)
        hipLaunchKernelGGL(split_critic_bwd_b_ride_kernel<CartRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c,
                           ride_args<CartEnv>(a, r), own);
    else
        hipLaunchKernelGGL(split_critic_bwd_b_ride_kernel<PendRow>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c,
                           ride_args<PendEnv>(a, r), own);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

extern "C" {

int rpo_split_policy_a(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 32u, a, c)) return e;
    if (!a.batch_out || (!a.part_pi && !a.part_pol) || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows);
    if (u->env == 0) hipLaunchKernelGGL(split_policy_a_kernel<CartRow>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(split_policy_a_kernel<PendRow>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_policy_b(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 32u | 2u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out || !a.ctrl || (!a.part_pi && !a.part_pol) || !a.nu || !a.noise_out || !a.actions || !a.g_act || !a.lag_partial) return RPO_ERR_NULL;
    if (a.twin ? (!a.raw || !a.logp) : !a.ap_det) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k]) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, K);
    if (u->env == 0) hipLaunchKernelGGL(split_policy_b_kernel<CartPol>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    else hipLaunchKernelGGL(split_policy_b_kernel<PendPol>, grid, dim3(kNsThreads), 0, (hipStream_t)stream, a, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

static int policy_front_launch(const rpo_split_update* u, int with_a, void* stream) {
    SplitArgs a; CartConsts c;
    if (!u) return RPO_ERR_NULL;
    if (int e = to_args(u, 32u | 2u, a, c)) return e;
    const int K = a.twin ? 2 : 1;
    if (!a.batch_out || !a.ctrl || (!a.part_pi && !a.part_pol) || !a.nu || !a.noise_out || !a.actions || !a.g_act || !a.lag_partial ||
        !a.tile_sync || !a.da_part)
        return RPO_ERR_NULL;
    if (a.twin ? (!a.raw || !a.logp) : !a.ap_det) return RPO_ERR_NULL;
    if (with_a && (!a.x0_a || !a.h1_a)) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k] || !a.dx0[k]) return RPO_ERR_NULL;
    if (!a.dx0_a || !a.dout || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    const dim3 grid(kNsGroups, (a.B + kRows - 1) / kRows, with_a + 2 * K + 1);
    if (u->env == 0) hipLaunchKernelGGL(split_policy_front_kernel<CartPol>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c, with_a);
    else hipLaunchKernelGGL(split_policy_front_kernel<PendPol>, grid, dim3(kThreads), 0, (hipStream_t)stream, a, c, with_a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_policy_front(const rpo_split_update* u, void* stream) { return policy_front_launch(u, 1, stream); }
int rpo_split_policy_front_bc(const rpo_split_update* u, void* stream) { return policy_front_launch(u, 0, stream); }

int rpo_split_policy_c(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 2u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    if (!a.da_part || !a.lag_partial || (a.twin && !a.logp)) return RPO_ERR_NULL;
    for (int k = 0; k < K; ++k)
        if (!a.part_q[k] || !a.x0[k] || !a.h1[k] || !a.dx0[k]) return RPO_ERR_NULL;
    hipLaunchKernelGGL(split_policy_c_kernel, dim3(K * T * kNsGroups), dim3(kThreads), 0, (hipStream_t)stream, a);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_policy_d(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 32u | 16u, a, c)) return e;
    const int K = a.twin ? 2 : 1, T = (a.B + kRows - 1) / kRows;
    if (!a.batch_out || !a.ctrl || !a.da_part || !a.g_act || !a.noise_out || !a.x0_a || !a.h1_a || !a.dx0_a || !a.dout) return RPO_ERR_NULL;
    if (a.twin ? !a.raw : !a.ap_det) return RPO_ERR_NULL;
    if (a.shared_embedding)
        for (int k = 0; k < K; ++k)
            if (!a.dx0[k]) return RPO_ERR_NULL;
    const int grid = T * kNsGroups;
    if (u->env == 0) hipLaunchKernelGGL(split_policy_d_kernel<CartPol>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a, c);
    else hipLaunchKernelGGL(split_policy_d_kernel<PendPol>, dim3(grid), dim3(kThreads), 0, (hipStream_t)stream, a, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

int rpo_split_policy_e(const rpo_split_update* u, void* stream) {
    SplitArgs a; CartConsts c;
    if (int e = to_args(u, 32u | 16u, a, c)) return e;
    if (!a.batch_out || !a.dx0_a || !a.lag_partial || !a.lag_out || !a.nu_grad || !a.dout || !a.x0_a || !a.h1_a) return RPO_ERR_NULL;
    const int fl_blocks = mlp_fl_blocks(a.actor.E * (a.actor.S + 1));
    if (u->env == 0) hipLaunchKernelGGL(split_policy_e_kernel<CartPol>, dim3(kWeightBlocks + fl_blocks + 1), dim3(kThreads), 0, (hipStream_t)stream, a, fl_blocks);
    else hipLaunchKernelGGL(split_policy_e_kernel<PendPol>, dim3(kWeightBlocks + fl_blocks + 1), dim3(kThreads), 0, (hipStream_t)stream, a, fl_blocks);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
