// Device-side pieces of SpringPendulum-v0 shared by pendulum.hip and the fused pipelines (fused.hip).
#pragma once
#include "common.h"

namespace rpo_pend_dev {

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
    RPO_FP_STRICT
    Eq e;
    e.C_p = sin_t;   // fx is the basic action (pendulum.py:48,275-278)
    e.C_o = cos_t;
    e.C_o_inv = 1.0f / cos_t;
    e.b = -kMDt * ldot - (l * kM * thdot * thdot - kK * (l - kL0) - kM * kG * cos_t);   // :287
    return e;
}

__device__ __forceinline__ void reset_internal(float (&s)[4], uint64_t seed, uint32_t env_id, uint32_t episode) {
    RPO_FP_STRICT
    const rpo_u4 r = rpo_philox(seed, env_id, episode, RPO_STREAM_RESET);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = kResetLo[k] + rpo_u01(w[k]) * (kResetHi[k] - kResetLo[k]);
}

__device__ __forceinline__ void store_obs(float* __restrict__ o, float cs, float sn, float thdot, float l, float ldot) {
    o[0] = cs; o[1] = sn; o[2] = thdot; o[3] = l; o[4] = ldot;
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

// One lane's step: dynamics, violations, TimeLimit, statistics, auto-reset (pendulum.py:80-128 + the bookkeeping of
// rpo_ddpg.py:120-145).  `row` gets the 4 float4 chunks of the transition row, `ns` / (ncs, nsn) the state and
// observation the lane continues from.
__device__ __forceinline__ void pend_lane(const StepArgs& p, int i, float4 s, float2 a, const RpoEpisode& ep, float (&ns)[4],
                                          float& ncs, float& nsn, float4 (&row)[4], float (&st)[kStepStats]) {
    RPO_FP_STRICT
    const float th = s.x, thdot = s.y, l = s.z, ldot = s.w;
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
    // (pendulum.py:85-89 asserts on a NaN action; the speed clip below would launder a NaN: tested before it)
    rpo_flag_nonfinite(p.ctrl, a.x != a.x || a.y != a.y || !__builtin_isfinite((nth + nthdot) + (nl + nldot)));
    nthdot = fminf(fmaxf(nthdot, -kMaxSpeed), kMaxSpeed);
    const bool terminated = nl <= 0.5f || nl >= 1.5f || nth >= kThetaLim || nth <= -kThetaLim;   // :124
    const float reward = 1.0f / (100.0f * costs + 1.0f);
    const int len = ep.len + 1;
    const bool done = terminated || len >= p.max_episode_steps;
    const float ret = ep.ret + reward;
    sincosf(nth, &nsn, &ncs);

    row[0] = make_float4(cs, sn, thdot, l);
    row[1] = make_float4(ldot, a.x, a.y, ncs);
    row[2] = make_float4(nsn, nthdot, nl, nldot);
    row[3] = make_float4(reward, done ? 1.0f : 0.0f, h, gi);
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
    ns[0] = nth; ns[1] = nthdot; ns[2] = nl; ns[3] = nldot;
    if (done && p.auto_reset) {
        const unsigned episode = ep.count + 1u;
        p.ep_count[i] = episode;
        reset_internal(ns, p.seed, p.env_id_base + (uint32_t)i, episode);
        sincosf(ns[0], &nsn, &ncs);
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
    RPO_FP_STRICT
    const float Gx = 2.0f * ax, Gy = 2.0f * ay;                                  // set_ineq :296
    const float dgp = Gx - Gy * (e.C_o_inv * e.C_p);                             // :334-335
    const float bgp = kMaxSum - (e.b * e.C_o_inv) * Gy;                          // :336
    const float bm = ax * dgp - bgp;                                             // :337
    gx = (bm > 0.0f) ? dgp : 0.0f;                                               // :339
    gy = -(gx * e.C_p) * e.C_o_inv;                                              // :342
}

// Exploration + complete_partial (pendulum.py:256-262) + grad_steps with the row-wise ineq_partial_grad
// (pendulum.py:331-343 as evaluated for B = 1) for one lane whose observation is o[0..4].
__device__ __forceinline__ float2 pend_explore_project(const ActArgs& p, const float* o, int i, float ap_in, float eps_t,
                                                       long long t, int& iters) {
    RPO_FP_STRICT
    const Eq e = set_eq(o[0], o[1], o[2], o[3], o[4]);
    float ax = (p.noise_mode == RPO_NOISE_UNIFORM) ? 0.0f : ap_in;
    if (p.noise_mode == RPO_NOISE_EXPLICIT) {
        ax = rpo_clamp(ax + eps_t * p.noise[i], p.box_lo, p.box_hi);
    } else if (p.noise_mode == RPO_NOISE_PHILOX) {
        const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
        ax = rpo_clamp(ax + eps_t * rpo_normal(r.x, r.y), p.box_lo, p.box_hi);
    } else if (p.noise_mode == RPO_NOISE_UNIFORM) {
        const rpo_u4 r = rpo_philox(p.seed, p.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
        const float scale = (p.box_hi - p.box_lo) * 0.5f;
        ax = scale * (2.0f * rpo_u01(r.x) - 1.0f) + (p.box_lo + scale);
    } else if (p.noise_mode == RPO_NOISE_CLIP_ONLY) {
        ax = rpo_clamp(ax, p.box_lo, p.box_hi);
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
    iters = k;
    return make_float2(ax, ay);
}

// RPODDPG.grad_steps on a BATCH, literally (rpo_ddpg.py:266-286 with pendulum.py:331-343 for B > 1): one workgroup owns
// the batch, `lds` holds n + 4 floats.  Body of rpo_pendulum_project_batchref and of the column-split update's
// head + projection kernel (nsplit.hip): contraction off, so that both round identically.
//
// The coupling sum grad_i = sum_j 1[a_x,i dgp_j - bgp_i > 0] dgp_j is n^2 (mul, compare, select, add) per GRG iteration on
// ONE compute unit.  LPS lanes share a sample (sample i = thread / LPS, lane q = thread % LPS sums j = q, q + LPS, ...;
// two accumulators per lane; the LPS partials are added pairwise in a fixed butterfly), so that at batch 256 the CU runs
// 16 waves instead of 4 and the dependent add chains overlap: ~1.8 instead of ~4 us per iteration (40 -> 18 us for the
// 10 iterations a batch with one infeasible row takes).  `o` = the sample's observation (NULL beyond n), `ap_i` its basic
// action; every lane of a sample carries the same state.
template <int LPS>
__device__ __forceinline__ void project_batchref_body(int n, const float* o, float ap_i, float* __restrict__ action,
                                                      int* __restrict__ iters_out, int max_steps, float corr_lr,
                                                      float corr_eps, float corr_momentum, float* lds) {
    RPO_FP_STRICT
    float* dgp_s = lds;
    int* flag = reinterpret_cast<int*>(lds + n);                // two flags, used alternately: iteration k raises flag[k & 1]
    const int i = threadIdx.x / LPS, q = threadIdx.x % LPS;
    const bool live = i < n;
    Eq e = {0.0f, 1.0f, 1.0f, 0.0f};
    float ax = 0.0f, ay = 0.0f, ox = 0.0f, oy = 0.0f;
    if (live) {
        e = set_eq(o[0], o[1], o[2], o[3], o[4]);
        ax = ap_i;
        ay = (e.b - ax * e.C_p) * e.C_o_inv;                       // complete_partial :256-262
    }
    if (threadIdx.x < 2) flag[threadIdx.x] = 0;
    __syncthreads();
    int k = 0;
    for (; k < max_steps; ++k) {
        if (live && q == 0) {
            const float h = e.b - (ax * e.C_p + ay * e.C_o);
            const float g = ax * ax + ay * ay - kMaxSum;
            if (fabsf(h) > corr_eps || g > corr_eps) atomicOr(flag + (k & 1), 1);
            dgp_s[i] = 2.0f * ax - 2.0f * ay * (e.C_o_inv * e.C_p);                  // :334-335
        }
        if (threadIdx.x == 0) flag[(k + 1) & 1] = 0;             // nobody reads it before the next barrier pair
        __syncthreads();
        if (k > 0 && flag[k & 1] == 0) break;                      // batch-global stop test, rpo_ddpg.py:271-272
        {
            const float bgp = kMaxSum - (e.b * e.C_o_inv) * (2.0f * ay);             // :336
            // [B,1] @ [1,B] coupling, :337-339.  Lane q of a sample sums the contiguous quarter j in [q c, (q + 1) c), c = the
            // chunk length rounded to 4: one 16-byte LDS read feeds four terms (the strided form read every term on its own,
            // and this loop is instruction-issue-bound on ONE compute unit), four accumulators added pairwise.
            const int chunk = ((n + LPS - 1) / LPS + 3) & ~3, j0 = q * chunk;
            const int j1 = (j0 + chunk < n) ? j0 + chunk : n;
            float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
            int j = j0;
            for (; j + 4 <= j1; j += 4) {
                const float4 d = *reinterpret_cast<const float4*>(dgp_s + j);
                g0 += (ax * d.x - bgp > 0.0f) ? d.x : 0.0f;
                g1 += (ax * d.y - bgp > 0.0f) ? d.y : 0.0f;
                g2 += (ax * d.z - bgp > 0.0f) ? d.z : 0.0f;
                g3 += (ax * d.w - bgp > 0.0f) ? d.w : 0.0f;
            }
            for (; j < j1; ++j) {
                const float d0 = dgp_s[j];
                g0 += (ax * d0 - bgp > 0.0f) ? d0 : 0.0f;
            }
            float grad = (g0 + g1) + (g2 + g3);
            static_assert(LPS == 1 || LPS == 4, "lanes per sample");
            if (LPS == 4) grad = rpo_quad_sum(grad);               // same value in the 4 lanes (== the xor butterfly, DPP)
            const float gy = -(grad * e.C_p) * e.C_o_inv;                            // :342
            const float sx = corr_lr * grad + corr_momentum * ox;
            const float sy = corr_lr * gy + corr_momentum * oy;
            ax -= sx; ay -= sy;
            ox = sx; oy = sy;
        }
        __syncthreads();                                           // every lane has read dgp_s before the next round rewrites it
    }
    if (live && q == 0) reinterpret_cast<float2*>(action)[i] = make_float2(ax, ay);
    if (threadIdx.x == 0 && iters_out) *iters_out = k;
}

// The same for n <= 256 on a 1024-thread workgroup, laid out for the vector ALU (the n^2 predicate sum is issue-bound on the
// ONE compute unit that owns the batch; the LPS form above spent 3 us per GRG iteration, 2/3 of it in LDS traffic -- every
// thread read 64 values per iteration -- and in two 16-wave barriers):
//   * thread (g = tid / 16, c = tid % 16) owns the FOUR samples 4 g .. 4 g + 3 against the 16 values j = 64 q + 4 c + m
//     (q, m < 4): 4 ds_read_b128 per iteration instead of 16, each value used for four samples from a register;
//   * the 16 partial sums of a sample meet in a DPP butterfly inside the row (quad_perm, row_half_mirror, row_mirror: every
//     lane ends with the same bits), and all 16 lanes step their four samples redundantly -- the state stays in registers;
//   * one barrier per iteration: dgp is double-buffered, and the batch-global stop test (rpo_ddpg.py:271-272) is one
//     monotone word -- iteration k stores k + 1 when a row is infeasible, readers test >= k + 1 (a later iteration only
//     exists if this one went on, so a fast wave's k + 2 means the same as k + 1).
// The predicate is evaluated as fl(a_x,i * dgp_j) > bgp_i: with gradual underflow (f32 denormals are on in this build, as
// in torch on the CPU) fl(t - b) > 0 <=> t > b, so the selected set is exactly the reference's clamp(... - bgp, 0) > 0.
// The sum's order differs from the LPS form's (and from the reference's BLAS matmul, which has none to follow).
// Thread tid < n brings the basic action of sample tid (`ap_mine`) and reads its observation; lds: 8 * 256 + 4 floats.
constexpr int kWideLds = 8 * 256 + 4;

__device__ __forceinline__ float rpo_row16_allsum(float v) {      // sum over the 16 lanes of a DPP row, in EVERY lane
    v += rpo_dpp_mov<0xB1>(v);                                   // quad_perm:[1,0,3,2]
    v += rpo_dpp_mov<0x4E>(v);                                   // quad_perm:[2,3,0,1]
    v += rpo_dpp_mov<0x141>(v);                                  // row_half_mirror: the neighbouring quad's sum
    v += rpo_dpp_mov<0x140>(v);                                  // row_mirror: the other half's sum
    return v;
}

// One sample's state in the batched projection, and the pieces of a GRG iteration -- shared by the one-workgroup form below
// and the eight-workgroup form of nsplit.hip (split_pend_head_project_multi_kernel), which must round identically.
struct PbSample { float ax, ay, ox, oy, Cp, Co, Ci, bb, cc, bc; };

__device__ __forceinline__ void pb_init(PbSample& s, float ax, float ay, float Cp, float Co, float Ci, float bb) {
    RPO_FP_STRICT
    s.ax = ax; s.ay = ay; s.ox = 0.0f; s.oy = 0.0f; s.Cp = Cp; s.Co = Co; s.Ci = Ci; s.bb = bb;
    s.cc = Ci * Cp;                                                // :334-335 (C_o_inv C_p)
    s.bc = bb * Ci;                                                // :336 (b C_o_inv)
}
__device__ __forceinline__ float pb_complete(const Eq& e, float ax) {
    RPO_FP_STRICT
    return (e.b - ax * e.C_p) * e.C_o_inv;                         // complete_partial :256-262
}
// stop test of the row (rpo_ddpg.py:271-272) and its dgp (pendulum.py:334-335) at the current action
__device__ __forceinline__ bool pb_pre(const PbSample& s, float corr_eps, float& dgp) {
    RPO_FP_STRICT
    const float h = s.bb - (s.ax * s.Cp + s.ay * s.Co);
    const float g = s.ax * s.ax + s.ay * s.ay - kMaxSum;
    dgp = 2.0f * s.ax - 2.0f * s.ay * s.cc;
    return fabsf(h) > corr_eps || g > corr_eps;
}
__device__ __forceinline__ float pb_bgp(const PbSample& s) {
    RPO_FP_STRICT
    return kMaxSum - s.bc * (2.0f * s.ay);                         // :336
}
// the thread's 16 values of the row sum, j = 64 q + 4 c + m in the order (q, m): d = the dgp array of this iteration
__device__ __forceinline__ void pb_load16(const float* d, int c, float (&dv)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(d)[q * 16 + c];
        dv[4 * q] = v.x; dv[4 * q + 1] = v.y; dv[4 * q + 2] = v.z; dv[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ float pb_partial(float ax, float bgp, const float (&dv)[16]) {
    RPO_FP_STRICT
    float g = 0.0f;
#pragma unroll
    for (int u = 0; u < 16; ++u) g += (ax * dv[u] > bgp) ? dv[u] : 0.0f;   // [B,1] @ [1,B] coupling, :337-339
    return g;
}
__device__ __forceinline__ void pb_step(PbSample& s, float grad, float corr_lr, float corr_momentum) {
    RPO_FP_STRICT
    const float gy = -(grad * s.Cp) * s.Ci;                        // :342
    const float sx = corr_lr * grad + corr_momentum * s.ox;
    const float sy = corr_lr * gy + corr_momentum * s.oy;
    s.ax -= sx; s.ay -= sy;
    s.ox = sx; s.oy = sy;
}

__device__ __forceinline__ void project_batchref_wide(int n, const float* __restrict__ obs, int obs_stride, float ap_mine,
                                                      float* __restrict__ action, int* __restrict__ iters_out,
                                                      int max_steps, float corr_lr, float corr_eps, float corr_momentum,
                                                      float* lds) {
    RPO_FP_STRICT
    // lds: ax | ay | C_p | C_o | C_o_inv | b (256 each), dgp[2][256], stop word
    float* st = lds;
    float* dbuf = lds + 6 * 256;
    int* stop = reinterpret_cast<int*>(lds + 8 * 256);      // (plain LDS accesses: the barriers order them)
    const int tid = threadIdx.x, c = tid & 15, grp = tid >> 4;
    if (tid < 256) {
        Eq e = {0.0f, 1.0f, 1.0f, 0.0f};
        float ax = 0.0f, ay = 0.0f;
        if (tid < n) {
            const float* o = obs + (size_t)tid * obs_stride;
            e = set_eq(o[0], o[1], o[2], o[3], o[4]);
            ax = ap_mine;
            ay = pb_complete(e, ax);
        }
        st[tid] = ax; st[256 + tid] = ay; st[512 + tid] = e.C_p; st[768 + tid] = e.C_o; st[1024 + tid] = e.C_o_inv;
        st[1280 + tid] = e.b;
    }
    if (tid == 0) *stop = 0;
    __syncthreads();
    PbSample sm[4];
    {
        float v[6][4];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const float4 x = reinterpret_cast<const float4*>(st + 256 * a)[grp];
            v[a][0] = x.x; v[a][1] = x.y; v[a][2] = x.z; v[a][3] = x.w;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) pb_init(sm[s], v[0][s], v[1][s], v[2][s], v[3][s], v[4][s], v[5][s]);
    }
    const int mine = 4 * grp + c;                                  // lanes c < 4 publish dgp of sample 4 grp + c
    int k = 0;
    for (; k < max_steps; ++k) {
        bool viol = false;
        float dg[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) viol = pb_pre(sm[s], corr_eps, dg[s]) || viol;
        float* dk = dbuf + (k & 1) * 256;
        if (c < 4) dk[mine] = mine < n ? (c == 0 ? dg[0] : c == 1 ? dg[1] : c == 2 ? dg[2] : dg[3]) : 0.0f;
        if (viol) *stop = k + 1;                                   // (every writer of this iteration stores the same value)
        __syncthreads();
        if (k > 0 && *stop < k + 1) break;                         // batch-global stop test, rpo_ddpg.py:271-272
        float dv[16];
        pb_load16(dk, c, dv);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float grad = rpo_row16_allsum(pb_partial(sm[s].ax, pb_bgp(sm[s]), dv));
            pb_step(sm[s], grad, corr_lr, corr_momentum);
        }
    }
    if (c == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (4 * grp + s < n) reinterpret_cast<float2*>(action)[4 * grp + s] = make_float2(sm[s].ax, sm[s].ay);
    }
    if (tid == 0 && iters_out) *iters_out = k;
}

// One row of nu . relu(g(a)) with g = |a|^2 - 32 (pendulum.py:302-311; rpo_sac.py:326-335): returns nu0 * relu(g),
// `dist` = relu(g) (d/d nu) and (g0, g1) = scale * d/d action.  Shared by rpo_pendulum_lagrangian and the fused pipelines.
__device__ __forceinline__ float lagrangian_row(float ax, float ay, float nu0, float scale, float& dist, float& g0, float& g1) {
    RPO_FP_STRICT
    const float g = ax * ax + ay * ay - kMaxSum;
    dist = fmaxf(g, 0.0f);
    const float k = (g > 0.0f) ? 2.0f * scale * nu0 : 0.0f;
    g0 = k * ax; g1 = k * ay;
    return nu0 * dist;
}

// autograd through complete_partial (pendulum.py:256-262): d a_y / d a_x = -C_p * C_o_inv = -sin / cos
__device__ __forceinline__ float complete_bwd_row(const float* o, float g0, float g1) {
    RPO_FP_STRICT
    return g0 - g1 * (o[1] * (1.0f / o[0]));
}

}  // namespace rpo_pend_dev
