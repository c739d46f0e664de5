// The one-launch rollout in its STREAMING form (round 6; default from RPO_ROLLOUT_STREAM_FROM = 65 536 lanes): a translation unit of
// its own because it is compiled with -fno-slp-vectorize (rpo_amd/csrc/build.py).  Under plain -O3 the SLP vectoriser packs the
// head's lane-local fma chains into v_pk_fma_f32 behind register shuffles -- packed f32 VALU is an anti-lever beside f32 MFMAs
// (they share the vector lanes) -- and the epilogue of the two-head (RPOSAC) variant then spilled 26 registers per tile into a
// private segment (measured: 712 -> 641 us at 2^20 lanes for RPOSAC, 631 -> 624 us for RPODDPG, no scratch; the row-tile and
// column-split kernels measured 0.6 % SLOWER without SLP and keep it).
#include <stdlib.h>

#include "cartsafe_dev.h"
#include "heads_dev.h"
#include "mlp_stream.h"
#include "pendulum_dev.h"
#include "rollout_args.h"

namespace {

using namespace rpo_mlp_dev;

// ------------------------------------------------------------------------------- rollout, streaming form (round 6)
// From 65 536 lanes the rollout is throughput-, not latency-bound, and the row-tile kernel above (RT = 4: every 64-lane
// workgroup pulls W0 through its CU again and meets at four barriers per tile) stops at 0.54 of the f32 MFMA peak at 2^20 lanes.
// This is the forward of mlp_stream.h -- one persistent workgroup per CU, the actor's hidden matrix stationary in LDS, every
// WAVE owning whole 16-lane tiles, both layers transposed on the matrix cores, no barrier after the staging -- with the rest of
// the vector step as its epilogue: the tile's outputs stay in the wave's LDS slot, and after G tiles (G = 1 is what ships: 16
// lanes of the wave busy; G = 4 measured the same) every such lane runs ONE env lane through
// box / exploration noise or rsample -> Complete -> GRG projection -> env step -> ring row -> statistics: the functions of
// rollout_kernel (rollout_env.h), so state, actions and ring rows are the same bits (tests/test_full_size_gpu.py); the ring row
// leaves as a full 128-byte line.  Statistics: a wave reduces its lanes' values after every group and keeps ITS sums in LDS
// (nothing stays in registers across the MFMA loops); one commit per workgroup at the end.
struct RolloutStreamIn {                                         // what stream_inputs / stream_tile read (mlp.hip's FwdArgs)
    Mlp net;
    int n;
    const float* s; int s_stride;
    const float* a; int a_stride;
    float* out; float* x0_save; float* h1_save;
    int out_mode; float scale, base;
};
struct RolloutKeep {                                             // EMIT of stream_tile: the outputs of row `row` -> the wave's slot
    float* slot;
    int row0;
    __device__ __forceinline__ void operator()(int row, float o0, float o1, bool two) const {
        slot[(row - row0) * 2] = o0;
        slot[(row - row0) * 2 + 1] = two ? o1 : 0.0f;
    }
};
constexpr int kRolloutStreamWaves = 16;

#ifndef RPO_RSTREAM_SKIP
#define RPO_RSTREAM_SKIP 0         // timing-only builds: 1 no per-lane phase (MLP only), 2 no MLP (per-lane phase only)
#endif

// The per-lane phase reads its ~60 launch parameters (exploration / projection / step arguments, the env's constant table)
// from a copy in LDS through a pointer the optimiser cannot see through: as kernel arguments they are loaded once at the top
// and stay live across the MFMA loops (first build: 166 spilled SGPRs, 125 spilled VGPRs, a 284-byte private segment).
template <class ENV>
struct RolloutStreamParams {
    RolloutArgs<ENV> p;
    typename ENV::Consts c;
};

template <class ENV, int G>
__device__ __attribute__((noinline)) void rollout_stream_lanes(const RolloutStreamParams<ENV>* par, const float* slot, float* stat,
                                                               int grow0, int lane) {
    constexpr int kStats = 10, kLanes = kRows * G;
    const RolloutArgs<ENV>& p = par->p;
    const int n = p.step.n;
    const long long t = p.step.ctrl[RPO_CTRL_T];
    float st[kStats];
#pragma unroll
    for (int k = 0; k < kStats; ++k) st[k] = 0.0f;
    float iters_f = 0.0f;
    const int i = grow0 + lane;
    if (lane < kLanes && i < n) {
        const float eps_t = fmaxf(p.act.eps_end, p.act.eps_start - p.act.eps_decay * (float)t);
        const bool philox_noise = !p.gauss && p.act.noise_mode == RPO_NOISE_PHILOX;
        const long long ring_base = p.step.rows ? (t % p.step.cap_steps) * (long long)n : 0;
        float obs[8];
        ENV::lane_obs(p.step, i, obs);
        const RpoEpisode ep = ENV::episode(p.step, i);
        float ap = slot[lane * 2];
        float draw = 0.0f;
        if (p.gauss || philox_noise) {
            const rpo_u4 u = p.gauss ? rpo_philox(p.act.seed, p.act.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_POLICY,
                                                  (uint32_t)p.step.ctrl[RPO_CTRL_UPDATES])
                                     : rpo_philox(p.act.seed, p.act.env_id_base + (uint32_t)i, (uint32_t)t, RPO_STREAM_ACT);
            draw = rpo_normal(u.x, u.y);
        }
        if (p.gauss) ap = rpo_head_dev::gauss_head_row(ap, slot[lane * 2 + 1], draw, p.scale, p.base, p.act.box_lo, p.act.box_hi, 0, nullptr);
        typename ENV::ActArgs act = p.act;
        if (philox_noise) {
            ap = rpo_explore_clip(ap, eps_t, draw, p.act.box_lo, p.act.box_hi);
            act.noise_mode = RPO_NOISE_NONE;
        }
        int k;
        const float2 a = ENV::project(act, par->c, obs, i, ap, eps_t, t, k);
        iters_f = (float)k;
        reinterpret_cast<float2*>(p.act.action)[i] = a;
        ENV::lane(p.step, par->c, i, obs, a, ep, ring_base, st, true);
    }
    if (p.step.stats) {                                          // the wave's sums of this group -> its LDS row
        float red[kStats + 1];
#pragma unroll
        for (int k = 0; k < kStats; ++k) red[k] = st[k];
        red[kStats] = iters_f;
        rpo_wave_reduce_many<kStats + 1, (G > 1 ? 32 : 8)>(red, 3u << 8);
        float mine = 0.0f;
#pragma unroll
        for (int k = 0; k < kStats + 1; ++k)
            if (lane == k) mine = red[k];
        if (lane < kStats + 1) {
            const float cur = stat[lane];
            stat[lane] = ((3u << 8) >> lane) & 1u ? fmaxf(cur, mine) : cur + mine;
        }
    }
}

template <class ENV, int G>
__global__ __launch_bounds__(kRolloutStreamWaves * 64, kRolloutStreamWaves / 4) void rollout_stream_kernel(RolloutArgs<ENV> p_in, typename ENV::Consts c_in) {
    constexpr int H = 256, NW = kRolloutStreamWaves, kStats = 10, kLanes = kRows * G;
    __shared__ StreamLds<H> lds;
    __shared__ float out_s[NW][kLanes * 2];
    __shared__ float stat_s[NW][16];
    __shared__ RolloutStreamParams<ENV> par_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = p_in.step.n;
    const RolloutStreamIn in{p_in.actor, n, ENV::obs_rows(p_in.step), ENV::OBS, nullptr, 0, nullptr, nullptr, nullptr, p_in.gauss ? 0 : 1, p_in.scale, p_in.base};
    const bool gauss = p_in.gauss != 0;
    stream_stage<H, NW>(p_in.actor, lds, tid);
    if (tid == 0) { par_s.p = p_in; par_s.c = c_in; }
    if (lane < 16) stat_s[wave][lane] = 0.0f;
    __syncthreads();
    const float b1a = p_in.actor.b1[0], b1b = gauss ? p_in.actor.b1b[0] : 0.0f;
    const int tiles = (n + kRows - 1) / kRows, groups = (tiles + G - 1) / G;
    const int g0 = blockIdx.x * NW + wave, dg = gridDim.x * NW;
    float in3[3] = {0.0f, 0.0f, 0.0f};
    if (g0 < groups) stream_inputs(in, (long long)g0 * kLanes, lane, in3);
    for (int g = g0; g < groups; g += dg) {
        const int grow0 = g * kLanes;
        const RolloutKeep keep{&out_s[wave][0], grow0};
#pragma unroll 1
        for (int q = 0; q < G; ++q) {
            const int row0 = grow0 + q * kRows;
            // the next tile's inputs land under this tile's MFMAs (rows beyond n: a clamped row, never used)
            const long long nrow0 = q + 1 < G ? (long long)row0 + kRows : (g + dg < groups ? (long long)(g + dg) * kLanes : (long long)row0);
            float nxt[3];
            stream_inputs(in, nrow0, lane, nxt);
            if (!(RPO_RSTREAM_SKIP & 2) && row0 < n) {
                if (row0 + kRows <= n) {
                    if (gauss) stream_tile<H, true, 2, 0, 1>(in, lds, row0, lane, b1a, b1b, in3, keep);
                    else stream_tile<H, true, 2, 0, 0>(in, lds, row0, lane, b1a, b1b, in3, keep);
                } else {
                    stream_tile_any<H, 2>(in, lds, row0, lane, b1a, b1b, in3, keep);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) in3[ks] = nxt[ks];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the slot is written and read by lanes of ONE wave)
        __builtin_amdgcn_wave_barrier();
        if (RPO_RSTREAM_SKIP & 1) continue;
        // ---- one env lane per thread: the per-lane chain of rollout_kernel, OUT OF LINE (inlined, its registers and the
        // values the optimiser hoists out of it competed with the 126 registers of the MFMA loops: 64-125 spilled VGPRs)
        rollout_stream_lanes<ENV, G>(&par_s, &out_s[wave][0], &stat_s[wave][0], grow0, lane);
    }
    __syncthreads();
    const long long t = p_in.step.ctrl[RPO_CTRL_T];
    if (p_in.step.stats && tid < kStats + 1) {                   // the waves' sums in wave order; lane k commits statistic k: ONE
        const int k = tid;                                       // request per workgroup (rpo_stats_commit's form)
        const bool mx = ((3u << 8) >> k) & 1u;
        float r = stat_s[0][k];
        for (int w = 1; w < NW; ++w) r = mx ? fmaxf(r, stat_s[w][k]) : r + stat_s[w][k];
        const int slot = k == 0 ? RPO_STAT_REWARD_SUM : k == 1 ? RPO_STAT_EPISODES : k == 2 ? RPO_STAT_RETURN_SUM : k == 3 ? RPO_STAT_LENGTH_SUM :
                         k == 4 ? RPO_STAT_MAX_INEQ_SUM : k == 5 ? RPO_STAT_MAX_EQ_SUM : k == 6 ? RPO_STAT_VIOL_COUNT : k == 7 ? RPO_STAT_TERMINATED :
                         k == 8 ? RPO_STAT_MAX_INEQ_MAX : k == 9 ? RPO_STAT_MAX_EQ_MAX : RPO_STAT_PROJ_ITERS;
        float* row = rpo_stats_row(p_in.step.stats, p_in.step.stats_cap, t);
        if (!mx && r != 0.0f) atomicAdd(row + slot, r);
        if (mx && r > 0.0f) rpo_atomic_max_nonneg(row + slot, r);
    }
    if (!p_in.defer_clock) rpo_step_epilogue(p_in.step.ctrl, t, p_in.step.stats, p_in.step.stats_cap);
}

static int rollout_stream_cus() { return rpo_cu_count(); }

template <class ENV>
bool rollout_stream_ok(const RolloutArgs<ENV>& args) {
    const Mlp& a = args.actor;
    return stream_shape_ok(a) && a.A == 0 && ENV::obs_rows(args.step) != nullptr && args.step.ctrl != nullptr &&
           ((reinterpret_cast<uintptr_t>(a.W0)) & 15u) == 0;
}


template <class ENV>
int launch_stream(const void* args_v, const void* consts_v, int n_envs, void* stream) {
    const RolloutArgs<ENV>& args = *static_cast<const RolloutArgs<ENV>*>(args_v);
    const typename ENV::Consts& c = *static_cast<const typename ENV::Consts*>(consts_v);
    if (!rollout_stream_ok<ENV>(args)) return -1;
    // G = 4 (64-lane groups: every lane of the wave busy in the per-lane phase) once every wave of the chip gets a group; G = 1
    // below.  Timing-only builds at 2^20 lanes (RPO_RSTREAM_SKIP, one box): MLP alone 507 us, per-lane phase alone 114 us (G = 1) /
    // 60 us (G = 4), together 627 / 557 us -- the per-lane phase's vector instructions ADD to the MFMA time (f32 MFMA shares the
    // vector lanes), so issuing them for 64 instead of 16 lanes is worth 11 %.  At 65 536 lanes G = 4 leaves three waves in four
    // without a tile (147 us instead of 52).  rpo_tuning(RPO_TUNE_ROLLOUT_WIDE): 3 forces the streaming form with this size
    // rule, 4 with G = 4 (tests).
    const int sel = rpo_tune(RPO_TUNE_ROLLOUT_WIDE);
    const bool g4 = sel == 4 || n_envs >= 4 * kRows * kRolloutStreamWaves * rollout_stream_cus();
    const int groups = (n_envs + (g4 ? 4 : 1) * kRows - 1) / ((g4 ? 4 : 1) * kRows);
    int gx = (groups + kRolloutStreamWaves - 1) / kRolloutStreamWaves;
    if (gx > rollout_stream_cus()) gx = rollout_stream_cus();
    if (g4) hipLaunchKernelGGL((rollout_stream_kernel<ENV, 4>), dim3(gx), dim3(kRolloutStreamWaves * 64), 0, (hipStream_t)stream, args, c);
    else hipLaunchKernelGGL((rollout_stream_kernel<ENV, 1>), dim3(gx), dim3(kRolloutStreamWaves * 64), 0, (hipStream_t)stream, args, c);
    RPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace

int rpo_rollout_stream_launch(int env, const void* args, const void* consts, int n_envs, void* stream) {
    return env == 0 ? launch_stream<CartEnv>(args, consts, n_envs, stream) : launch_stream<PendEnv>(args, consts, n_envs, stream);
}
