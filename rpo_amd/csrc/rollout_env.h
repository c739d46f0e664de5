// Env policies of the rollout kernels: how a lane's observation is staged for the actor, how its raw action is explored +
// projected, and how the lane is stepped and scattered into the replay ring.  Shared by the one-launch rollout
// (fused.hip) and the rollout stages that ride along with the update launches (nsplit.hip).
#pragma once
#include "cartsafe_dev.h"
#include "pendulum_dev.h"

namespace {

struct CartEnv {
    typedef rpo_cart_dev::StepArgs StepArgs;
    typedef rpo_cart_dev::ActArgs ActArgs;
    typedef rpo_cart_dev::CartConsts Consts;
    static constexpr int OBS = 6;
    __device__ static __forceinline__ void stage_obs(const StepArgs& p, int row0, int rows, float* in_s, int stride, int tid_in = -1) {
        const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in;
        if (tid < rows * 6) {                                  // CartSafe observes its state directly
            const int r = tid / 6, i = tid - r * 6;
            in_s[r * stride + i] = (row0 + r < p.n) ? p.state[(size_t)(row0 + r) * 6 + i] : 0.0f;
        }
    }
    // the same observation for ONE lane, in registers (the riding tail of nsplit.hip: one thread per lane)
    __device__ static __forceinline__ void lane_obs(const StepArgs& p, int i, float (&o)[8]) {
#pragma unroll
        for (int q = 0; q < 6; ++q) o[q] = p.state[(size_t)i * 6 + q];
        o[6] = o[7] = 0.0f;
    }
    __device__ static __forceinline__ float2 project(const ActArgs& a, const Consts& c, const float* obs, int i, float ap,
                                                     float eps_t, long long t, int& k) {
        return rpo_cart_dev::cart_explore_project(a, c, i, ap, eps_t, t, k);
    }
    __device__ static __forceinline__ RpoEpisode episode(const StepArgs& p, int i) {
        return rpo_load_episode(p.ep_len, p.ep_ret, p.ep_count, i);
    }
    // `full_line`: the 32 bytes of padding behind the transition are written too (zeros, like the step kernel's tile path), so
    // that a whole 128-byte line leaves the CU -- the streaming rollout at >= 65 536 lanes, where the ring is HBM traffic
    __device__ static __forceinline__ void lane(const StepArgs& p, const Consts& c, int i, const float* obs, float2 a,
                                                const RpoEpisode& ep, long long ring_base,
                                                float (&st)[rpo_cart_dev::kStepStats], bool full_line = false) {
        float s[6], ns[6];
        float4 row[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) s[q] = obs[q];
        rpo_cart_dev::cart_lane(p, c, i, s, a, ep, ns, row, st);
        if (p.rows) {
            float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_CART_RING);
#pragma unroll
            for (int q = 0; q < 6; ++q) gr[q] = row[q];
            if (full_line) {
#pragma unroll
                for (int q = 6; q < RPO_CART_RING / 4; ++q) gr[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        rpo_cart_dev::store_state(p.state + (size_t)i * 6, ns);
    }
    // where the actor's inputs of the lanes live, as rows of OBS floats (the streaming rollout reads them in the MFMA layout)
    __device__ __host__ static __forceinline__ const float* obs_rows(const StepArgs& p) { return p.state; }
};

struct PendEnv {
    typedef rpo_pend_dev::StepArgs StepArgs;
    typedef rpo_pend_dev::ActArgs ActArgs;
    struct Consts { int unused; };
    static constexpr int OBS = 5;
    __device__ static __forceinline__ void stage_obs(const StepArgs& p, int row0, int rows, float* in_s, int stride, int tid_in = -1) {
        const int tid = tid_in < 0 ? (int)threadIdx.x : tid_in;
        if (tid < rows) {                                      // obs = (cos, sin, theta_dot, l, l_dot) of the internal state
            float4 s = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
            if (row0 + tid < p.n) s = reinterpret_cast<const float4*>(p.internal)[row0 + tid];
            float sn, cs;
            sincosf(s.x, &sn, &cs);
            float* o = in_s + tid * stride;
            o[0] = cs; o[1] = sn; o[2] = s.y; o[3] = s.z; o[4] = s.w;
        }
    }
    __device__ static __forceinline__ void lane_obs(const StepArgs& p, int i, float (&o)[8]) {
        const float4 s = reinterpret_cast<const float4*>(p.internal)[i];
        float sn, cs;
        sincosf(s.x, &sn, &cs);
        o[0] = cs; o[1] = sn; o[2] = s.y; o[3] = s.z; o[4] = s.w;
        o[5] = o[6] = o[7] = 0.0f;
    }
    __device__ static __forceinline__ float2 project(const ActArgs& a, const Consts&, const float* obs, int i, float ap,
                                                     float eps_t, long long t, int& k) {
        return rpo_pend_dev::pend_explore_project(a, obs, i, ap, eps_t, t, k);
    }
    __device__ static __forceinline__ RpoEpisode episode(const StepArgs& p, int i) {
        return rpo_load_episode(p.ep_len, p.ep_ret, p.ep_count, i);
    }
    __device__ static __forceinline__ void lane(const StepArgs& p, const Consts&, int i, const float* obs, float2 a,
                                                const RpoEpisode& ep, long long ring_base,
                                                float (&st)[rpo_pend_dev::kStepStats], bool full_line = false) {
        const float4 s = reinterpret_cast<const float4*>(p.internal)[i];
        float ns[4], ncs, nsn;
        float4 row[4];
        rpo_pend_dev::pend_lane(p, i, s, a, ep, ns, ncs, nsn, row, st);
        if (p.rows) {
            float4* gr = reinterpret_cast<float4*>(p.rows + (size_t)(ring_base + i) * RPO_PEND_RING);
#pragma unroll
            for (int q = 0; q < 4; ++q) gr[q] = row[q];
        }
        reinterpret_cast<float4*>(p.internal)[i] = make_float4(ns[0], ns[1], ns[2], ns[3]);
        if (p.obs) rpo_pend_dev::store_obs(p.obs + (size_t)i * 5, ncs, nsn, ns[1], ns[2], ns[3]);
    }
    __device__ __host__ static __forceinline__ const float* obs_rows(const StepArgs& p) { return p.obs; }   // (NULL: no streaming form)
};

}  // namespace
