// Launch arguments of the one-launch rollout, shared by its row-tile form (fused.hip) and its streaming form
// (rollout_stream.hip: a translation unit of its own, compiled without SLP vectorisation).
#pragma once
#include "mlp_tile.h"
#include "rollout_env.h"

namespace {

using rpo_mlp_dev::Mlp;

template <class ENV>
struct RolloutArgs {
    Mlp actor;
    float scale, base;            // tanh box of the actor output (BoxConstraint)
    int gauss;                    // 0: deterministic actor + exploration noise (DDPG); 1: squashed-Gaussian sample (SAC)
    int defer_clock;              // 1: the caller's next launch advances ctrl[T] and clears the next statistics row
    typename ENV::ActArgs act;    // exploration / projection parameters, action out
    typename ENV::StepArgs step;  // env state, bookkeeping, ring, statistics, ctrl
};

}  // namespace

// The streaming form (rollout_stream.hip).  `args` / `consts`: a RolloutArgs<CartEnv | PendEnv> and that env's Consts -- passed as
// untyped pointers because the env policies live in anonymous namespaces (one type per translation unit, same layout).
// Returns -1 when the form does not apply (shape, missing observation rows), 0 after a launch, > 0 = hipError_t.
__attribute__((visibility("hidden"))) int rpo_rollout_stream_launch(int env /* 0 CartSafe, 1 SpringPendulum */, const void* args,
                                                                    const void* consts, int n_envs, void* stream);
