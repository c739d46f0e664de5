#!/usr/bin/env python
"""Launch the hot kernels a few times at fixed sizes (for rocprofv3 --pmc / --kernel-trace collection).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o fetch -- python3 tools/kernel_probe.py step
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402


def probe_step(reps=5, sizes=(4096, 1 << 20)):
    from bench import make_trainer
    from rpo_amd import ops
    from rpo_amd.env.vec import VecEnv
    tr = make_trainer(4096, torch.device("cuda"), 10, capacity=8)
    k = tr.kernels
    scale, base = tr._box_affine
    for n in sizes:                                             # SURVEY 8d's micro-benchmark sizes; `step65k`: 65 536 alone (round 6: the
                                                                # streaming rollout launches the same grid at 65 536 and 2^20 lanes)
        v = VecEnv(k, n, torch.device("cuda"), seed=3, stats_cap=64)
        v.reset()
        rows = torch.zeros(8 * n, k.ring_floats, device="cuda")
        ap = torch.zeros(n, device="cuda")
        batch = torch.zeros(n if n > 4096 else 256, k.row_floats, device="cuda")
        for _ in range(reps):
            k.act_project(v.obs, ap, None, v.action, None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10.0, 10.0, 10, 2e-2, 1e-5,
                          0.0, v.seed, 0, v.ctrl, v.stats)
            k.step(v.internal, v.obs, v.action, v.ep_len, v.ep_ret, v.ep_count, rows, 8, v.stats, v.ctrl, 200, True, 1e-3,
                   v.seed, 0)
            ops.replay_sample_gather(rows, 8, n, batch, None, 1, 0, v.ctrl)
            if n >= 65536:                                      # the one-launch rollout in its streaming form (rollout_stream.hip, round 6)
                k.rollout(tr.fused.descs["actor"], False, scale, base, v.internal, None, v.action, v.ep_len, v.ep_ret, v.ep_count,
                          rows, 8, v.stats, v.ctrl, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10.0, 10.0, 10, 2e-2, 1e-5, 0.0, 200, True,
                          1e-3, v.seed, 0)
        torch.cuda.synchronize()


def probe_mlp(reps=5):
    from rpo_amd import ops
    from rpo_amd.algo.model import ActionEmbedding, SharedPolicy, SharedValueAdd, StateEmbedding
    from test_mlp_gpu import aligned_params, desc_for
    S, A, E, H = 6, 2, 128, 256
    actor = aligned_params(SharedPolicy(S, 1, StateEmbedding(S, E, H), E, H, 1, None))
    critic = aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H))
    da, dc = desc_for(ops, actor, "actor", S, 0, E, H), desc_for(ops, critic, "add", S, A, E, H)
    for n in (256, 4096, 65536):
        s, a = torch.randn(n, S, device="cuda"), torch.randn(n, A, device="cuda")
        out = torch.empty(n, 1, device="cuda")
        x0, h1 = torch.empty(n, E, device="cuda"), torch.empty(n, H, device="cuda")
        dh, dx0, dA = torch.empty(n, H, device="cuda"), torch.empty(n, E, device="cuda"), torch.empty(n, A, device="cuda")
        dout = torch.randn(n, 1, device="cuda")
        for _ in range(reps):
            ops.mlp_forward(da, s, None, out)
            ops.mlp_forward(dc, s, a, out, x0, h1)
            if n <= 4096:
                ops.mlp_backward(dc, s, a, x0, h1, dout, dh, dx0, dA)
        torch.cuda.synchronize()


def probe_iter(iters=24):
    """Whole iterations of the bench trainer, eagerly (every kernel is a separate dispatch the profiler can attribute)."""
    from bench import make_trainer
    tr = make_trainer(4096, torch.device("cuda"), 10 ** 9, capacity=64)
    tr._graphs.enabled = False
    tr.vec.reset()
    tr.run_steps(iters)
    torch.cuda.synchronize()


def probe_ride(periods=6):
    """policy_fre periods of the cart-SAC bench trainer as its graph windows issue them (the rollout riding on the critic
    update's launches), eagerly."""
    from bench import make_trainer
    tr = make_trainer(4096, torch.device("cuda"), 10 ** 9, capacity=64, workload="cart_sac")
    tr._graphs.enabled = False
    tr.vec.reset()
    tr.run_steps(tr.policy_fre)
    assert tr._ride_ok(True)
    for _ in range(periods):
        t0 = tr._t
        tr._sync_uclock(rollout_pending=True)
        tr._uclock_ok = True
        tr._ridden_window(t0, tr.policy_fre)
        for i in range(tr.policy_fre):
            tr._advance_host(t0 + i + 1)
        tr._updates += tr.policy_fre
    torch.cuda.synchronize()


def probe_window(workload, periods=6):
    """policy_fre periods of the bench trainer of `workload` with the launches its graph windows issue (the next vector step
    riding on the update's launches where the schedule does that), eagerly: every kernel is a dispatch of its own that the
    profiler's counters attribute.  Same recipe as bench.kernel_clinic."""
    from bench import envs_per_gpu, make_trainer
    tr = make_trainer(envs_per_gpu(workload), torch.device("cuda"), 10 ** 9, capacity=64, workload=workload)
    tr._graphs.enabled = False
    tr.vec.reset()
    tr.run_steps(2 * tr.policy_fre)
    tr._flush_tail()
    for _ in range(periods):
        t0 = tr._t
        if getattr(tr, "_ride_ok", lambda d: False)(True):
            tr._sync_uclock(rollout_pending=True)
            tr._uclock_ok = True
            tr._ridden_window(t0, tr.policy_fre)
            for i in range(tr.policy_fre):
                tr._advance_host(t0 + i + 1)
            tr._updates += tr.policy_fre
        else:
            for i in range(tr.policy_fre):
                tr._iteration(False, True, (t0 + i + 1) % tr.policy_fre == 0)
                tr._advance_host(t0 + i + 1)
                tr._updates += 1
    torch.cuda.synchronize()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "step"
    if what.startswith("window:"):
        probe_window(what.split(":", 1)[1])
    else:
        {"step": probe_step, "step65k": lambda: probe_step(sizes=(65536,)), "mlp": probe_mlp, "iter": probe_iter,
         "ride": probe_ride}[what]()
