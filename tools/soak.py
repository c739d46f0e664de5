#!/usr/bin/env python
"""Soak run: many iterations of a bench workload through the graph windows, then consistency checks of every device-side
counter the cross-launch hand-overs maintain (rollout clock, update clock, optimiser step counters, arrival words at rest,
gradmax slots) and of the parameters / statistics.    python tools/soak.py [workload] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cart_ddpg"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    dev = torch.device("cuda")
    n = bench.envs_per_gpu(workload)
    tr = bench.make_trainer(n, dev, 10 ** 9, workload=workload)
    tr.vec.reset()
    tr.run_steps(steps)
    tr._harvest(final=True)
    torch.cuda.synchronize()
    ag = tr.agent
    T = hip_T = int(tr.vec.ctrl[0])
    assert T == steps == tr._t, (T, steps, tr._t)
    if tr._uctrl is not tr.vec.ctrl:
        assert int(tr._uctrl[0]) == steps + 1, int(tr._uctrl[0])
    assert int(tr.vec.ctrl[1]) == 0 and int(tr.vec.ctrl[16:].abs().max()) == 0          # arrival counters at rest
    from rpo_amd import _lib
    assert int(tr.vec.ctrl[_lib.CONST["RPO_CTRL_NONFINITE"]]) == 0                       # the failure word stayed clear (round 6)
    assert int(ag.critic_optim.step_dev[0]) == steps - tr.warmup + (1 if tr.warmup else 0) or int(ag.critic_optim.step_dev[0]) == steps
    assert int(ag.actor_optim.step_dev[0]) == steps // tr.policy_fre
    for opt in (ag.critic_optim, ag.actor_optim):
        assert int(opt.step_dev[2]) == 0 and int(opt.step_dev[32:].abs().max()) == 0
    su = getattr(tr, "_split_cache", None)
    front = bool(getattr(tr, "_front_cache", False))
    if su and su._held.get("tile_sync") is not None:                                 # fused front launches: words at rest, no wait gave up
        assert int(su._held["tile_sync"].abs().max()) == 0
    ws = su._held.get("proj_ws") if su else None
    pfront = bool(getattr(tr, "_pfront", False))
    if ws is not None:                                          # granule workspace of the batch-coupled projection: no wait gave
        from rpo_amd import ops                                 # up, the readers' count at rest, one epoch per launch
        assert int(ws[ops.PROJ_WS_GAVE_UP]) == 0 and int(ws[ops.PROJ_WS_GAVE_UP + 1]) == 0
        assert int(ws[ops.PROJ_WS_GAVE_UP - 1]) == int(ag.critic_optim.step_dev[0]), (int(ws[ops.PROJ_WS_GAVE_UP - 1]), steps)
    for p in (ag.flat.data, ag.nju.weight, ag.critic_target_flat):
        assert bool(torch.isfinite(p).all())
    assert tr.env_steps == steps * n and 0.0 <= tr.viol_rate <= 1.0
    print("%s: %d steps ok; violation rate %.5f, ride %s, fused front %s, pendulum front with granules %s, prepared Adam, |params| max %.3f" % (
        workload, steps, tr.viol_rate, bool(getattr(tr, "_ride_ok", lambda d: False)(True)), front, pfront, float(ag.flat.data.abs().max())))


if __name__ == "__main__":
    main()
