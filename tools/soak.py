#!/usr/bin/env python
"""Soak run: many iterations of a bench workload through the graph windows, then consistency checks of every device-side
counter the cross-launch hand-overs maintain (rollout clock, update clock, optimiser step counters, arrival words at rest,
gradmax slots) and of the parameters / statistics.    python tools/soak.py [workload] [steps] [--large-batch]
--large-batch: the update batch is 256 x lanes rows per vector step (the mode whose split-K scratch a hipGraph
memset node corrupted until round 6); additionally every padding float of the flat parameter buffer, of its gradient and of the
optimisers' moments must still be exactly zero at the end, and a second run of the same length must end with the same bits."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    large = "--large-batch" in sys.argv
    workload = argv[0] if len(argv) > 0 else "cart_ddpg"
    steps = int(argv[1]) if len(argv) > 1 else 100000
    dev = torch.device("cuda")
    n = bench.envs_per_gpu(workload)
    if large:
        return large_batch(workload, steps, n, dev)
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


def large_batch(workload, steps, n, dev):
    from rpo_amd import _lib
    ends = []
    for rep in range(2):
        tr = bench.make_trainer(n, dev, 10 ** 9, capacity=64, workload=workload, torch_seed=123, seed=7000, batch_size=256 * n)
        tr.vec.reset()
        tr.run_steps(steps)
        tr._harvest(final=True)
        torch.cuda.synchronize()
        assert int(tr.vec.ctrl[0]) == steps == tr._t
        assert int(tr.vec.ctrl[_lib.CONST["RPO_CTRL_NONFINITE"]]) == 0
        assert any(e["graph"] is not None for e in tr._graphs.entries.values()) and not tr._graphs.capture_failed
        fl = tr.agent.flat
        pad = torch.ones(fl.total, dtype=torch.bool, device=dev)
        for mod in (tr.agent.actor, tr.agent.critic, tr.agent.nju):
            for p_ in mod.parameters():
                o = fl.offset.get(id(p_))
                if o is not None:
                    pad[o:o + p_.numel()] = False
        assert int(pad.sum()) >= 3 and not bool(fl.data[pad].any()) and not bool(fl.grad[pad].any())
        for opt, (lo, hi) in ((tr.agent.critic_optim, fl.critic_range), (tr.agent.actor_optim, fl.actor_range)):
            assert not bool(opt.exp_avg[pad[lo:hi]].any()) and not bool(opt.exp_avg_sq[pad[lo:hi]].any())
        assert bool(torch.isfinite(fl.data).all())
        ends.append((fl.data.clone(), tr.agent.nju.weight.detach().clone(), tr.viol_rate))
        del tr
        torch.cuda.empty_cache()
    assert torch.equal(ends[0][0], ends[1][0]) and torch.equal(ends[0][1], ends[1][1]), "two runs of the same length differ"
    print("%s, large batch (%d rows per update): 2 x %d steps ok; same bits at the end of both runs, padding floats zero, failure "
          "word clear; violation rate %.5f, |params| max %.3f" % (workload, 256 * n, steps, ends[0][2], float(ends[0][0].abs().max())))


if __name__ == "__main__":
    main()
