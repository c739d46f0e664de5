"""The N>1 path on CPU: world_size-2 `gloo`, trainer host logic driven by the oracle backend (tests/oracle_backend.py).

Checks: env lanes shard by global env id (rank r owns ids [r*n, (r+1)*n)) and reproduce the single-process trajectories;
the flat gradient slices are all-reduced to the mean of the per-rank gradients; replicas stay bit-identical through
several updates with no parameter traffic after the one-off broadcast of rank 0's initial state; statistics are reduced
over ranks.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, algo, envname, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RPO_VERBOSE="0")
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_backend as ob
    from test_train_step_golden import build_trainer
    # rank 1 deliberately seeds differently: the trainer makes rank 0's initial weights, targets, multipliers and Philox
    # seed everybody's at construction (one broadcast each), so the replicas still start identical
    torch.manual_seed(5 + 17 * rank)
    tr = build_trainer(algo, envname, ob, torch.device("cpu"), num_envs=16)
    seeds = [None, None]
    dist.all_gather_object(seeds, (tr.seed, float(tr.agent.flat.data.double().sum())))
    assert seeds[0] == seeds[1]
    assert tr.n_local == 8 and tr.vec.env_id_base == 8 * rank and tr.dist.world == 2
    tr.vec.reset()
    first_state = tr.vec.internal.clone()
    tr.run_steps(1)
    after_one = tr.vec.internal.clone()
    # gradient bucket: per-rank critic gradient before the collective vs the reduced one
    cols = tr._sample()
    tr._critic_update(cols)
    fl = tr.agent.flat
    local = fl.gradient(fl.critic_range).clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    tr.dist.mean_([fl.gradient(fl.critic_range)])
    np.testing.assert_allclose(fl.gradient(fl.critic_range).numpy(), (sum(gathered) / world).numpy(), rtol=1e-6, atol=1e-9)
    assert not torch.equal(gathered[0], gathered[1])            # the shards really sampled different data
    tr.run_steps(7)                                             # includes two policy steps (t = 4, 8)
    tr._harvest(final=True)
    _check_graph_agreement(tr, rank)
    torch.save(dict(first=first_state, after_one=after_one, flat=fl.data.clone(), nju=tr.agent.nju.weight.data.clone(),
                    target=tr.agent.critic_target_flat.clone(), env_steps=float(tr.env_steps), viol=float(tr.viol_steps)),
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _check_graph_agreement(tr, rank):
    """A hipGraph capture that fails on ONE rank takes every rank to eager launches together (_GraphCache.run): ranks that
    replay captured collectives cannot pair with a rank that issues them eagerly.  No GPU here: torch.cuda.graph is replaced
    by a stand-in whose capture raises on rank 1 only; both ranks must end up eager, with their host state restored, `fn`
    executed exactly once for real, and the bench's "every rank replayed graphs" flag false on both."""
    from rpo_amd.algo import trainer as T
    assert tr.dist.all_ok(True, tr.device) and not tr.dist.all_ok(rank == 0, tr.device)
    calls, restored = [], []

    class FakeGraph(object):
        def replay(self):
            calls.append("replay")

    class FakeCapture(object):
        def __init__(self, g, **kw):
            pass

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            if et is None and rank == 1:
                raise RuntimeError("capture failed on this rank")
            return False
    real = (torch.cuda.CUDAGraph, torch.cuda.graph, torch.cuda.synchronize)
    torch.cuda.CUDAGraph, torch.cuda.graph, torch.cuda.synchronize = FakeGraph, FakeCapture, lambda *a: None
    orig_set = tr._set_host_state
    tr._set_host_state = lambda st: (restored.append(True), orig_set(st))
    try:
        cache = T._GraphCache(True, warm=0, owner=tr)
        cache.run("window", lambda: calls.append("fn"))
        assert calls == ["fn", "fn"] and restored == [True]      # once inside the (dropped) capture, once eagerly
        assert not cache.enabled and cache.capture_failed
        cache.run("window", lambda: calls.append("eager"))
        assert calls[-1] == "eager"
        assert not tr.dist.replaying_everywhere(cache, tr.device)
    finally:
        torch.cuda.CUDAGraph, torch.cuda.graph, torch.cuda.synchronize = real
        tr._set_host_state = orig_set


@pytest.mark.parametrize("algo,envname", [("ddpg", "cart"), ("sac", "pendulum")])
def test_two_rank_data_parallel(tmp_path, algo, envname):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, algo, envname, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"), weights_only=False)
    # replicas identical, bit for bit, after 8 iterations with all-reduced gradients
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["nju"], r1["nju"]) and torch.equal(r0["target"], r1["target"])
    assert r0["env_steps"] == r1["env_steps"] == 16 * 8          # statistics are summed over ranks
    assert r0["viol"] == r1["viol"]
    # single-process run with the same 16 global lanes: same initial states and same first vector step per env id
    sys.path.insert(0, HERE)
    import oracle_backend as ob
    from test_train_step_golden import build_trainer
    torch.set_num_threads(1)
    torch.manual_seed(5)
    tr = build_trainer(algo, envname, ob, torch.device("cpu"), num_envs=16)
    tr.vec.reset()
    both_first = torch.cat([r0["first"], r1["first"]])
    assert torch.equal(tr.vec.internal, both_first)
    tr.run_steps(1)
    assert torch.equal(tr.vec.internal, torch.cat([r0["after_one"], r1["after_one"]]))


def test_updates_per_step_cadence():
    """updates_per_step = G: G updates per vector step, the policy cadence follows the update counter, every extra
    update draws its own replay batch (4th Philox counter word = update index within the step)."""
    sys.path.insert(0, HERE)
    import oracle_backend as ob
    from test_train_step_golden import build_trainer
    torch.set_num_threads(1)
    torch.manual_seed(5)
    tr = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=8, updates_per_step=3)
    seen = []
    orig = ob.replay_sample_gather

    def spy(rows, cap_steps, n_envs, out, idx_out, seed, salt, ctrl):
        seen.append((int(ctrl[0]), int(ctrl[2])))
        return orig(rows, cap_steps, n_envs, out, idx_out, seed, salt, ctrl)
    tr.buffer._ops = type("P", (), {"replay_sample_gather": staticmethod(spy)})()
    tr.vec.reset()
    tr.run_steps(4)
    assert int(tr.agent.critic_optim.step_dev[0]) == 12 and int(tr.agent.actor_optim.step_dev[0]) == 3
    assert seen == [(t, k) for t in range(1, 5) for k in range(3)]
    assert int(tr.vec.ctrl[2]) == 0 and int(tr.vec.ctrl[0]) == 4


def test_warmup_phase_uses_uniform_actions_and_delays_training():
    """rpo_ddpg.py:97-101,160-161: for t < warmup the basic action is BoxConstraint.sample (uniform in the box), and
    train(t) starts at loop index t == warmup."""
    sys.path.insert(0, HERE)
    import oracle_backend as ob
    from oracle import philox
    from test_train_step_golden import build_trainer
    torch.set_num_threads(1)
    torch.manual_seed(5)
    tr = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=8, warmup=3)
    tr.vec.reset()
    tr.run_steps(1)
    r = philox.draw(tr.seed, np.arange(8), 0, philox.STREAM_ACT)
    ap = 10.0 * (2.0 * philox.u01(r[:, 0]) - 1.0)
    free = np.abs(ap) < 6.9                                     # larger draws are moved by the projection afterwards
    np.testing.assert_allclose(tr.buffer.rows[:8, 7].numpy()[free], ap[free], rtol=1e-5, atol=1e-5)   # basic action = index 1
    assert np.all(np.abs(tr.buffer.rows[:8, 7].numpy()[~free]) < np.abs(ap[~free]))
    assert int(tr.agent.critic_optim.step_dev[0]) == 0
    tr.run_steps(1)
    assert int(tr.agent.critic_optim.step_dev[0]) == 0
    tr.run_steps(3)                                              # loop indices 3, 4, 5 train
    assert int(tr.agent.critic_optim.step_dev[0]) == 3 and int(tr.agent.actor_optim.step_dev[0]) == 1
