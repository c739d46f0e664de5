"""Exact resume: save() after k iterations, load() into a freshly constructed trainer, continue -- the
result equals the uninterrupted run bit for bit (parameters, targets, optimiser state, env lanes, replay ring, Philox
counters).  CPU: through the oracle backend; GPU: through the HIP kernels with hipGraph replay."""
import numpy as np
import pytest
import torch

import oracle_backend as ob
from test_train_step_golden import build_trainer


def _resume_roundtrip(algo, envname, backend, device, tmp_path, n_envs, first, second, **kw):
    def fresh():
        torch.manual_seed(5)
        tr = build_trainer(algo, envname, backend, device, num_envs=n_envs, **kw)
        tr.work_dir = str(tmp_path / "ckpt")
        return tr
    a = fresh()
    a.vec.reset()
    a.run_steps(first + second)
    b = fresh()
    b.vec.reset()
    b.run_steps(first)
    b.save()
    c = fresh()                    # new process stand-in: nothing shared with b but the files
    c.load()
    c.run_steps(second)
    if device.type == "cuda":
        torch.cuda.synchronize()
    for name in ("internal", "ep_len", "ep_ret", "ep_count"):
        assert torch.equal(getattr(a.vec, name), getattr(c.vec, name)), name
    assert torch.equal(a.buffer.rows, c.buffer.rows)
    assert torch.equal(a.agent.flat.data, c.agent.flat.data)
    assert torch.equal(a.agent.critic_target_flat, c.agent.critic_target_flat)
    assert torch.equal(a.agent.critic_optim.exp_avg_sq, c.agent.critic_optim.exp_avg_sq)
    assert torch.equal(a.agent.nju.weight, c.agent.nju.weight)
    assert int(c.vec.ctrl[0]) == first + second == c._t
    a._harvest(), c._harvest()
    assert a.env_steps == c.env_steps and abs(a.viol_rate - c.viol_rate) < 1e-12


@pytest.mark.parametrize("algo,envname", [("ddpg", "cart"), ("sac", "pendulum")])
def test_resume_is_exact_on_oracle_backend(algo, envname, tmp_path):
    torch.set_num_threads(1)
    _resume_roundtrip(algo, envname, ob, torch.device("cpu"), tmp_path, 4, 9, 7, use_graph=False, capacity=8)


def test_load_rejects_a_checkpoint_of_another_configuration(tmp_path):
    torch.set_num_threads(1)
    torch.manual_seed(5)
    a = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=4, use_graph=False, capacity=8)
    a.work_dir = str(tmp_path / "ckpt")
    a.vec.reset()
    a.run_steps(3)
    a.save(replay=False)
    b = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=8, use_graph=False, capacity=8)
    b.work_dir = a.work_dir
    with pytest.raises(ValueError, match="num_envs"):
        b.load()
    # a larger ring would silently remap the stored rows (position = t mod capacity): refused, nothing modified
    torch.manual_seed(5)
    c = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=4, use_graph=False, capacity=16)
    c.work_dir = a.work_dir
    before = c.agent.flat.data.clone()
    with pytest.raises(ValueError, match="capacity"):
        c.load()
    assert torch.equal(before, c.agent.flat.data) and c._t == 0
    # same configuration, but the checkpoint holds no replay rows: training cannot resume (the sampler would draw
    # zeros); the networks alone can be restored
    torch.manual_seed(6)
    d = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=4, use_graph=False, capacity=8)
    d.work_dir = a.work_dir
    with pytest.raises(ValueError, match="replay=False"):
        d.load()
    assert not torch.equal(d.agent.flat.data, a.agent.flat.data)
    d.load(weights_only=True)
    assert torch.equal(d.agent.flat.data, a.agent.flat.data) and d._t == 0 and int(d.vec.ctrl[0]) == 0
    # another algorithm on the same env
    e = build_trainer("sac", "cart", ob, torch.device("cpu"), num_envs=4, use_graph=False, capacity=8)
    e.work_dir = a.work_dir
    with pytest.raises(ValueError, match="algo"):
        e.load()


@pytest.mark.gpu
@pytest.mark.parametrize("algo,envname", [("ddpg", "cart"), ("sac", "pendulum"), ("ddpg", "evopf256")])
def test_resume_is_exact_on_gpu(algo, envname, tmp_path):
    from rpo_amd import ops
    _resume_roundtrip(algo, envname, ops, torch.device("cuda"), tmp_path, 64, 14, 10, use_graph=True, capacity=16)
