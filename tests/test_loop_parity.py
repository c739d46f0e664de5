"""The shipped trainer's host loop against the oracle loop over MANY iterations, with every random draw shared.

The reference fixtures pin four updates on a fixed buffer; this test pins what they cannot: rollout -> store -> sample ->
update over 40 whole iterations (10 policy steps, multiplier leaving zero, an episode end with its reset), for the
trainer exactly as shipped (driven by the oracle backend on the CPU) against oracle/rpo_loop.py -- the restatement of
rpo/algo/rpo_ddpg.py:79-205 that the 384-seed statistics show to be distribution-equal to the reference.  The oracle loop
receives the trainer's own Philox draws: exploration noise (seed; env 0, t, STREAM_ACT), reset states (seed; env 0,
episode), replay indices (buffer seed; slot, t) and the noise of take_action inside the actor loss (seed; slot,
t + SALT_ACTOR, STREAM_POLICY).  Float32 round-off in two different formulations of the same arithmetic, amplified by the
dynamics: states, parameters and multipliers agree to 2e-5 / 1e-5 throughout.
"""
import numpy as np
import pytest
import torch

import oracle_backend as ob
from oracle import philox, rpo_loop
from rpo_amd.algo.trainer import _SALT_ACTOR
from test_train_step_golden import COMMON, HP, build_trainer


@pytest.mark.parametrize("steps", [40])
def test_shipped_loop_equals_oracle_loop_with_shared_draws(steps):
    torch.set_num_threads(1)
    torch.manual_seed(5)
    tr = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=1, capacity=4000, use_graph=False)
    seed, bseed = tr.seed, tr.buffer.seed

    class PhiloxCart(rpo_loop.CartAdapter):
        def __init__(self):
            super().__init__(1)
            self.episode = 0

        def reset(self):
            self.state = philox.cart_reset(seed, np.array([0]), np.array([self.episode], dtype=np.uint32)).astype(np.float64)
            self.episode += 1
            self.len = 0
            return self.state[0].copy()

    torch.manual_seed(5)                                        # same construction order => same initial weights
    hp = dict(HP[("ddpg", "cart")])
    hp.pop("eval_lr")
    env = PhiloxCart()
    orc = rpo_loop.OracleRPO(env, sac=False, eval_lr=2e-2, eval_steps=50, capacity=4000,
                             **{k: v for k, v in COMMON.items() if k not in ("eval_steps", "capacity")}, **hp)
    for k, v in tr.agent.actor.state_dict().items():
        assert torch.equal(v, orc.nets.actor[k].detach()), k

    def noise_fn(shape, tag):
        if tag == "rollout":
            r = philox.draw(seed, np.array([0]), orc.t, philox.STREAM_ACT)
        else:
            r = philox.draw(seed, np.arange(shape[0]), orc.t + _SALT_ACTOR, philox.STREAM_POLICY)
        return torch.tensor(philox.normal(r[:, 0], r[:, 1])).reshape(shape)
    orc.noise_fn = noise_fn
    orc.index_fn = lambda size, num: philox.sample_indices(bseed, num, orc.t, 0, size)
    tr.vec.reset()
    worst = dict(state=0.0, actor=0.0, critic=0.0, nu=0.0)
    for _ in range(steps):
        tr.run_steps(1)
        orc.run(1)
        worst["state"] = max(worst["state"], float(np.abs(tr.vec.internal.numpy()[0] - np.asarray(orc.state, dtype=np.float32)).max()))
        worst["actor"] = max(worst["actor"], max(float((tr.agent.actor.state_dict()[k] - orc.nets.actor[k].detach()).abs().max())
                                                 for k in orc.nets.actor))
        worst["critic"] = max(worst["critic"], max(float((tr.agent.critic.state_dict()[k] - orc.nets.critic[k].detach()).abs().max())
                                                   for k in orc.nets.critic))
        worst["nu"] = max(worst["nu"], float((tr.agent.nju.weight.detach() - orc.nju.detach()).abs().max()))
    assert env.episode >= 2 and float(orc.nju.detach().max()) > 0.05     # an episode ended; the multiplier left zero
    assert worst["state"] < 2e-5 and worst["actor"] < 1e-5 and worst["critic"] < 1e-5 and worst["nu"] < 1e-5, worst


def test_constructor_refuses_what_it_cannot_reproduce(monkeypatch):
    """corr_mode=1 (rpo_ddpg.py:276-278) is a shape error in the reference for every env (Dual.forward on action-sized
    gradients), so it is refused when the trainer is built, not at the first projection.  A SpringPendulum training batch
    beyond 1024 rows cannot use the reference's sample-coupled projection: refused unless row-wise projection is asked for
    explicitly, and the mode is recorded on the trainer."""
    dev = torch.device("cpu")
    with pytest.raises(ValueError, match="corr_mode"):
        build_trainer("ddpg", "cart", ob, dev, fused=False, num_envs=1, corr_mode=1)
    with pytest.raises(ValueError, match="RPO_ROWWISE_PROJECTION"):
        build_trainer("sac", "pendulum", ob, dev, fused=False, num_envs=4, batch_size=2048, capacity=600)
    assert build_trainer("sac", "pendulum", ob, dev, fused=False, num_envs=1).projection_mode == "batch-reference"
    assert build_trainer("ddpg", "cart", ob, dev, fused=False, num_envs=4, batch_size=2048, capacity=600).projection_mode == "row-wise"
    monkeypatch.setenv("RPO_ROWWISE_PROJECTION", "1")
    tr = build_trainer("sac", "pendulum", ob, dev, fused=False, num_envs=4, batch_size=2048, capacity=600)
    assert tr.projection_mode == "row-wise" and not tr.batch_reference
