"""The Lagrangian baselines DDPG_LA / SAC_LA (rpo/algo/ddpg_lag.py, sac_lag.py) on the HIP kernels: the reference's update
steps on the GPU, whole iterations on all three envs, and the equality-residual backward of EVOPF-v0."""
import os

import numpy as np
import pytest
import torch

import test_train_step_golden as tsg
from oracle import evopf as oe

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("algo,envname", [("ddpgla", "cart"), ("sacla", "pendulum")])
def test_update_matches_reference_on_gpu(algo, envname):
    from rpo_amd import ops
    g, tr, closs, aloss, proxy = tsg.run_product_update(_golden, algo, envname, ops, torch.device("cuda"), fused=False)
    ag = tr.agent
    np.testing.assert_allclose(closs, g["critic_losses"], rtol=1e-4)
    np.testing.assert_allclose(aloss, g["actor_losses"], rtol=1e-4, atol=1e-6)
    for name, net in (("critic4", ag.critic), ("actor4", ag.actor), ("critic_target4", ag.critic_target)):
        for k, v in tsg.sd(g, name).items():
            np.testing.assert_allclose(net.state_dict()[k].cpu().numpy(), v, rtol=0, atol=5e-6, err_msg=name + "." + k)
    np.testing.assert_allclose(ag.lamb.weight.detach().cpu().numpy(), g["lamb4"], rtol=1e-4)
    np.testing.assert_allclose(ag.nju.weight.detach().cpu().numpy(), g["nju4"], rtol=1e-4)


def test_evopf_eq_vjp_matches_oracle_and_autograd_sign():
    from rpo_amd.env import EVOPFEnv
    env = EVOPFEnv(device="cuda")
    rng = np.random.RandomState(2)
    ids = np.arange(40)
    s = np.concatenate([oe.episode_demand(5, ids, 0, 7), rng.uniform(0.1, 0.8, size=(40, 5)), oe.episode_price(5, ids, 0, 7)], axis=1)
    lo, hi = oe.partial_box(s)
    ap = lo + 0.5 * (hi - lo)
    ap[:, 4:9] = rng.uniform(1.0, 1.06, size=(40, 5))
    a = (oe.complete_partial(s, ap) + 0.02 * rng.randn(40, 43)).astype(np.float32)
    w = rng.randn(40, 28).astype(np.float32)
    at = torch.tensor(a, device="cuda", requires_grad=True)
    env.eq_resid(torch.tensor(s, dtype=torch.float32, device="cuda"), at).backward(torch.tensor(w, device="cuda"))
    jac = oe.eq_jac(a.astype(np.float64))
    jac[:, :, oe.GRID.pe0:] *= -1.0                              # autograd differentiates the residual's +pe (hazard E1)
    want = np.einsum("nev,ne->nv", jac, w.astype(np.float64))
    np.testing.assert_allclose(at.grad.cpu().numpy(), want, rtol=1e-4, atol=2e-4)
    # finite-difference check of one column per block against the residual itself
    h = 1e-3
    for v in (1, 7, 12, 30, 40):
        e = np.zeros(43)
        e[v] = h
        fd = ((oe.eq_resid(s, a + e) - oe.eq_resid(s, a - e)) / (2 * h) * w).sum(axis=1)
        np.testing.assert_allclose(at.grad.cpu().numpy()[:, v], fd, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("algo,envname", [("ddpg", "cart"), ("sac", "pendulum"), ("ddpg", "evopf"), ("sac", "evopf")])
def test_baselines_run_on_all_envs(algo, envname):
    from rpo_amd import gym_shim, ops
    from rpo_amd.algo import DDPG_LA, SAC_LA
    from rpo_amd.env import CartSafeEnv, EVOPFEnv, SpringPendulumEnv
    torch.manual_seed(1)
    if envname == "evopf":
        env = EVOPFEnv(device="cuda")
    else:
        env = gym_shim.TimeLimit((CartSafeEnv if envname == "cart" else SpringPendulumEnv)(device="cuda"), 200)
    kw = dict(automatic_entropy_tuning=False, alpha=0.05) if algo == "sac" else {}
    tr = (DDPG_LA if algo == "ddpg" else SAC_LA)(
        env, "/tmp/rpo_la", name="la", logger=None, warmup=4, batch_size=64, capacity=64, embed_dim=64, hidden_dim=64,
        policy_fre=2, max_epochs=40, init_nju=0.1, init_lamb=0.1, lr_dual=0.05, shape=True, value_type="cat",
        shared_param=False, num_envs=32, seed=3, **kw)
    tr.vec.reset()
    tr.run_steps(30)
    tr._harvest()
    assert int(tr.vec.ctrl[0]) == 30 and tr.env_steps == 32 * 30
    assert tr.viol_rate > 0.5                                    # no equation solver: the equalities are violated
    assert abs(float(tr.agent.lamb.weight.reshape(-1)[0]) - 0.1) > 1e-3      # lambda is stepped (ddpg_lag.py:197)
    for p in (tr.agent.flat.data, tr.agent.nju.weight, tr.agent.lamb.weight):
        assert bool(torch.isfinite(p).all())
    res = tr.eval()
    assert len(res) == 10 and np.isfinite(res).all()


@pytest.mark.parametrize("algo,envname", [("ddpg", "cart"), ("sac", "pendulum")])
def test_baseline_graph_windows_equal_eager(algo, envname, monkeypatch):
    """The baselines' iterations (torch modules + autograd + the fused optimiser / env kernels) are captured in hipGraphs
    like the RPO trainers': windows of 8 iterations, single iterations for the ragged parts -- same bits as eager launches."""
    from rpo_amd import gym_shim
    from rpo_amd.algo import DDPG_LA, SAC_LA
    from rpo_amd.env import CartSafeEnv, SpringPendulumEnv
    out = []
    for graph in ("1", "0"):
        monkeypatch.setenv("RPO_GRAPH_CYCLE", "8" if graph == "1" else "0")     # (0: no hipGraphs, eager launches)
        torch.manual_seed(1)
        env = gym_shim.TimeLimit((CartSafeEnv if envname == "cart" else SpringPendulumEnv)(device="cuda"), 200)
        kw = dict(automatic_entropy_tuning=False, alpha=0.05) if algo == "sac" else {}
        tr = (DDPG_LA if algo == "ddpg" else SAC_LA)(
            env, "/tmp/rpo_la", name="la", logger=None, warmup=0, batch_size=64, capacity=128, embed_dim=64, hidden_dim=64,
            policy_fre=2, max_epochs=100, init_nju=0.1, init_lamb=0.1, lr_dual=0.05, value_type="add", shared_param=True,
            num_envs=32, seed=3, **kw)
        tr.vec.reset()
        tr.run_steps(70)
        torch.cuda.synchronize()
        out.append(tr)
    a, b = out
    assert a._graphs.enabled and any(k[0] == "cycle" and e["graph"] is not None for k, e in a._graphs.entries.items()
                                     if isinstance(k, tuple))
    assert not b._graphs.enabled
    assert torch.equal(a.agent.flat.data, b.agent.flat.data) and torch.equal(a.buffer.rows, b.buffer.rows)
    assert torch.equal(a.agent.nju.weight, b.agent.nju.weight) and torch.equal(a.agent.lamb.weight, b.agent.lamb.weight)
    assert torch.equal(a.vec.internal, b.vec.internal) and int(a.vec.ctrl[0]) == 70
