"""The evaluation protocol (rpo_ddpg.py:207-264, rpo_sac.py:221-278) against the reference's own 10-tuple.

Fixtures ``tests/golden/eval_*.npz`` (``make_golden.py eval``): a policy trained by the unmodified reference for 400
iterations with the scripts' hyper-parameters, 10 injected initial states, the 10-tuple ``eval()`` returned and the
per-episode summaries it is built from.  ``*_sat`` shift the actor's output bias so that the evaluation-time projection
(50 GRG steps) leaves a non-zero inequality violation.  Here the same state_dict and initial states go through the
shipped ``eval()``: on the CPU suite driven by the oracle backend (float64 dynamics, like the reference), on the GPU
through the HIP kernels (float32 dynamics).

Tolerances.  An episode's return is its length (cart: reward 1 per step) or a sum of O(1) rewards; float32 dynamics
on the GPU may end an episode one step earlier or later when the terminating state sits within round-off of the
threshold, so the GPU leg allows the mean return to move by one step of one episode in ten (0.1 for cart) and the std
accordingly; violations agree to 2e-5 relative + 2e-6 absolute (float32 constraint arithmetic on both sides).
"""
import numpy as np
import pytest
import torch

import oracle_backend as ob
from test_train_step_golden import build_trainer

CASES = [("ddpg", "cart", ""), ("sac", "pendulum", ""), ("sac", "cart", ""), ("ddpg", "pendulum", ""),
         ("ddpg", "cart", "_sat"), ("ddpg", "pendulum", "_sat")]


def run_eval(golden, algo, envname, tag, backend, device):
    g = golden("eval_%s_%s%s" % (algo, envname, tag))
    torch.manual_seed(1)
    tr = build_trainer(algo, envname, backend, device, num_envs=1, use_graph=False)
    sd = {k[len("actor."):]: torch.tensor(g[k]) for k in g.files if k.startswith("actor.")}
    tr.agent.actor.load_state_dict(sd)
    tr._eval_init_inject = torch.tensor(g["init"], dtype=torch.float32, device=device)
    return g, np.array(tr.eval(), dtype=np.float64)


def check(g, res, ret_tol):
    want = g["result"]
    # (mean, std) of: return, mean ineq, mean eq, max ineq, max eq
    np.testing.assert_allclose(res[0], want[0], rtol=0, atol=ret_tol[0], err_msg="mean return")
    np.testing.assert_allclose(res[1], want[1], rtol=0, atol=ret_tol[1], err_msg="std return")
    for i, name in ((2, "mean_ineq"), (3, "mean_ineq std"), (6, "max_ineq"), (7, "max_ineq std")):
        np.testing.assert_allclose(res[i], want[i], rtol=ret_tol[2], atol=2e-6, err_msg=name)
    for i in (4, 5, 8, 9):                                  # equality residuals are round-off on both sides
        assert abs(res[i]) < 2e-5 and abs(want[i]) < 2e-5


@pytest.mark.parametrize("algo,envname,tag", CASES)
def test_eval_matches_reference_on_oracle_backend(golden, algo, envname, tag):
    """Host logic of eval() (lanes, alive masks, running means, horizon) with float64 dynamics: exact returns."""
    torch.set_num_threads(1)
    g, res = run_eval(golden, algo, envname, tag, ob, torch.device("cpu"))
    check(g, res, (1e-4, 1e-4, 2e-5))
    assert len(res) == 10


@pytest.mark.gpu
@pytest.mark.parametrize("algo,envname,tag", CASES)
def test_eval_matches_reference_on_gpu(golden, algo, envname, tag):
    from rpo_amd import ops
    g, res = run_eval(golden, algo, envname, tag, ops, torch.device("cuda"))
    step = 1.0 if envname == "cart" else float(np.max(np.abs(g["ep_return"] / np.maximum(g["ep_length"], 1))))
    # one episode in ten one step longer or shorter; the "mean ineq" entries are running means over the episode, so
    # a one-step change of an episode's length moves them by ~1/length
    check(g, res, (0.1 * step + 1e-4, 0.35 * step + 1e-4, 2e-2 if tag else 2e-5))
