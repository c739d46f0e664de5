"""Statistical parity with the reference's CPU runs (north_star: returns and constraint-violation rate).

tests/golden/training_stats_*.npz hold the statistics of 3000-iteration training runs of the unmodified reference
(scripts/cart_exp.py: 384 seeds; scripts/pen_exp_sac.py: 192 seeds (round 4; 24 before); scripts/cart_exp_sac.py: 384 seeds (round 4; 96 before); scripts/pen_exp.py: 576 seeds;
tests/golden/make_golden.py stats).  The same runs are
repeated here with the shipped trainers at num_envs = 1 -- the reference's cadence, step for step -- on the HIP kernels,
on TWICE as many seeds (GPU runs are cheap).  Random streams differ (Philox vs numpy/torch global generators) and
trajectories are chaotic, so the comparison is between seed-averaged statistics: violation rate = fraction of env steps
with max(max_ineq, max_eq) > 1e-3.

Resolution.  The seed-to-seed spread of the cart-RPODDPG violation rate is ~3.3e-3, so the standard error of the
difference of the two means is ~4e-4 with 96 + 192 seeds: the north_star's 1e-3 is a 2.4-sigma effect.  The test asserts
(a) that resolution (SE of the difference <= 5e-4) and (b) |rate_gpu - rate_ref| <= 1e-3 + 2 SE, i.e. that a difference
of 1e-3 is not excluded at two sigma (measured: 1.1e-3 +- 0.5e-3).  The mean of the
per-step maximum inequality violation must agree within 15 % + 2 SE (a 70 % gap, as an earlier 5-seed version of this
test could not tell apart, is ~10 SE here), the mean episodic return within 5 % + 2 SE (round 3; with 48 reference
seeds pendulum-RPODDPG had shown -13 % at 1.4 sigma: with 576 against 1152 the difference is +0.6 %, z = 0.2 -- the first 192
reference seeds average 36.9, the next 192 30.8).  The equality constraint must
hold to float32 round-off on every step of every run.
"""
import json
import os

import numpy as np
import pytest
import torch

from test_train_step_golden import build_trainer

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(1500, method="thread")
# (algo, env, largest standard error of the violation-rate difference the case must reach, GPU runs per reference run; the
# two cases with hundreds of reference seeds and a small spread run ONE GPU seed per reference seed: the suite is host-bound
# at num_envs = 1 and took 17 minutes on a slow box with 1152 pendulum-RPODDPG runs).
# cart-RPOSAC (config 4's algorithm, scripts/cart_exp_sac.py) has a seed-to-seed spread of 7.4e-3 -- more than twice
# cart-RPODDPG's: 96 reference seeds resolved 1e-3 at one sigma only (round 3); with 384 reference and 384 GPU runs (round 4)
# the standard error of the difference is 5.9e-4.
@pytest.mark.parametrize("algo,envname,se_max,gpu_per_ref", [("ddpg", "cart", 5e-4, 2), ("sac", "pendulum", 5e-4, 2),
                                                             ("sac", "cart", 6.5e-4, 1), ("ddpg", "pendulum", 5e-4, 1)])
def test_training_statistics_match_reference(golden, algo, envname, se_max, gpu_per_ref):
    from rpo_amd import ops
    from rpo_amd.utils.logger import Logger
    g = golden("training_stats_%s_%s" % (algo, envname))
    ref, steps = g["stats"], int(g["steps"])
    n_gpu = gpu_per_ref * len(ref)
    os.environ["RPO_VERBOSE"] = "0"
    rows = []
    for seed in range(n_gpu):
        torch.manual_seed(123 + seed)
        # every run has its own initial weights (torch seed) AND its own Philox seed: exploration noise, reset states,
        # replay indices and update noise are independent across runs (with one shared Philox seed the runs share one noise
        # realisation, and its luck does not average out: +1e-3 on the violation rate, 2.7 sigma, in an earlier version)
        tr = build_trainer(algo, envname, ops, torch.device("cuda"), num_envs=1, capacity=steps, seed=5000 + seed)
        tr.max_epochs = steps
        tr.logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=steps)
        tr.run(eval=False)
        n = tr.logger.pointer
        mi, me, rw = [tr.logger.tracker[k][:n] for k in ("max_ineq", "max_eq", "reward")]
        viol = np.maximum(mi, me) > 1e-3
        rows.append([n, viol.mean(), mi.mean(), me.mean(), rw.mean(), rw[n // 2:].mean(),
                     float(tr.agent.nju.weight.detach().abs().max())])
        assert me.max() < 1e-4                                  # the equality holds on every step (equation solver)
        assert abs(tr.viol_rate - viol.mean()) < 5e-3           # device-side counter agrees with the logged rows
        del tr
    got = np.array(rows)

    def se(col):
        return float(np.sqrt(ref[:, col].var(ddof=1) / len(ref) + got[:, col].var(ddof=1) / len(got)))
    out = {"ref_seeds": len(ref), "gpu_seeds": len(got), "steps": steps, "ref_mean": ref.mean(0).tolist(),
           "gpu_mean": got.mean(0).tolist(), "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist(),
           "se_of_difference": [se(c) for c in range(ref.shape[1])], "columns": [str(c) for c in g["columns"]]}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/statistical_parity_%s_%s.json" % (algo, envname), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    d_viol = abs(got[:, 1].mean() - ref[:, 1].mean())
    assert se(1) <= se_max, se(1)                                 # 5e-4: the comparison resolves 1e-3 at two sigma
    assert d_viol <= 1e-3 + 2 * se(1), (d_viol, se(1))
    d_ineq = abs(got[:, 2].mean() - ref[:, 2].mean())
    assert d_ineq <= 0.15 * ref[:, 2].mean() + 2 * se(2) + 1e-5, (d_ineq, se(2))
    for col in (4, 5):                                           # episodic return, whole run and second half
        d = abs(got[:, col].mean() - ref[:, col].mean())
        assert d <= 0.05 * ref[:, col].mean() + 2 * se(col), (col, d, se(col))
    assert got[:, 0].min() > 0.9 * steps                         # the logger received (almost) every step


@pytest.mark.timeout(1500, method="thread")
@pytest.mark.parametrize("algo", ["ddpg", "sac"])
def test_evopf_training_statistics_match_reference(golden, algo):
    """EVOPF-v0, RPODDPG / RPOSAC with the hyper-parameters of scripts/evopf_exp.py / evopf_exp_sac.py: 24 seeds x 960
    iterations (40 days) of the reference (on the pypower stand-in, tests/golden/make_evopf_golden.py stats [sac]; 3 seeds
    until round 3) vs TWICE as many runs of the shipped trainer at num_envs = 1 on the HIP kernels (MLP kernels,
    wave-per-lane power flow).  Different random days and exploration streams, so only seed-averaged statistics are compared,
    each within 2 standard errors of the difference + 10 % of the reference's mean (round 3: 3 SE + 0.05 absolute on the rate,
    30 % / 15 % on the others -- a slack that could not see a 10 % bias): violation rate, mean max-inequality violation, mean
    return over the run and over its second half; the equalities hold to the level the reference reaches (GRG drift)."""
    from rpo_amd import ops
    from rpo_amd.algo import RPODDPG, RPOSAC
    from rpo_amd.env import EVOPFEnv
    from rpo_amd.utils.logger import Logger
    import test_train_step_golden as tsg
    g = golden("training_stats_%s_evopf" % algo)
    ref, steps = g["stats"], int(g["steps"])
    assert len(ref) >= 24
    hp = {k: v for k, v in tsg.EVOPF_HP.items() if k not in ("embed_dim", "hidden_dim", "init_nju", "capacity")}
    if algo == "sac":                                            # scripts/evopf_exp_sac.py:29-32
        del hp["gamma"]
        hp.update(grad_eps=0.1, alpha=0.001, automatic_entropy_tuning=False, fixed=False)
    cls = RPODDPG if algo == "ddpg" else RPOSAC
    rows = []
    os.environ["RPO_VERBOSE"] = "0"
    for seed in range(2 * len(ref)):
        torch.manual_seed(123 + seed)
        tr = cls(EVOPFEnv(device="cuda"), "/tmp/rpo_test", name="t", logger=None, max_epochs=steps, capacity=20000,
                 device=torch.device("cuda"), num_envs=1, seed=1000 + seed, **hp)
        assert tr.fused is not None
        tr.logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=steps)
        tr.run(eval=False)
        n = tr.logger.pointer
        mi, me, rw = [tr.logger.tracker[k][:n] for k in ("max_ineq", "max_eq", "reward")]
        viol = np.maximum(mi, me) > 1e-3
        rows.append([n, viol.mean(), mi.mean(), me.mean(), me.max(), rw.mean(), rw[n // 2:].mean()])
        del tr
    got = np.array(rows)

    def se(col):
        return float(np.sqrt(ref[:, col].var(ddof=1) / len(ref) + got[:, col].var(ddof=1) / len(got)))
    out = {"ref_seeds": len(ref), "gpu_seeds": len(got), "steps": steps, "ref_mean": ref.mean(0).tolist(),
           "gpu_mean": got.mean(0).tolist(), "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist(),
           "se_of_difference": [se(c) for c in range(ref.shape[1])], "columns": [str(c) for c in g["columns"]],
           "bound": "2 SE + 10 % of the reference mean"}
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/statistical_parity_%s_evopf.json" % algo, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))
    assert got[:, 0].min() == steps                              # 40 complete days, every step logged
    for col in (1, 2, 5, 6):                                     # violation rate, mean max_ineq, return, return (2nd half)
        d = abs(got[:, col].mean() - ref[:, col].mean())
        assert d <= 2 * se(col) + 0.10 * abs(ref[:, col].mean()), (g["columns"][col], d, se(col), ref[:, col].mean())
    # equality drift no worse than the reference's -- on all but the rare step where Newton does not converge: SAC's exploration
    # can propose generator voltages at the lower edge of the box with almost no generation, for which the power flow has no
    # solution near the flat start (round 4: 1 step in 46 080, max_eq 117; the float64 oracle with LAPACK pivoting does not
    # converge on that input either: tests/golden/evopf_newton_divergence.npz, test_evopf_oracle.py).  The reference's 24 runs
    # contain no such step (23 040 steps), which does not separate the two rates.
    drift = np.sort(got[:, 4])
    assert drift[int(0.95 * len(drift)) - 1] <= max(2.0 * ref[:, 4].max(), 1e-2), drift[-5:]
    assert (got[:, 4] > 0.1).sum() <= 2, drift[-5:]


@pytest.mark.timeout(1500, method="thread")
def test_vectorised_cadences_learn_like_the_reference(golden):
    """Learning quality at the BENCHMARKED cadences: cart-RPODDPG at num_envs = 4096 with (a) one batch-256
    update per vector step (the headline's cadence) and (b) one batch-2^20 update per vector step (`large_batch`), 8 seeds
    each, to the budget of UPDATES of the reference's runs (3000; tests/golden/training_stats_ddpg_cart.npz, 384 seeds).  The
    reference's per-run statistics are computed per lane from the replay ring exactly as its Logger records them
    (tools/cadence_learning.py) and averaged over the lanes of a run; a run is one sample.

    (a) must reproduce the reference at matched updates: return (whole run, second half) within 2 SE + 10 %, violation rate
    within 2 SE + 1e-3 (measured round 4: 25.6 +- 2.1 / 34.1 +- 3.5 vs 27.5 / 32.9; 1.07e-2 +- 0.18e-2 vs 1.31e-2).
    (b) is ANOTHER optimiser regime (4096 x the batch, same learning rates and clip): measured 24.1 +- 1.8 / 31.4 +- 3.0 and
    0.76e-2 +- 0.12e-2 -- fewer violations, ~15 % less return at matched updates -- so it is held to: violation rate not above
    the reference's (2 SE), return not more than 2 SE + 25 % below; bench.py labels its figure accordingly."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import cadence_learning as cl
    g = golden("training_stats_ddpg_cart")
    ref, steps = g["stats"], int(g["steps"])
    dev = torch.device("cuda")
    res = {"reference": cl.reference_row(), "modes": []}
    for mode in ("reference_cadence", "large_batch"):
        m = cl.run_mode(mode, 8, steps, dev)
        res["modes"].append(m)

        def se(col, key):
            return float(np.sqrt(ref[:, col].var(ddof=1) / len(ref) + m["se"][key] ** 2))
        d_viol = m["mean"]["viol_rate"] - ref[:, 1].mean()
        d_ret = m["mean"]["mean_return_per_step"] - ref[:, 4].mean()
        d_ret2 = m["mean"]["mean_return_second_half"] - ref[:, 5].mean()
        assert abs(m["mean"]["device_viol_rate"] - m["mean"]["viol_rate"]) < 1e-3      # device counter == per-lane replay of the ring
        assert m["mean"]["logged_steps"] > 0.95 * steps
        if mode == "reference_cadence":
            assert abs(d_viol) <= 2 * se(1, "viol_rate") + 1e-3, (d_viol, se(1, "viol_rate"))
            assert abs(d_ret) <= 2 * se(4, "mean_return_per_step") + 0.10 * ref[:, 4].mean(), d_ret
            assert abs(d_ret2) <= 2 * se(5, "mean_return_second_half") + 0.10 * ref[:, 5].mean(), d_ret2
        else:
            assert d_viol <= 2 * se(1, "viol_rate"), (d_viol, se(1, "viol_rate"))
            assert d_ret >= -(2 * se(4, "mean_return_per_step") + 0.25 * ref[:, 4].mean()), d_ret
            assert d_ret2 >= -(2 * se(5, "mean_return_second_half") + 0.25 * ref[:, 5].mean()), d_ret2
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/cadence_learning.json", "w") as f:
        json.dump(res, f, indent=1)
