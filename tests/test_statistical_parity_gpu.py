"""Statistical parity with the reference's CPU runs (north_star: returns and constraint-violation rate) -- the IN-SUITE leg.

tests/golden/training_stats_*.npz hold the statistics of 3000-iteration training runs of the unmodified reference
(scripts/cart_exp.py: 1536 seeds (round 5; 384 before); scripts/pen_exp_sac.py: 192; scripts/cart_exp_sac.py: 384;
scripts/pen_exp.py: 576; tests/golden/make_golden.py stats).  The same runs are repeated with the shipped trainers at
num_envs = 1 -- the reference's cadence, step for step -- on the HIP kernels.  Random streams differ (Philox vs numpy/torch
global generators) and trajectories are chaotic, so the comparison is between seed-averaged statistics: violation rate =
fraction of env steps with max(max_ineq, max_eq) > 1e-3.

Since round 5 the comparison lives in TWO places (the suite took 470 s, 415 of them here, 1017 s on a
slow box against the driver's 1200 s):
  * the full-resolution claim -- as many GPU seeds as reference seeds or more, standard error of the difference down to
    2.7e-4 for cart-RPODDPG -- is collected by `tools/statistical_parity.py` / `tools/cadence_learning.py` on the GPU box,
    its ROWS are committed under profiles/ (r05_stat_rows_*.npz, r05_cadence_learning.json) and asserted by
    tests/test_statistical_evidence.py on the CPU (deterministic, no GPU);
  * here, on every GPU run: GPU_SEEDS fresh seeds per case -- enough to catch a broken kernel or host loop (a 70 % gap in the
    per-step violation, a violation rate off by 2e-3, an equality residual above float32 round-off), not to resolve 1e-3.
Every run has its own initial weights (torch seed) AND its own Philox seed (exploration noise, reset states, replay indices,
update noise): with one shared Philox seed the runs share one noise realisation, and its luck does not average out.
"""
import json
import os

import numpy as np
import pytest
import torch

from test_train_step_golden import build_trainer

pytestmark = pytest.mark.gpu


GPU_SEEDS = 128          # per case, ~0.13 s each (host-bound at num_envs = 1)


@pytest.mark.timeout(900, method="thread")
# (algo, env, largest standard error of the violation-rate difference the in-suite sample must reach: the seed-to-seed spread
#  is 7.8e-3 for cart-RPODDPG, 7.4e-3 for cart-RPOSAC, ~1e-3 on SpringPendulum)
@pytest.mark.parametrize("algo,envname,se_max", [("ddpg", "cart", 9e-4), ("sac", "pendulum", 5e-4),
                                                 ("sac", "cart", 9e-4), ("ddpg", "pendulum", 5e-4)])
def test_training_statistics_match_reference(golden, algo, envname, se_max):
    from rpo_amd import ops
    from rpo_amd.utils.logger import Logger
    g = golden("training_stats_%s_%s" % (algo, envname))
    ref, steps = g["stats"], int(g["steps"])
    n_gpu = GPU_SEEDS
    os.environ["RPO_VERBOSE"] = "0"
    rows = []
    for seed in range(100000, 100000 + n_gpu):                  # (seeds the committed evidence rows did not use)
        torch.manual_seed(123 + seed)
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
    assert se(1) <= se_max, se(1)
    assert d_viol <= 1e-3 + 2.5 * se(1), (d_viol, se(1))          # (resolution of THIS sample; the 1e-3 claim: test_statistical_evidence.py)
    d_ineq = abs(got[:, 2].mean() - ref[:, 2].mean())
    assert d_ineq <= 0.15 * ref[:, 2].mean() + 2.5 * se(2) + 1e-5, (d_ineq, se(2))
    for col in (4, 5):                                           # episodic return, whole run and second half
        d = abs(got[:, col].mean() - ref[:, col].mean())
        assert d <= 0.05 * ref[:, col].mean() + 2.5 * se(col), (col, d, se(col))
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
    rows, outliers, drift_converged = [], [], []
    from oracle import evopf as oe                               # (test infrastructure: the checker of the outlier steps)
    os.environ["RPO_VERBOSE"] = "0"
    # GPU seeds: twice the reference's while that pin was thin (24 reference seeds until round 5), as many as the reference's
    # since it has 96 (round 6, VERDICT r05 next 7: SE of the difference 0.25 -> 0.14 of the seed-to-seed spread)
    for seed in range(2 * len(ref) if len(ref) < 96 else len(ref)):
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
        # every step that stored an equality violation > 0.1 is replayed through the float64 oracle: the stored
        # basic action completed from the flat start, numpy LAPACK solves with partial pivoting
        c = tr.kernels.cols
        ring = tr.buffer.rows[:n].cpu().numpy().astype(np.float64)       # (num_envs = 1: ring row = step)
        eqv = np.abs(ring[:, c["eq_viol"][0]:c["eq_viol"][1]]).max(1)
        ok_steps = np.ones(n, dtype=bool)
        for tb in np.nonzero(eqv > 0.1)[0]:
            st = ring[tb:tb + 1, c["state"][0]:c["state"][1]]
            ap = ring[tb:tb + 1, c["action"][0]:c["action"][1]][:, oe.GRID.partial_actions]
            a64, _, _, its = oe.complete_partial(st, ap, return_aux=True)
            diverges = int(np.asarray(its).max()) >= 50 or float(np.abs(oe.eq_resid(st, a64)).max()) > 0.1
            outliers.append((seed, int(tb), float(eqv[tb]), bool(diverges)))
            ok_steps[tb] = not diverges
        drift_converged.append(float(eqv[ok_steps].max()))
        del tr
    got = np.array(rows)

    def se(col):
        return float(np.sqrt(ref[:, col].var(ddof=1) / len(ref) + got[:, col].var(ddof=1) / len(got)))
    out = {"ref_seeds": len(ref), "gpu_seeds": len(got), "steps": steps, "ref_mean": ref.mean(0).tolist(),
           "gpu_mean": got.mean(0).tolist(), "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist(),
           "se_of_difference": [se(c) for c in range(ref.shape[1])], "columns": [str(c) for c in g["columns"]],
           "bound": "2 SE + 10 % of the reference mean",
           "steps_with_eq_violation_above_0.1": [dict(seed=a, step=b, max_eq=c, oracle_diverges_too=d) for a, b, c, d in outliers]}
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
    # Since round 5 no blanket allowance: EVERY step with a stored violation > 0.1 is replayed through the float64
    # oracle above and must fail to converge there too (a property of the input, not of the float32 static-order solver); over
    # all other steps the strict bound of round 3 holds again: max drift <= 2 x the reference's max.
    assert all(d for _, _, _, d in outliers), outliers
    assert len(outliers) <= 4, outliers                          # (r04: 1 step in 46 080)
    assert max(drift_converged) <= max(2.0 * ref[:, 4].max(), 1e-2), sorted(drift_converged)[-5:]


@pytest.mark.timeout(900, method="thread")
def test_vectorised_cadences_learn_like_the_reference(golden):
    """Learning at the BENCHMARKED cadences, in-suite leg: cart-RPODDPG at num_envs = 4096 with (a) one batch-256 update per
    vector step (the headline's cadence), 24 seeds, and (b) one batch-2^20 update per vector step (`large_batch`), ONE seed, to
    the budget of UPDATES of the reference's runs (3000).  The reference's per-run statistics are computed per lane from the
    replay ring exactly as its Logger records them (tools/cadence_learning.py) and averaged over the lanes of a run.

    The claim with resolution -- 128 + 32 seeds, standard error 6e-4 on the violation rate -- is profiles/r05_cadence_learning.json,
    asserted on the CPU by tests/test_statistical_evidence.py; here: the device-side violation counter equals the per-lane replay
    of the ring, almost every step is logged, and the statistics sit inside wide sanity bands around the committed ones (a
    broken update shows as a violation rate of 3e-2+ or a return below 15)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import cadence_learning as cl
    g = golden("training_stats_ddpg_cart")
    ref, steps = g["stats"], int(g["steps"])
    dev = torch.device("cuda")
    res = {"reference": cl.reference_row(), "modes": []}
    for mode, seeds in (("reference_cadence", 24), ("large_batch", 1)):
        m = cl.run_mode(mode, seeds, steps, dev)
        res["modes"].append(m)
        assert abs(m["mean"]["device_viol_rate"] - m["mean"]["viol_rate"]) < 1e-3      # device counter == per-lane replay of the ring
        assert m["mean"]["logged_steps"] > 0.95 * steps
        if mode == "reference_cadence":
            se_v = float(np.sqrt(ref[:, 1].var(ddof=1) / len(ref) + m["se"]["viol_rate"] ** 2))
            se_r = float(np.sqrt(ref[:, 4].var(ddof=1) / len(ref) + m["se"]["mean_return_per_step"] ** 2))
            # anchored on the REFERENCE's golden statistics (round 5 had left only the bands around this repo's own
            # earlier measurement): the headline's cadence must not violate more than the reference (2 SE + 1e-3) nor return less
            # (2 SE + 10 %) at matched updates -- the bounds tests/test_statistical_evidence.py holds the 128-seed rows to
            assert m["mean"]["viol_rate"] - ref[:, 1].mean() <= 2 * se_v + 1e-3, (m["mean"]["viol_rate"], ref[:, 1].mean(), se_v)
            assert m["mean"]["mean_return_per_step"] - ref[:, 4].mean() >= -(2 * se_r + 0.10 * ref[:, 4].mean()), (m["mean"], se_r)
            # ... and the bands around the committed measurement, which see a kernel regression the one-sided bounds would pass
            assert abs(m["mean"]["viol_rate"] - 1.10e-2) <= 3 * se_v + 1e-3, (m["mean"]["viol_rate"], se_v)   # (r05: 1.10e-2 +- 0.06e-2)
            assert 25.0 < m["mean"]["mean_return_per_step"] < 42.0, m["mean"]                               # (r05: 33.4 +- 1.6)
        else:
            assert m["mean"]["viol_rate"] < 2e-2 and m["mean"]["mean_return_per_step"] > 12.0, m["mean"]     # (r05: 0.66e-2, 26.6)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/cadence_learning_suite.json", "w") as f:
        json.dump(res, f, indent=1)
