"""The statistical side of parity (north_star: returns, constraint-violation rate within 1e-3 of the reference), asserted on
COMMITTED evidence -- no GPU needed, deterministic.

Since round 5 the heavy runs live outside the `-m gpu` suite: `tools/statistical_parity.py` and
`tools/cadence_learning.py` collect them on the GPU box, their per-seed ROWS are committed under profiles/ and compared here
with the reference's own runs (tests/golden/training_stats_*.npz, recorded from the unmodified reference by
tests/golden/make_golden.py stats).  The in-suite GPU leg (tests/test_statistical_parity_gpu.py) re-runs a small fresh sample on
every GPU run.

Columns of a row: logged_steps, viol_rate, mean_max_ineq, mean_max_eq, mean_return_per_step, mean_return_second_half, max_nu;
violation rate = fraction of env steps with max(max_ineq, max_eq) > 1e-3.
"""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def _provenance(stamp, what):
    """committed rows say which kernels produced them (tools/provenance.py).  Rows of another ABI version are STALE:
    fail (regenerate: tools/collect_statistics.sh).  Rows of this ABI whose kernel-source hash differs from the tree's were
    measured before a later kernel edit: the numbers below are still asserted, with a warning that names the regeneration job."""
    import sys
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import provenance
    if stamp.get("abi") is None:
        return                                                  # (round-5 files carry no stamp; they were produced at ABI 5)
    assert int(stamp["abi"]) == provenance.abi_version(), "%s was produced at ABI %s, the tree is ABI %d: regenerate with " \
        "tools/collect_statistics.sh" % (what, stamp["abi"], provenance.abi_version())
    if str(stamp.get("csrc_sha16")) != provenance.csrc_sha16():
        warnings.warn("%s was produced by kernel sources %s, the tree is %s: regenerate with tools/collect_statistics.sh before "
                      "quoting it for this tree" % (what, stamp.get("csrc_sha16"), provenance.csrc_sha16()))


def _rows(algo, envname):
    """The newest committed per-seed rows of (algo, env): round 6's (stamped with ABI + source hash) where they exist."""
    for r in ("r06", "r05"):
        path = os.path.join(PROFILES, "%s_stat_rows_%s_%s.npz" % (r, algo, envname))
        if os.path.exists(path):
            z = np.load(path)
            _provenance({"abi": int(z["abi"]) if "abi" in z.files else None,
                         "csrc_sha16": str(z["csrc_sha16"]) if "csrc_sha16" in z.files else None}, os.path.basename(path))
            return z["stats"]
    raise AssertionError("no committed rows for %s %s" % (algo, envname))


def _cadence(name):
    for r in ("r06", "r05"):
        path = os.path.join(PROFILES, "%s_%s.json" % (r, name))
        if os.path.exists(path):
            with open(path) as f:
                res = json.load(f)
            _provenance(res, os.path.basename(path))
            return res
    raise AssertionError("no committed " + name)


def _compare(algo, envname):
    ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_%s_%s.npz" % (algo, envname)))["stats"]
    got = _rows(algo, envname)
    se = np.sqrt(ref.var(0, ddof=1) / len(ref) + got.var(0, ddof=1) / len(got))
    d = got.mean(0) - ref.mean(0)
    return ref, got, d, se


def test_cart_ddpg_violation_rate_at_1536_plus_1536_seeds():
    """The headline's algorithm (scripts/cart_exp.py, the only script with a shared embedding).  Rounds 3-4 saw a
    same-sign offset of +0.7e-3 in two GPU samples against the reference's first 384 seeds at a resolution (SE 4.8e-4) that
    could not tell it from zero.  With 1536 reference runs (tests/golden/make_golden.py stats, ~9 core-hours) and 1536 GPU
    runs the standard error of the difference is 2.7e-4 and the offset is gone: +0.12e-3 (z = 0.45); the reference's first 384
    seeds were a low draw of its own distribution (return 27.5 there, 28.9 over all 1536).  No column differs at 2 sigma."""
    ref, got, d, se = _compare("ddpg", "cart")
    assert len(ref) >= 1536 and len(got) >= 1536
    assert se[1] <= 3.0e-4, se[1]                                # resolves the north_star's 1e-3 at 3.5 sigma, 0.5e-3 at 1.8
    assert abs(d[1]) <= 0.5e-3, (d[1], se[1])                    # "either |delta| < 0.5e-3 or a documented root cause"
    z = d / np.maximum(se, 1e-300)
    assert np.abs(z[[0, 1, 2, 4, 5, 6]]).max() < 2.5, z          # logged steps, violations, returns, largest multiplier
    assert got[:, 3].max() < 1e-5 and ref[:, 3].max() < 1e-5     # the equality holds to float32 round-off on both sides
    assert got[:, 0].min() > 0.9 * 3000


@pytest.mark.parametrize("algo,envname,se_max", [("sac", "pendulum", 1.0e-4), ("sac", "cart", 4.0e-4), ("ddpg", "pendulum", 6.0e-5)])
def test_other_workloads_match_the_reference(algo, envname, se_max):
    """pendulum-RPOSAC (config 3: 192 reference seeds), cart-RPOSAC (config 4's algorithm: 1152 since round 6, 384 before: SE of
    the difference 5.9e-4 -> 3.5e-4), pendulum-RPODDPG (576), one or
    two GPU runs per reference run: |delta violation rate| <= 1e-3 + 2 SE at a resolution that can see 1e-3, per-step violation
    within 15 % + 2 SE, returns within 5 % + 2 SE."""
    ref, got, d, se = _compare(algo, envname)
    assert len(got) >= len(ref)
    assert se[1] <= se_max, se[1]
    assert abs(d[1]) <= 1e-3 + 2 * se[1], (d[1], se[1])
    assert abs(d[2]) <= 0.15 * ref[:, 2].mean() + 2 * se[2] + 1e-5, (d[2], se[2])
    for col in (4, 5):
        assert abs(d[col]) <= 0.05 * ref[:, col].mean() + 2 * se[col], (col, d[col], se[col])
    assert got[:, 3].max() < 1e-4


def test_vectorised_cadences_at_128_and_32_seeds():
    """cart-RPODDPG at 4096 lanes to the reference's budget of UPDATES (3000): (a) one batch-256 update per vector step (the
    headline's cadence), 128 seeds; (b) one batch-2^20 update per vector step (`large_batch`), 32 seeds
    (profiles/r06_cadence_learning.json -- r05_* before the kernels of ABI 6 --, tools/cadence_learning.py).  Round 4's 8 + 8 seeds (SE 1.8e-3) could not see 1e-3.

    What 128 seeds show: the vectorised cadence does NOT reproduce the reference's numbers at matched updates -- it is BETTER on
    both: violation rate 1.10e-2 +- 0.06e-2 vs 1.33e-2 +- 0.02e-2 (-2.2e-3, z = -3.5), return 33.4 +- 1.6 vs 28.9 +- 0.35.  The lane
    sweep below shows where that comes from.  Asserted: the resolution, and that the vectorised cadences are not WORSE than the
    reference (violations, 2 SE) and not below it in return (2 SE + 10 %)."""
    res = _cadence("cadence_learning")
    ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_ddpg_cart.npz"))["stats"]
    modes = {m["mode"]: m for m in res["modes"]}
    a, b = modes["reference_cadence"], modes["large_batch"]
    assert a["seeds"] >= 128 and b["seeds"] >= 32 and a["lanes"] == 4096 and a["updates"] == 3000

    def se(m, key, col):
        return float(np.sqrt(ref[:, col].var(ddof=1) / len(ref) + m["se"][key] ** 2))
    assert a["se"]["viol_rate"] <= 8e-4 and b["se"]["viol_rate"] <= 1.2e-3          # (round 4: 1.8e-3 / 1.2e-3 on 8 seeds)
    for m, slack in ((a, 0.10), (b, 0.10)):       # (the large batch needed 25 % until the memset-node bug of DESIGN 4.5 was fixed)
        assert m["mean"]["viol_rate"] - ref[:, 1].mean() <= 2 * se(m, "viol_rate", 1)
        assert abs(m["mean"]["device_viol_rate"] - m["mean"]["viol_rate"]) < 1e-3
        for key, col in (("mean_return_per_step", 4), ("mean_return_second_half", 5)):
            assert m["mean"][key] - ref[:, col].mean() >= -(2 * se(m, key, col) + slack * ref[:, col].mean()), (m["mode"], key)
    assert abs(a["mean"]["viol_rate"] - ref[:, 1].mean()) <= 3.5e-3                  # the measured shift is -2.2e-3 +- 0.6e-3


def test_lane_sweep_locates_the_shift_of_the_vectorised_cadence():
    """Reference cadence at 1, 16 and 256 lanes, 128 seeds each (profiles/r06_cadence_lanes.json): one lane reproduces the
    reference (every |z| < 2.5: it IS the reference's algorithm step for step); the shift of the statistics is complete at 16
    lanes and does not grow to 4096 -- it comes with the FIRST independent histories a batch can draw from (the reference fills
    its replay buffer with one correlated trajectory: for the first 256 steps a batch of 256 resamples fewer than 256 distinct
    transitions; 97 % of a run's violations fall into its first 250 steps), not with the kernels and not with the lane count."""
    res = _cadence("cadence_lanes")
    rows = {m["lanes"]: m for m in res["sweep"]}
    assert set(rows) >= {1, 16, 256}
    one = rows[1]["z_vs_reference"]
    assert max(abs(v) for v in one.values()) < 2.5, one
    v16, v256 = rows[16]["mean"]["viol_rate"], rows[256]["mean"]["viol_rate"]
    assert v16 < rows[1]["mean"]["viol_rate"] - 1e-3 and abs(v16 - v256) < 1.5e-3
    assert rows[16]["mean"]["mean_return_per_step"] > rows[1]["mean"]["mean_return_per_step"] + 2.0


def test_pooled_independent_samples_of_the_headline_violation_rate():
    """VERDICT r05 next 5.  bench.py's num_envs = 1 figure of rounds 3-5 re-ran seeds 0..383 every time -- the first quarter of the
    committed 1536-seed rows, so the driver's lines added no information (identical 0.013766... in r04 and r05).  Since round 6
    every bench.py run draws 384 FRESH seeds (base printed in the line), `tools/append_n1_sample.py` records each line in
    profiles/history/n1_violation_samples.json, and this test pools all independent GPU samples against the reference's 1536
    runs: the pooled difference must sit inside north_star's 1e-3 with two standard errors to spare, at a pooled SE <= 2.8e-4."""
    with open(os.path.join(PROFILES, "history", "n1_violation_samples.json")) as f:
        samples = json.load(f)["samples"]
    spans = sorted((s["seed_base"], s["seed_base"] + s["seeds"]) for s in samples)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), "overlapping seed ranges: not independent samples"
    n = np.array([s["seeds"] for s in samples], dtype=np.float64)
    mean = float((n * np.array([s["mean"] for s in samples])).sum() / n.sum())
    # pooled standard error of the seed-weighted mean from the samples' own standard errors
    se_gpu = float(np.sqrt(((n * np.array([s["se"] for s in samples])) ** 2).sum()) / n.sum())
    ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_ddpg_cart.npz"))["stats"][:, 1]
    se = float(np.sqrt(se_gpu ** 2 + ref.var(ddof=1) / len(ref)))
    d = mean - float(ref.mean())
    assert n.sum() >= 1536 and len(ref) >= 1536
    assert se <= 2.8e-4, se
    assert abs(d) <= 1e-3 - 2 * se, (d, se)
    for s in samples:                                           # no single sample is an outlier of the pool (3 of its own SE + the reference's)
        assert abs(s["mean"] - float(ref.mean())) <= 3 * np.sqrt(s["se"] ** 2 + ref.var(ddof=1) / len(ref)) + 0.5e-3, s
