#!/usr/bin/env python
"""Seed statistics of the shipped trainers at num_envs = 1 on the HIP kernels, in the protocol of
tests/test_statistical_parity_gpu.py (one torch seed AND one Philox seed per run, `steps` iterations of the script's
hyper-parameters, the reference's Logger rows) -- the heavy side of the statistical parity claim, run OUTSIDE the `-m gpu`
suite so that the suite stays short:

    python tools/statistical_parity.py ddpg cart 1536            # -> gpurun_out/stat_rows_ddpg_cart.npz + .json

The rows have the columns of tests/golden/training_stats_<algo>_<env>.npz (tests/golden/make_golden.py _stats_run):
logged_steps, viol_rate, mean_max_ineq, mean_max_eq, mean_return_per_step, mean_return_second_half, max_nu.  The JSON next to
them holds the comparison with the reference rows that file has at the time of the run (means, standard errors of the
difference, z-scores); tools/compare_stats.py recomputes it on the CPU from the committed rows (no GPU needed).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import bench  # noqa: E402

WORKLOAD = {("ddpg", "cart"): "cart_ddpg", ("sac", "cart"): "cart_sac", ("ddpg", "pendulum"): "pen_ddpg",
            ("sac", "pendulum"): "pen_sac"}
COLUMNS = ["logged_steps", "viol_rate", "mean_max_ineq", "mean_max_eq", "mean_return_per_step", "mean_return_second_half",
           "max_nu"]


def gpu_rows(algo, envname, seeds, steps=3000, first=0, device=None):
    from rpo_amd.utils.logger import Logger
    device = device or torch.device("cuda")
    rows = []
    for seed in range(first, first + seeds):
        tr = bench.make_trainer(1, device, steps, capacity=steps, workload=WORKLOAD[(algo, envname)],
                                torch_seed=123 + seed, seed=5000 + seed)
        tr.logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=steps)
        tr.run(eval=False)
        n = tr.logger.pointer
        mi, me, rw = [tr.logger.tracker[k][:n] for k in ("max_ineq", "max_eq", "reward")]
        viol = np.maximum(mi, me) > 1e-3
        rows.append([n, viol.mean(), mi.mean(), me.mean(), rw.mean(), rw[n // 2:].mean(),
                     float(tr.agent.nju.weight.detach().abs().max())])
        assert me.max() < 1e-4, (seed, me.max())                  # the equality holds on every step (equation solver)
        del tr
    return np.array(rows)


def compare(ref, got):
    """Means, standard errors of the difference and z-scores, column by column."""
    out = {"ref_seeds": int(len(ref)), "gpu_seeds": int(len(got)), "columns": COLUMNS,
           "ref_mean": ref.mean(0).tolist(), "gpu_mean": got.mean(0).tolist(),
           "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist()}
    se = np.sqrt(ref.var(0, ddof=1) / len(ref) + got.var(0, ddof=1) / len(got))
    d = got.mean(0) - ref.mean(0)
    out["gpu_minus_ref"] = d.tolist()
    out["se_of_difference"] = se.tolist()
    out["z"] = [float(x / s) if s > 0 else 0.0 for x, s in zip(d, se)]
    return out


def main():
    algo, envname = sys.argv[1], sys.argv[2]
    seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 384
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
    t0 = time.time()
    got = gpu_rows(algo, envname, seeds, steps)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import provenance
    st = provenance.stamp()
    np.savez_compressed(os.path.join(out_dir, "stat_rows_%s_%s.npz" % (algo, envname)), stats=got, steps=steps,
                        columns=np.array(COLUMNS), torch_seed_base=123, philox_seed_base=5000, abi=st["abi"],
                        csrc_sha16=np.array(st["csrc_sha16"]))
    ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_%s_%s.npz" % (algo, envname)))["stats"]
    res = compare(ref, got)
    res.update(steps=steps, seconds=time.time() - t0, **st)
    with open(os.path.join(out_dir, "stat_rows_%s_%s.json" % (algo, envname)), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({k: res[k] for k in ("ref_seeds", "gpu_seeds", "gpu_minus_ref", "se_of_difference", "z", "seconds")}))


if __name__ == "__main__":
    main()
