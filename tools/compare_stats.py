#!/usr/bin/env python
"""Recompute the comparison JSONs of the committed statistical evidence on the CPU (no GPU): for every
profiles/r05_stat_rows_<algo>_<env>.npz (GPU rows, tools/statistical_parity.py) against tests/golden/training_stats_<algo>_<env>.npz
(the unmodified reference's rows, tests/golden/make_golden.py stats): means, standard errors of the difference, z-scores.

    python tools/compare_stats.py            # rewrites profiles/r05_stat_rows_*.json, prints one line per case
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cols = ["logged_steps", "viol_rate", "mean_max_ineq", "mean_max_eq", "mean_return_per_step", "mean_return_second_half", "max_nu"]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_stat_rows_*.npz"))):
        case = os.path.basename(path)[len("r05_stat_rows_"):-4]
        got = np.load(path)["stats"]
        ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_%s.npz" % case))["stats"]
        se = np.sqrt(ref.var(0, ddof=1) / len(ref) + got.var(0, ddof=1) / len(got))
        d = got.mean(0) - ref.mean(0)
        out = {"case": case, "columns": cols, "ref_seeds": int(len(ref)), "gpu_seeds": int(len(got)),
               "ref_mean": ref.mean(0).tolist(), "gpu_mean": got.mean(0).tolist(), "gpu_minus_ref": d.tolist(),
               "se_of_difference": se.tolist(), "z": [float(x / s) if s > 0 else 0.0 for x, s in zip(d, se)],
               "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist()}
        with open(path[:-4] + ".json", "w") as f:
            json.dump(out, f, indent=1)
        print("%-14s ref %4d gpu %4d  viol %.5f vs %.5f  delta %+.2e +- %.2e (z %+.2f)  return %.2f vs %.2f (z %+.2f)  second half z %+.2f"
              % (case, len(ref), len(got), ref[:, 1].mean(), got[:, 1].mean(), d[1], se[1], d[1] / se[1], ref[:, 4].mean(),
                 got[:, 4].mean(), d[4] / se[4], d[5] / se[5]))


if __name__ == "__main__":
    main()
