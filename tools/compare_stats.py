#!/usr/bin/env python
"""Recompute the comparison JSONs of the committed statistical evidence on the CPU (no GPU): for every
profiles/rNN_stat_rows_<algo>_<env>.npz (GPU rows, tools/statistical_parity.py) against tests/golden/training_stats_<algo>_<env>.npz
(the unmodified reference's rows, tests/golden/make_golden.py stats): means, standard errors of the difference, z-scores.

    python tools/compare_stats.py [rNN]      # (default r06) rewrites profiles/rNN_stat_rows_*.json -- keeping the rows' provenance
                                             # stamp --, prints one line per case
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
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_stat_rows_*.npz" % rnd))):
        case = os.path.basename(path)[len("%s_stat_rows_" % rnd):-4]
        z = np.load(path)
        got = z["stats"]
        ref = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_%s.npz" % case))["stats"]
        se = np.sqrt(ref.var(0, ddof=1) / len(ref) + got.var(0, ddof=1) / len(got))
        d = got.mean(0) - ref.mean(0)
        out = {"case": case, "columns": cols, "ref_seeds": int(len(ref)), "gpu_seeds": int(len(got)),
               "ref_mean": ref.mean(0).tolist(), "gpu_mean": got.mean(0).tolist(), "gpu_minus_ref": d.tolist(),
               "se_of_difference": se.tolist(), "z": [float(x / s) if s > 0 else 0.0 for x, s in zip(d, se)],
               "ref_std": ref.std(0, ddof=1).tolist(), "gpu_std": got.std(0, ddof=1).tolist()}
        for k in ("abi", "csrc_sha16"):                          # (tools/provenance.py: which kernels produced the rows)
            if k in z.files:
                out[k] = int(z[k]) if k == "abi" else str(z[k])
        if os.path.exists(path[:-4] + ".json"):                  # keep what the collecting run recorded beside the comparison
            with open(path[:-4] + ".json") as f:                 # (steps per run, seconds)
                out = dict(json.load(f), **out)
        with open(path[:-4] + ".json", "w") as f:
            json.dump(out, f, indent=1)
        print("%-14s ref %4d gpu %4d  viol %.5f vs %.5f  delta %+.2e +- %.2e (z %+.2f)  return %.2f vs %.2f (z %+.2f)  second half z %+.2f"
              % (case, len(ref), len(got), ref[:, 1].mean(), got[:, 1].mean(), d[1], se[1], d[1] / se[1], ref[:, 4].mean(),
                 got[:, 4].mean(), d[4] / se[4], d[5] / se[5]))


if __name__ == "__main__":
    main()
