#!/usr/bin/env python
"""Append the num_envs = 1 violation-rate sample of a bench.py line to profiles/history/n1_violation_samples.json.

    python tools/append_n1_sample.py <bench line .json> [label]

Since round 6 every bench.py run draws its 384 seeds from a fresh base (`constraint_violation_rate_n1_seed_base`, printed in the
line), so every line is an INDEPENDENT sample of cart-RPODDPG's violation rate in the reference's setting;
tests/test_statistical_evidence.py pools the samples of this file against the reference's 1536 runs.  A line whose seed range
overlaps an entry already in the file is refused (it would be counted twice)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "profiles", "history", "n1_violation_samples.json")


def main():
    with open(sys.argv[1]) as f:
        text = f.read().strip().splitlines()
    line = json.loads([l for l in text if l.strip().startswith("{")][-1])
    if "line" in line and "value" not in line:
        line = line["line"]
    base, seeds = line.get("constraint_violation_rate_n1_seed_base"), line.get("constraint_violation_rate_n1_seeds")
    if base is None or seeds is None:
        sys.exit("this line has no independent n = 1 sample (written by a bench.py older than round 6, or --n1-seeds 0)")
    hist = json.load(open(PATH)) if os.path.exists(PATH) else {"what": __doc__.split("\n\n")[2], "samples": []}
    for s in hist["samples"]:
        if s["seed_base"] < base + seeds and base < s["seed_base"] + s["seeds"]:
            sys.exit("seed range [%d, %d) overlaps the sample %r" % (base, base + seeds, s["label"]))
    hist["samples"].append({"label": sys.argv[2] if len(sys.argv) > 2 else os.path.basename(sys.argv[1]), "seed_base": int(base),
                            "seeds": int(seeds), "mean": line["constraint_violation_rate_n1"],
                            "se": line["constraint_violation_rate_n1_se"], "bench_py_sha16": line.get("bench_py_sha16"),
                            "headline_env_steps_per_s": line.get("value")})
    with open(PATH, "w") as f:
        json.dump(hist, f, indent=1)
    print("%d samples, %d seeds" % (len(hist["samples"]), sum(s["seeds"] for s in hist["samples"])))


if __name__ == "__main__":
    main()
