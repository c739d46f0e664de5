"""A/B of one library kernel-variant switch (rpo_tuning) on a bench workload, alternating in separate processes:
    python tools/ab_tuning.py evopf_ddpg gemm_produce 0 1 [rounds]
Prints env-steps/s and us per iteration of `bench.py --workload W --tuning KEY=VALUE` for each value."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload, key = sys.argv[1], sys.argv[2]
values = sys.argv[3:5]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 2
for r in range(rounds):
    for v in values:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-clinic", "--no-extras",
                              "--workload", workload, "--tuning", "%s=%s" % (key, v)], capture_output=True, text=True,
                             timeout=600).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print("%s %s=%s: %.4g env-steps/s, %.2f us per iteration" % (workload, key, v, d["value"], 1e3 * d["ms_per_step"]), flush=True)
