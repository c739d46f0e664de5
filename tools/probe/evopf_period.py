"""EVOPF windows: how long are a critic-only iteration (Y) and a policy iteration (X)?  Times the bench workload at
policy_fre = 4 (the scripts' value) and 16 and solves  4 avg4 = 3 Y + X,  16 avg16 = 15 Y + X;  with and without the
second captured branch (RPO_SCHEDULE=branch=0: the serial chain).
    python tools/probe/evopf_period.py [evopf_ddpg|evopf_sac] [iterations]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "evopf_ddpg"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
dev = torch.device("cuda")
bench.spin_up(dev)


def measure(fre, branch):
    os.environ["RPO_SCHEDULE"] = "branch=%d" % branch
    tr = bench.make_trainer(1024, dev, 10 ** 9, workload=workload, policy_fre=fre)
    tr.vec.reset()
    tr.run_steps(6 * 16)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        tr.run_steps(iters)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e6)
    return best


for branch in (1, 0):
    a4, a16 = measure(4, branch), measure(16, branch)
    y = (16 * a16 - 4 * a4) / 12.0
    x = 4 * a4 - 3 * y
    print("%s branch=%d: %.1f us per iteration at policy_fre 4 (%.2f M env-steps/s), %.1f at 16  ->  critic-only iteration "
          "%.1f us, policy iteration %.1f us" % (workload, branch, a4, 1024 / a4, a16, y, x), flush=True)
