#!/usr/bin/env python
"""One update on a LARGE batch per vector step (256 sampled transitions consumed per env step, here as one
batch of 256 * N instead of N sequential batch-256 updates): time per iteration and per launch.

    python tools/probe_large_batch.py [workload] [batch] [lanes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402


def main(workload="cart_ddpg", batch=1 << 20, lanes=4096):
    import bench
    batch, lanes = int(batch), int(lanes)
    tr = bench.make_trainer(lanes, torch.device("cuda"), 10 ** 9, capacity=64, workload=workload, batch_size=batch, use_graph=False)
    tr.vec.reset()
    tr.run_steps(3 * max(tr._cycle, 4))                       # (past the first actor update: its buffers are allocated)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2 * max(tr._cycle, 4)
    tr.run_steps(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s, batch %d, %d lanes: %.3f ms per iteration = %.3f M env-steps/s, %.1f M sampled transitions/s" % (
        workload, batch, lanes, dt * 1e3, lanes / dt * 1e-6, batch / dt * 1e-6), flush=True)
    clinic = bench.kernel_clinic(tr, workload)
    for k, v in sorted(clinic.items(), key=lambda kv: -kv[1]["us"] * kv[1].get("launches_per_period", 1)):
        print("  %-40s %10.1f us x%d" % (k, v["us"], v.get("launches_per_period", 0)) +
              ("   %8.2f %s (%.3f of peak)" % (v["rate"], v["unit"], v["frac"]) if "rate" in v else ""))


if __name__ == "__main__":
    main(*sys.argv[1:])
