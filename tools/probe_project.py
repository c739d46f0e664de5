#!/usr/bin/env python
"""Duration of the batch-coupled SpringPendulum projection (rpo_pendulum_project_batchref) by GRG iteration count:
a hipGraph of 100 back-to-back launches, timed with events.  python tools/probe_project.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from rpo_amd import ops
    g = np.load(os.path.join(ROOT, "tests", "golden", "pendulum_grad_steps.npz"))
    k = ops.PendulumKernels()
    for n in (256, 1024):
        idx = np.arange(n) % 256
        obs = torch.as_tensor(g["obs32"][idx], dtype=torch.float32).cuda()
        ap = torch.as_tensor(g["ap"].reshape(-1)[idx], dtype=torch.float32).cuda()
        action = torch.zeros(n, 2, device="cuda")
        iters = torch.zeros(1, dtype=torch.int32, device="cuda")
        for steps in (0, 1, 2, 5, 10):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(3):
                    k.project_batchref(obs, ap, action, iters, steps, 2e-3, 1e-5, 0.0)
                s.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    for _ in range(100):
                        k.project_batchref(obs, ap, action, iters, steps, 2e-3, 1e-5, 0.0)
                gr.replay()
                s.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(5):
                    gr.replay()
                e1.record(s)
                s.synchronize()
            print("n=%d max_steps=%d iters=%d: %.2f us per launch" % (n, steps, int(iters.item()), e0.elapsed_time(e1) * 1e3 / 500),
                  flush=True)
            if n > 256:
                continue
            want = action.clone()
            for mode in (0, 1, 2):
                ws = torch.zeros(5408, dtype=torch.int64, device="cuda")
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    for _ in range(3):
                        k.project_batchref_ws(obs, ap, action, iters, steps, 2e-3, 1e-5, 0.0, ws, mode)
                    s.synchronize()
                    same = bool(torch.equal(action, want))
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=s):
                        for _ in range(100):
                            k.project_batchref_ws(obs, ap, action, iters, steps, 2e-3, 1e-5, 0.0, ws, mode)
                    gr.replay()
                    s.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                    for _ in range(5):
                        gr.replay()
                    e1.record(s)
                    s.synchronize()
                print("   8 workgroups, store mode %d: %.2f us per launch, iters=%d, bits equal: %s, gave up: %d, epoch %d" % (
                    mode, e0.elapsed_time(e1) * 1e3 / 500, int(iters.item()), same and bool(torch.equal(action, want)),
                    int(ws[529]), int(ws[528])), flush=True)


if __name__ == "__main__":
    main()
