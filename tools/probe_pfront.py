#!/usr/bin/env python
"""Phase timestamps of the SpringPendulum fused front (rpo_split_critic_pfront), from a library built with -DRPO_PM_TRACE
(HIPCC_EXTRA=-DRPO_PM_TRACE python rpo_amd/csrc/build.py --force): 100 MHz stamps behind the projection workspace.

    python tools/probe_pfront.py [workload]      (development aid; the shipped library has no stamps)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

NAMES = {0: "proj start", 1: "proj rows gathered", 2: "proj pi granules in", 3: "proj head done", 4: "proj xcc known", 36: "proj loop done",
         40: "pi start", 41: "pi slab done", 42: "pi granules out", 44: "q start", 45: "q arrived", 48: "fwd_b start",
         49: "fwd_b pre-wait", 50: "fwd_b actions in", 51: "fwd_b arrived", 52: "bwd_a start", 53: "bwd_a end"}


def main(workload="pen_sac"):
    from bench import make_trainer
    from rpo_amd import ops
    ops.PROJ_WS_WORDS += 64                                      # (room for the stamps behind the projection workspace)
    tr = make_trainer(4096, torch.device("cuda"), 10 ** 9, capacity=64, workload=workload, use_graph=False)
    tr.vec.reset()
    tr.run_steps(400)
    su = tr._split_state()
    su.set(proj_iters=torch.zeros(1, dtype=torch.int32, device="cuda"))
    ws = su._held["proj_ws"]
    base = 5408
    acc = {}
    for it in range(200):
        tr.run_steps(1)
        torch.cuda.synchronize()
        st = ws[base:base + 64].cpu().numpy().astype(np.int64)
        t0 = min(int(st[40]), int(st[0]), int(st[44]))
        iters = int(su._held["proj_iters"].item()) if "proj_iters" in su._held else -1
        for k, name in NAMES.items():
            acc.setdefault(name, []).append((int(st[k]) - t0) * 10.0)
        for k in range(iters if iters > 0 else 0):
            acc.setdefault("proj iter %d" % k, []).append((int(st[5 + k]) - t0) * 10.0)
    for name, v in sorted(acc.items(), key=lambda kv: np.median(kv[1])):
        print("%-24s median %8.0f ns   (n=%d)" % (name, np.median(v), len(v)))


if __name__ == "__main__":
    main(*sys.argv[1:])
