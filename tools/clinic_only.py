#!/usr/bin/env python
"""Kernel clinic alone (GPU box): per-launch times of one policy_fre period of a workload, nothing else."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "cart_ddpg"
os.environ.setdefault("RPO_VERBOSE", "0")
dev = torch.device("cuda", 0)
tr = bench.make_trainer(bench.envs_per_gpu(w), dev, 10 ** 9, capacity=64, workload=w)
tr.vec.reset()
tr.run_steps(64)
out = bench.kernel_clinic(tr, "none" if w == "cart_ddpg" else w)
