#!/usr/bin/env python
"""The one-launch rollout at large lane counts, form by form (round 6): 64-lane row tiles (rollout_wide=1), the streaming form (3),
and what the size rule picks (2) -- event-timed like bench.py's clinic,
with the fraction of the f32 MFMA peak (67 584 flop per lane) and the env-steps/s of a rollout-only loop.

    python tools/probe/lanes_probe.py [cart_ddpg|cart_sac|pen_sac] [lanes ...]
RPO_HIP_LIBRARY selects another build (timing-only variants: -DRPO_RSTREAM_SKIP=1|2)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402
from rpo_amd import ops  # noqa: E402
from rpo_amd.env.vec import VecEnv  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "cart_ddpg"
sizes = [int(x) for x in sys.argv[1:] if x.isdigit()] or [65536, 1 << 20]
dev = torch.device("cuda")
bench.spin_up(dev, 1.0)
tr = bench.make_trainer(4096, dev, 10 ** 9, workload=workload, capacity=4)
k, f = tr.kernels, tr.fused
scale, base = tr._box_affine
fl = bench.mlp_flops(f.descs["actor"])
lib = os.path.basename(os.environ.get("RPO_HIP_LIBRARY", "librpo_hip.so"))
for n in sizes:
    big = VecEnv(k, n, dev, seed=3, stats_cap=64, max_episode_steps=200)
    big.reset()
    cap = max(2, min(8, (1 << 30) // (n * k.ring_floats * 4)))
    rows = torch.zeros(cap * n, k.ring_floats, device=dev)
    gauss = bool(tr._gauss_policy)
    noise = ops.NOISE_NONE if gauss else ops.NOISE_PHILOX

    def launch():
        k.rollout(f.descs["actor"], gauss, scale, base, big.internal, None if big.obs is big.internal else big.obs, big.action,
                  big.ep_len, big.ep_ret, big.ep_count, rows, cap, big.stats, big.ctrl, noise, tr.eps_start, tr.eps, tr.decay_value,
                  tr._box_lo, tr._box_hi, tr.max_steps, tr.corr_lr, tr.corr_eps, tr.corr_momentum, 200, True, 1e-3, big.seed,
                  big.env_id_base)
    for sel, label in ((1, "64-lane row tiles"), (3, "streaming form"), (4, "streaming, 64-lane groups forced"), (2, "size rule")):
        with ops.tuning(rollout_wide=sel):
            us = bench.time_kernel(launch, reps=10 if n > 200000 else 40)[0]
        tf = fl * n / us * 1e-6
        print("%-22s %-10s n=%-8d %-24s %9.2f us  %7.2f TFLOP/s = %.3f of the f32 MFMA peak  %8.1f M env-steps/s"
              % (lib, workload, n, label, us, tf, tf / bench.MFMA_F32_PEAK_TFLOPS, n / us), flush=True)
    del big, rows
    torch.cuda.empty_cache()
