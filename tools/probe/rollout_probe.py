#!/usr/bin/env python
"""Where do the 13 us of the headline's rollout launch go?  Times `trainer._rollout` (4096 lanes, one launch: actor MLP ->
box -> noise -> Complete -> GRG -> env step -> ring row) as a dependent chain inside a hipGraph; run it against timing-only
builds (RPO_HIP_LIBRARY=.../librpo_hip_skipN.so, tools/probe/build_stream_variants.sh with KIND=rollout)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cart_ddpg"
dev = torch.device("cuda")
bench.spin_up(dev, 1.0)
tr = bench.make_trainer(4096, dev, 10 ** 9, workload=workload)
tr.vec.reset()
tr.run_steps(8, train=False)
us = bench.time_kernel(lambda: tr._rollout(False))[0]
us_d = bench.time_kernel(lambda: tr._rollout(False, defer_clock=True))[0] if tr._defer_ok else float("nan")
print("%-24s %s rollout launch: %.2f us (step counter left to the next launch, as in the windows: %.2f us)"
      % (os.path.basename(os.environ.get("RPO_HIP_LIBRARY", "librpo_hip.so")), workload, us, us_d), flush=True)
