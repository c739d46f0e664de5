"""The UTD-matched leg of bench.py alone (updates_per_step = num_envs: 4096 sequential batch-256 updates per vector step):
us per update, and where the time goes (device-side kernel time needs rocprofv3; here: wall clock per update for a few
window lengths).   python tools/probe/utd_probe.py [workload]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "cart_ddpg"
dev = torch.device("cuda")
bench.spin_up(dev)
EPG = bench.envs_per_gpu(workload)
for cycle in (16, 64, 16):
    os.environ["RPO_GRAPH_CYCLE"] = str(cycle)
    utd = bench.make_trainer(EPG, dev, 10 ** 9, capacity=256, workload=workload, updates_per_step=EPG)
    utd.vec.reset()
    utd.run_steps(1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        utd.run_steps(4)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (4 * EPG) * 1e6)
    print("%s, windows of %d updates: %.2f us per sequential update" % (workload, cycle, best), flush=True)
    del utd
