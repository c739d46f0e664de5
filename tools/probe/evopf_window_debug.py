"""Debug aid: EVOPF windows of RPO_GRAPH_CYCLE iterations with / without the policy prefix branch (one config per process)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402

cycle, prefix, fused, algo, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
os.environ["RPO_GRAPH_CYCLE"] = cycle
import test_train_step_golden as tsg  # noqa: E402
from rpo_amd import ops  # noqa: E402
from rpo_amd.algo.trainer import RPOTrainerBase  # noqa: E402
if not prefix:
    RPOTrainerBase._policy_prefix_enabled = False
torch.manual_seed(5)
tr = tsg.build_trainer(algo, "evopf256" if fused else "evopf", ops, torch.device("cuda"), fused=bool(fused), num_envs=64, use_graph=True)
tr.vec.reset()
tr.run_steps(iters)
torch.cuda.synchronize()
print("ok", sys.argv[1:], [k for k, e in tr._graphs.entries.items() if e["graph"] is not None], float(tr.agent.flat.data.abs().sum()))
