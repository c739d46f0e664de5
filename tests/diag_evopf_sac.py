import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["RPO_VERBOSE"] = "0"
from rpo_amd.algo import RPOSAC
from rpo_amd.env import EVOPFEnv
from rpo_amd.utils.logger import Logger
import test_train_step_golden as tsg
from oracle import evopf as oe
hp = {k: v for k, v in tsg.EVOPF_HP.items() if k not in ("embed_dim", "hidden_dim", "init_nju", "capacity")}
del hp["gamma"]; hp.update(grad_eps=0.1, alpha=0.001, automatic_entropy_tuning=False, fixed=False)
steps = 960
found = []
for seed in range(48):
    torch.manual_seed(123 + seed)
    tr = RPOSAC(EVOPFEnv(device="cuda"), "/tmp/rpo_test", name="t", logger=None, max_epochs=steps, capacity=20000,
                device=torch.device("cuda"), num_envs=1, seed=1000 + seed, **hp)
    tr.logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), times=1, epochs=steps)
    tr.run(eval=False)
    me = tr.logger.tracker["max_eq"][:tr.logger.pointer]
    if me.max() > 0.05:
        t = int(np.argmax(me)); print("seed", seed, "step", t, "max_eq", me.max(), "around", me[max(0,t-2):t+3])
        c = tr.kernels.cols
        row = tr.buffer.rows[t].cpu().numpy()
        s, a = row[c["state"][0]:c["state"][1]], row[c["action"][0]:c["action"][1]]
        found.append((seed, t, s, a))
        np.savez("gpurun_out/diag_sac_%d.npz" % seed, s=s, a=a, eq=row[c["eq_viol"][0]:c["eq_viol"][1]])
    del tr
G = oe.GRID
env = EVOPFEnv(device="cuda")
os.environ["RPO_EVOPF_PIVOT"] = "dynamic"; dyn = EVOPFEnv(device="cuda"); dyn.kernels; os.environ.pop("RPO_EVOPF_PIVOT")
for seed, t, s, a in found:
    ap = a[G.partial_actions][None]
    s64 = s[None].astype(np.float64)
    lo, hi = oe.partial_box(s64)
    print("ap", ap, "\nbox frac", (ap - lo) / (hi - lo))
    want, jac, jn, its = oe.complete_partial(s64, ap.astype(np.float64), return_aux=True)
    print("oracle newton iters", its, "eq", np.abs(oe.eq_resid(s64, want)).max(), "vm", want[0, G.vm0:G.vm0+14])
    for name, e in (("static", env), ("dynamic", dyn)):
        got = e.complete_partial(torch.tensor(s[None], device="cuda"), torch.tensor(ap, device="cuda")).cpu().numpy()
        print(name, "eq", np.abs(oe.eq_resid(s64, got.astype(np.float64))).max(), "diff vs oracle", np.abs(got - want).max())
        p, it = e.project(torch.tensor(s[None], device="cuda"), torch.tensor(ap, device="cuda"), 10, 1e-4, return_iters=True)
        print(name, "project eq", np.abs(oe.eq_resid(s64, p.cpu().numpy().astype(np.float64))).max(), "iters", it.cpu().numpy())
    print("stored action eq", np.abs(oe.eq_resid(s64, a[None].astype(np.float64))).max())
