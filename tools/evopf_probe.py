"""Per-kernel timings of the EVOPF-v0 path on one MI355X (hipGraph of back-to-back launches between two HIP events).

    python tools/evopf_probe.py [n_envs]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rpo_amd import ops  # noqa: E402
from rpo_amd.env import EVOPFEnv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    env = EVOPFEnv(device="cuda")
    k = env.kernels
    v = env.make_vec(n, seed=3)
    v.reset()
    lo, hi = env.update(v.obs)
    ap = (lo + 0.5 * (hi - lo)).contiguous()
    ap[:, 4:9] = 1.07                                   # violated bounds: the projection runs all its iterations
    rows = torch.zeros(8 * n, k.ring_floats, device="cuda")
    it = torch.zeros(n, dtype=torch.int32, device="cuda")
    out = {}

    def act(steps, m=n):
        return lambda: k.act_project(v.obs[:m], ap[:m], None, v.action[:m], it[:m], ops.NOISE_PHILOX, 1e-4, 1e-4, 0.0, 0, 0,
                                     steps, 1e-4, 1e-5, 0.0, 3, 0, v.ctrl, v.stats)
    out["act_project(complete only) n=%d" % n] = bench.time_kernel(act(0), reps=20)[0]
    out["act_project(10 GRG steps) n=%d" % n] = bench.time_kernel(act(10), reps=20)[0]
    out["act_project(10 GRG steps) n=256"] = bench.time_kernel(act(10, 256), reps=20)[0]
    out["act_project(complete only) n=256"] = bench.time_kernel(act(0, 256), reps=20)[0]
    act(10)()
    print("mean GRG iterations:", float(it.float().mean()))
    out["step n=%d" % n] = bench.time_kernel(lambda: k.step(v.internal, v.obs, v.action, v.ep_len, v.ep_ret, v.ep_count, rows,
                                                            8, v.stats, v.ctrl, 2 ** 31 - 1, True, 1e-3, 3, 0), reps=20)[0]
    g = torch.randn(256, 43, device="cuda")
    gap = torch.zeros(256 * 14, device="cuda")
    a256 = v.action[:256].contiguous()
    out["complete_bwd n=256"] = bench.time_kernel(lambda: k.complete_bwd(None, g, gap, action=a256), reps=20)[0]
    nu = torch.rand(58, device="cuda")
    loss, ga, gnu = torch.zeros(1, device="cuda"), torch.zeros(256, 43, device="cuda"), torch.zeros(58, device="cuda")
    out["lagrangian n=256"] = bench.time_kernel(lambda: k.lagrangian(a256, nu, 1 / 256, loss, ga, gnu, obs=v.obs[:256]), reps=20)[0]
    step_o = torch.zeros(256, 43, device="cuda")
    out["ineq_partial_grad n=256"] = bench.time_kernel(lambda: k.ineq_partial_grad(v.obs[:256], a256, step_o), reps=20)[0]
    for name, us in out.items():
        print("%-40s %9.1f us" % (name, us))


if __name__ == "__main__":
    main()
