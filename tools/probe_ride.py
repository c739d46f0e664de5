"""Stand-alone durations (hipGraph of 100 back-to-back launches, bench.time_kernel) of the pieces of the riding rollout
at the bench size: the column-split actor forward at 256 / 1024 / 4096 lanes, the critic update stages with and without
their riders, and the one-launch rollout.  python tools/probe_ride.py [workload]  -> profiles/r02_probe_ride_*.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
_tk = bench.time_kernel
bench.time_kernel = lambda fn, reps=100: _tk(fn, reps)[0]
from rpo_amd import ops
os.environ["RPO_VERBOSE"] = "0"
dev = torch.device("cuda")
tr = bench.make_trainer(4096, dev, 10 ** 9, workload=sys.argv[1] if len(sys.argv) > 1 else "cart_sac")
tr.vec.reset(); tr.run_steps(64); torch.cuda.synchronize()
d = tr.fused.descs["actor"]
n = 4096
obs = tr.vec.obs
part = torch.zeros(8, n, 2, device=dev)
out = torch.zeros(n, d.n_out, device=dev)
for rows in (256, 1024, 4096):
    p = part[:, :rows].contiguous(); o = obs[:rows]
    t = bench.time_kernel(lambda: ops.mlp_forward_split([(d, o, None, p, None, None)]))
    print("mlp_forward_split actor rows=%d: %.2f us" % (rows, t))
t = bench.time_kernel(lambda: ops.mlp_forward(d, obs, None, out))
print("mlp_forward (row tile) 4096: %.2f us" % t)
t = bench.time_kernel(lambda: ops.mlp_split_head(d, part, out))
print("mlp_split_head 4096: %.2f us" % t)
su = tr._split_state(); rd = tr._rider()
for st in ("critic_fwd_a", "critic_fwd_b", "critic_bwd_a", "critic_bwd_b"):
    print(st, "%.2f us" % bench.time_kernel(lambda: su.run(st)))
rd.set(lane_begin=0, lane_end=2048)
print("fwd_a_ride[0,2048) %.2f us" % bench.time_kernel(lambda: su.run("critic_fwd_a", rider=rd)))
rd.set(lane_begin=2048, lane_end=4096)
print("fwd_b_ride[2048,4096) %.2f us" % bench.time_kernel(lambda: su.run("critic_fwd_b", rider=rd)))
print("bwd_b_ride %.2f us" % bench.time_kernel(lambda: su.run("critic_bwd_b", rider=rd)))
gm = tr.agent.critic_optim.gradmax
for label, val in (("gradmax", gm), ("no gradmax", None)):
    su.set(gradmax=val)
    print("critic_bwd_b (%s) %.2f us" % (label, bench.time_kernel(lambda: su.run("critic_bwd_b"))))
su.set(gradmax=gm)
opt = tr.agent.critic_optim
print("adam (critic slice) %.2f us" % bench.time_kernel(lambda: opt.step(target=None, tau=0.0, gradmax_ready=True)))
tgt = torch.zeros_like(opt.param)
print("polyak on the same slice (no arrival counting) %.2f us, n = %d" % (bench.time_kernel(lambda: ops.polyak(opt.param, tgt, 0.005)), opt.param.numel()))
print("rollout %.2f us" % bench.time_kernel(lambda: tr._rollout(False)))
