#!/usr/bin/env python
"""Benchmark of the RPO hot path on MI355X: env-steps/s of RPODDPG on CartSafe-v0 with 4096 vectorised envs per GPU
(BASELINE.json configs[1]), one constrained policy update of batch 256 per vector step (the reference's cadence,
rpo/algo/rpo_ddpg.py:160-161).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one vector step: actor forward on N lanes -> noise + equation solver + GRG projection -> fused env step +
violation bookkeeping + replay scatter -> sample/gather 256 transitions -> critic (and every 4th step actor + dual)
update.  Rank 0 prints ONE JSON line; everything else goes to stderr.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per env-step of the fused step + violation + replay-scatter kernel (CartSafe):
# read s 24 + a 8, write s' 24 (in place) + transition row 89
STEP_BYTES_PER_ENV = 145
HP = dict(batch_size=256, max_steps=10, warmup=0, lr_dual=0.2, corr_lr=2e-2, eps=1.0, eps_start=1.0, lr_actor=1e-4,
          lr_critic=3e-4, eps_epoch=20000, eval_lr=2e-2, eval_steps=50, grad_eps=0.1, corr_momentum=0.0, policy_fre=4,
          capacity=20000, shared_param=True, value_type="add", clip_thres=0.2, embed_dim=128,
          hidden_dim=256)        # scripts/cart_exp.py:26-28


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# the other scripts' hyper-parameters, for --workload
WORKLOADS = {
    "cart_ddpg": ("cart", "ddpg", {}),
    "cart_sac": ("cart", "sac", dict(eps=5e-3, eps_start=5e-3, shared_param=False, alpha=0.1, automatic_entropy_tuning=False)),
    "pen_ddpg": ("pen", "ddpg", dict(lr_dual=0.01, corr_lr=2e-3, eval_lr=2e-3, eps=0.5, eps_start=0.5, shared_param=False)),
    "pen_sac": ("pen", "sac", dict(lr_dual=0.01, corr_lr=2e-3, eval_lr=2e-3, eps=1e-2, eps_start=1e-2, shared_param=False,
                                   alpha=0.01, automatic_entropy_tuning=False)),
    # BASELINE.json configs[4]: EVOPF-v0, RPODDPG, 1024 envs on one MI355X (scripts/evopf_exp.py:29-31)
    "evopf_ddpg": ("evopf", "ddpg", None),
    # scripts/evopf_exp_sac.py:30-33
    "evopf_sac": ("evopf", "sac", dict(lr_actor=1e-4, lr_critic=3e-4, grad_eps=0.1, fixed=False, init_lamb=0.0, init_nju=0.0,
                                       alpha=0.001, automatic_entropy_tuning=False)),
}
EVOPF_HP = dict(batch_size=256, max_steps=10, warmup=0, lr_dual=2e-2, corr_lr=1e-4, eps=0.0001, eps_start=0.0001,
                eps_epoch=20000, eval_lr=1e-4, eval_steps=50, grad_eps=0.02, corr_momentum=0.0, policy_fre=4,
                ex_action_dim=1, gamma=0.95, capacity=20000, clip_thres=0.2, shared_param=False, value_type="cat")


def envs_per_gpu(workload):
    return 1024 if workload.startswith("evopf") else ENVS_PER_GPU


def make_trainer(n_envs, device, max_epochs, capacity=None, workload="cart_ddpg", updates_per_step=None):
    from rpo_amd import gym_shim
    from rpo_amd.algo import RPODDPG, RPOSAC
    from rpo_amd.env import CartSafeEnv, EVOPFEnv, SpringPendulumEnv
    np.random.seed(123)
    torch.manual_seed(123)                  # identical replicas on every rank
    envname, algo, over = WORKLOADS[workload]
    if envname == "evopf":
        env, hp = EVOPFEnv(), dict(EVOPF_HP)
        if over:
            hp.update(over)
            hp.pop("gamma", None)                     # evopf_exp_sac.py keeps RPOSAC's default discount
    else:
        env = gym_shim.TimeLimit(CartSafeEnv() if envname == "cart" else SpringPendulumEnv(), 200)
        hp = dict(HP)
        hp.update(over)
    if capacity is not None:
        hp["capacity"] = capacity
    cls = RPODDPG if algo == "ddpg" else RPOSAC
    return cls(env, "/tmp/rpo_bench", name="bench", logger=None, max_epochs=max_epochs, device=device,
               num_envs=n_envs, updates_per_step=updates_per_step, **hp)


def spin_up(device, seconds=1.5):
    """Untimed: keep the GPU busy for a moment before anything is measured, so that a fresh box has left its idle
    power state (one in ~10 fresh boxes otherwise measured the first window ~40 % slow)."""
    x = torch.randn(4096, 4096, device=device)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            x = torch.mm(x, x).clamp_(-1.0, 1.0)
        torch.cuda.synchronize()


def time_kernel(fn, reps=100):
    """Average duration (us) of one launch: `reps` back-to-back launches captured in a hipGraph and bracketed by ONE pair
    of HIP events on the replay stream (an event pair around a single launch has a ~13 us floor on this stack, far above
    these kernels).  The figure includes the ~1.5 us dependent-launch boundary between consecutive kernels."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / reps)
    return float(np.median(times)), float(np.mean(times))


MFMA_F32_PEAK_TFLOPS = 157.3              # v_mfma_f32_16x16x4_f32 == the f32 vector peak (MI355X_MICROARCH.md)
ACTOR_FLOPS = 2 * (6 * 128 + 128 * 256 + 256)                  # per row, rpo/algo/model/policy.py:24-33 at 6-128-256-1
CRITIC_FLOPS = 2 * (6 * 128 + 2 * 128 + 128 * 256 + 256)       # per row, model/value.py:51-59


def kernel_clinic(tr):
    """Per-launch durations of the kernels of one iteration (at the bench size) and of the streaming kernels at 1M
    lanes.  HBM-bound kernels are priced in algorithmic bytes, the MLP pipelines in flops."""
    from rpo_amd import ops
    from rpo_amd.env.vec import VecEnv
    out = {}
    k, v, buf, f, B = tr.kernels, tr.vec, tr.buffer, tr.fused, tr.batch_size
    scale, base = tr._box_affine
    dev = v.device

    def step_at(vec, rows, cap):
        return lambda: k.step(vec.internal, vec.obs, vec.action, vec.ep_len, vec.ep_ret, vec.ep_count, rows, cap,
                              vec.stats, vec.ctrl, 200, True, 1e-3, vec.seed, vec.env_id_base)

    def act_at(vec, a):
        return lambda: k.act_project(vec.obs, a, None, vec.action, None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10.0, 10.0, 10,
                                     2e-2, 1e-5, 0.0, vec.seed, vec.env_id_base, vec.ctrl, vec.stats)

    def rollout_at(vec, rows, cap):
        return lambda: k.rollout(f.descs["actor"], False, scale, base, vec.internal, None, vec.action, vec.ep_len,
                                 vec.ep_ret, vec.ep_count, rows, cap, vec.stats, vec.ctrl, ops.NOISE_PHILOX, 1.0, 1.0, 0.0,
                                 -10.0, 10.0, 10, 2e-2, 1e-5, 0.0, 200, True, 1e-3, vec.seed, vec.env_id_base)

    def hbm(name, n, us, bytes_per_unit):
        out[name] = dict(n=n, us=us, bound="hbm", work=bytes_per_unit * n, rate=bytes_per_unit * n / us * 1e-3,
                         unit="GB/s", peak=HBM_PEAK_GBS)

    def mfma(name, n, us, flops_per_unit):
        out[name] = dict(n=n, us=us, bound="mfma", work=flops_per_unit * n, rate=flops_per_unit * n / us * 1e-6,
                         unit="TFLOP/s", peak=MFMA_F32_PEAK_TFLOPS)

    # ---- the launches of one iteration at the bench size
    mfma("rollout_kernel<CartEnv>", v.n, time_kernel(rollout_at(v, buf.rows, buf.capacity))[0], ACTOR_FLOPS)
    d = f.descs["critic"]
    cf = lambda: k.ddpg_critic_forward(  # noqa: E731
        f.descs["actor_target"], f.descs["critic_target"], d, scale, base, buf.rows, buf.capacity, buf.n_envs, tr._batch,
        None, None, buf.seed, 0, buf.ctrl, 10, 2e-2, 1e-5, 0.0, -10.0, 10.0, f.buf("q", B, 1), f.buf("qn", B, 1),
        f.buf("critic.x0", B, d.ein), f.buf("critic.h1", B, d.H))
    mfma("cart_ddpg_critic_forward_kernel", B, time_kernel(cf)[0], ACTOR_FLOPS + 2 * CRITIC_FLOPS)
    cols = tr.buffer.split(tr._batch)
    td = ops.Td(f.buf("q", B, 1).view(-1), f.buf("qn", B, 1).view(-1), None, None, cols["reward"], cols["done"], 0.0, 0.95,
                f.buf("dq", B, 1).view(-1), f.buf("loss_parts", (B + 15) // 16))
    bw = lambda: f.backward("critic", cols["state"], cols["action"], None, td=td)   # noqa: E731  (TD prologue included)
    mfma("mlp_bwd_rows+weights_kernels", B, time_kernel(bw)[0], 2 * CRITIC_FLOPS)
    # ---- single-stage kernels (warm-up phase, SAC / pendulum path, API calls) and the streaming regime
    ap = torch.zeros(v.n, device=dev)
    hbm("cartsafe_step_kernel", v.n, time_kernel(step_at(v, buf.rows, buf.capacity))[0], STEP_BYTES_PER_ENV)
    hbm("cartsafe_act_project_kernel", v.n, time_kernel(act_at(v, ap))[0], 40)
    batch = torch.zeros(B, k.row_floats, device=dev)
    hbm("replay_sample_gather_kernel", B, time_kernel(lambda: ops.replay_sample_gather(
        buf.rows, buf.capacity, buf.n_envs, batch, None, 1, 0, v.ctrl))[0], 178 + 4)
    big_n = 1 << 20
    big = VecEnv(k, big_n, dev, seed=3, stats_cap=64)
    big.reset()
    rows = torch.zeros(8 * big_n, k.row_floats, device=dev)
    big_ap = torch.zeros(big_n, device=dev)
    hbm("cartsafe_step_kernel@1M", big_n, time_kernel(step_at(big, rows, 8), reps=20)[0], STEP_BYTES_PER_ENV)
    hbm("cartsafe_act_project_kernel@1M", big_n, time_kernel(act_at(big, big_ap), reps=20)[0], 40)
    big_batch = torch.zeros(big_n, k.row_floats, device=dev)
    hbm("replay_sample_gather_kernel@1M", big_n, time_kernel(lambda: ops.replay_sample_gather(
        rows, 8, big_n, big_batch, None, 1, 0, big.ctrl), reps=20)[0], 178 + 4)
    mfma("rollout_kernel<CartEnv>@1M", big_n, time_kernel(rollout_at(big, rows, 8), reps=5)[0], ACTOR_FLOPS)
    for name, e in out.items():
        e["frac"] = e["rate"] / e["peak"]
        log("  %-36s n=%-8d %9.2f us  %9.2f %-8s (%.1f%% of the %s peak)" % (name, e["n"], e["us"], e["rate"], e["unit"],
                                                                           100 * e["frac"], e["bound"]))
    del big, rows, big_batch
    torch.cuda.empty_cache()
    return out


def pmc_traffic(kernel, lanes):
    """HBM bytes per launch from the committed rocprofv3 PMC collection (profiles/r01_pmc_traffic.json; recipe and the
    gfx950 FETCH_SIZE correction are described there).  None when that (kernel, size) was not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["kernels"][kernel][str(lanes)]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(seconds=15.0):
    """The oracle's single-env, per-step CPU loop (oracle/rpo_loop.py: the reference's cadence -- one env step + one
    batch-256 update per iteration) timed on one host core for a bounded sample."""
    from oracle import rpo_loop
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                # the reference is 22x slower with 8 intra-op threads
    try:
        np.random.seed(123)
        torch.manual_seed(123)
        hp = {k: v for k, v in HP.items() if k not in ("grad_eps", "value_type")}
        tr = rpo_loop.OracleRPO(rpo_loop.CartAdapter(1, seed=0), sac=False, **hp)
        tr.run(50)                          # page in
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            tr.run(100)
            done += 100
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return dict(value=done / dt, unit="env-steps/s", cores=1, kind="port",
                sample="%d env steps (rollout + one batch-256 update each) of oracle/rpo_loop.py OracleRPO on "
                       "CartSafe-v0, 1 env, 1 thread, %.1f s" % (done, dt))


def main():
    os.environ.setdefault("RPO_VERBOSE", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clinic", action="store_true")
    ap.add_argument("--workload", default="cart_ddpg", choices=sorted(WORKLOADS),
                    help="cart_ddpg is the headline (BASELINE.json configs[1]); the others are extra measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    EPG = envs_per_gpu(args.workload)
    n_total = EPG * world
    spin_up(device)
    # untimed pre-conditioning on a throw-away trainer (tiny replay ring): the first process on a fresh box otherwise pays
    # for paging in the libraries and code paths of the iteration inside the timed window (measured: 0.14 instead of
    # 0.10 ms per iteration in 4 of 4 first-process runs)
    pre = make_trainer(n_total, device, 10 ** 9, capacity=64, workload=args.workload)
    pre.vec.reset()
    pre.run_steps(1500)
    pre._harvest(final=True)                # first use of the statistics path (gather / reduce / copy-out kernels)
    torch.cuda.synchronize()
    del pre
    tr = make_trainer(n_total, device, 10 ** 9, workload=args.workload)
    headline = args.workload == "cart_ddpg"
    tr.vec.reset()
    # untimed setup, independent of --warmup: three eager passes + capture of every hipGraph the steady state replays
    # (single iterations with / without the policy step, and the multi-iteration window), like a compile step
    tr.run_steps(5 * max(tr._cycle, 4))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    tr.run_steps(args.warmup)
    if os.environ.get("RPO_BENCH_DEBUG"):    # extra untimed windows, to see drift / host stalls (stderr)
        for w in range(6):
            fence()
            tw = time.perf_counter()
            tr.run_steps(500)
            fence()
            log("debug window %d: %.4f ms/iter" % (w, (time.perf_counter() - tw) / 500 * 1e3))
    fence()
    t0 = time.perf_counter()
    tr.run_steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    tr._harvest(final=True)
    value = n_total * args.steps / elapsed

    result = {
        "metric": "env-steps/sec (whole node), %s, rollout + one batch-256 constrained policy update per vector step"
                  % ("SafeCartpole-v0 (CartSafe-v0) RPODDPG" if headline else args.workload),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("CartSafe-v0 RPODDPG, %d vectorised envs per MI355X, scripts/cart_exp.py "
                                "hyper-parameters, update batch 256 every vector step (reference cadence), replay "
                                "capacity 20000 per env" % EPG) if headline else
                               "%s, %d vectorised envs per MI355X (extra measurement, not the headline)" % (args.workload, EPG),
                   "envs_per_gpu": EPG, "global_envs": n_total, "update_batch": 256,
                   "parallelism": "dp%d (env shards, RCCL all-reduce of the flat gradient bucket)" % world,
                   "hip_graph": bool(tr._graphs.enabled)},
        "constraint_violation_rate": tr.viol_rate,
        "mean_projection_iters": tr.proj_iters_mean,
    }

    if rank == 0:
        # (i) rollout-only throughput next to the headline, so that the cadence is visible
        ro = make_trainer(EPG, device, 10 ** 9, capacity=64, workload=args.workload) if world == 1 else None
        if ro is not None:
            ro.vec.reset()
            ro.run_steps(5 * max(ro._cycle, 4), train=False)      # eager passes + graph capture
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ro.run_steps(1000, train=False)
            torch.cuda.synchronize()
            result["rollout_only_env_steps_per_s"] = EPG * 1000 / (time.perf_counter() - t1)
            # (iii) UTD-matched: one batch-256 update per ENV step as in the reference, i.e. EPG updates per
            # vector step -- a bounded sample of vector steps
            utd = make_trainer(EPG, device, 10 ** 9, capacity=256, workload=args.workload,
                               updates_per_step=EPG)
            utd.vec.reset()
            utd.run_steps(1)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            utd.run_steps(4)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t2
            result["utd_matched_env_steps_per_s"] = EPG * 4 / dt
            result["utd_matched_updates_per_s"] = EPG * 4 / dt
            del utd
            del ro
        if not args.no_clinic and world == 1 and tr.fused is not None and headline:
            log("kernel clinic (hipGraph of back-to-back launches between two HIP events on the launch stream):")
            clinic = kernel_clinic(tr)
            # single launches of the iteration (the backward entry is a PAIR of launches, ~half each)
            in_iter = ("rollout_kernel<CartEnv>", "cart_ddpg_critic_forward_kernel")
            dom = max(in_iter, key=lambda n: clinic[n]["us"])
            d, st = clinic[dom], clinic["cartsafe_step_kernel@1M"]
            result["roofline"] = {
                "bound": d["bound"], "kernel": dom, "achieved": d["rate"], "peak": d["peak"], "unit": d["unit"],
                "frac": d["frac"], "traffic": pmc_traffic(dom, d["n"]), "launch_us": d["us"], "units_per_launch": d["n"],
                "algorithmic_flops_per_launch": d["work"],
                "note": "dominant launch of the iteration: ReplayBuffer.sample + pi_targ + projection + Q_targ | Q for 256 "
                        "samples, two independent workgroups per 16-row tile (TD/Huber is the prologue of the backward "
                        "pass); latency-bound (32 workgroups, the longer chain streams 2 x 128 KB of f32 weights at "
                        "~47 GB/s per CU). Streaming regime of the HBM-bound env-step "
                        "kernel (1M lanes): %.0f GB/s = %.3f of the 8 TB/s peak, PMC traffic %s B vs %d algorithmic B."
                        % (st["rate"], st["frac"], pmc_traffic("cartsafe_step_kernel", st["n"]), st["work"]),
                "hbm_streaming": {"kernel": "cartsafe_step_kernel", "bound": "hbm", "units_per_launch": st["n"],
                                  "launch_us": st["us"], "achieved": st["rate"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": st["frac"], "traffic": pmc_traffic("cartsafe_step_kernel", st["n"]),
                                  "algorithmic_bytes_per_launch": st["work"]},
                "all_kernels": {kk: {"us": vv["us"], "rate": vv["rate"], "unit": vv["unit"], "frac": vv["frac"]}
                                for kk, vv in clinic.items()},
            }
        if not args.no_cpu_baseline and world == 1 and headline:
            result["cpu_baseline"] = cpu_baseline()
            result["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
