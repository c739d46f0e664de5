#!/usr/bin/env python
"""Pre-flight of the first multi-GPU run (VERDICT r05 missing 2 / next 6; SURVEY 8e, BASELINE.json configs[3]).

No multi-rank RCCL run of this project exists yet: every box the builder gets has ONE GPU, and the driver's 8-GPU SCALE run is
the first time the collectives captured inside the 16-iteration hipGraph windows are replayed by more than one rank.  This
script makes that first run boring.  On a box with >= 2 GPUs (or, with ``--backend gloo``, on one GPU shared by the ranks: the
control-flow form that `-m gpu` exercises, tests/test_bench_gpu.py) it runs, each leg in FRESH child processes started before
this process has touched a GPU (an `exec` from a GPU-initialised process takes the pool's machines down):

  replicas   N ranks of config 4's per-rank workload (CartSafe-v0 RPOSAC, `--lanes` per rank, scripts/cart_exp_sac.py), 64
             iterations in hipGraph windows, then an all-gather of every rank's flat parameter buffer, targets and multipliers:
             replicas must be BIT-identical (identical optimiser steps on all-reduced gradients; no parameter broadcast after
             construction), the lanes must differ between ranks (sharded env ids), and every rank must have replayed graphs.
  timing     the same workload as one plain process, as a ONE-rank group (the measured intercept of DESIGN 7: one all-reduce +
             rpo_absmax_slots per collective), and as N ranks: microseconds per iteration and per collective.
  bench      `bench.py --gpus N --steps 20 --warmup 5 --no-cpu-baseline` (the driver's SCALE flags), and the same with
             RPO_GRAPH_CYCLE=0 (eager launches, eager collectives): which of the two is broken tells graph capture from RCCL.
  rccl test  tests/test_trainer_gpu.py::test_rccl_two_ranks_one_per_gpu (skips itself below two GPUs).

One JSON verdict on stdout (everything else: stderr); DESIGN.md 7 says what each outcome means.  Exit code 0 iff verdict "ok".
    python tools/scale_preflight.py [--gpus N] [--backend nccl|gloo] [--lanes 4096] [--quick] [--out FILE]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITERS = 64


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------------ the ranks
def worker(args):
    """One rank (started by torch.distributed.run, or alone): config 4's per-rank workload, ITERS iterations + timed windows."""
    sys.path.insert(0, ROOT)
    os.environ.setdefault("RPO_VERBOSE", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import bench
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    grouped = world > 1 or args.one_rank_group
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ["RPO_SCHEDULE"] = ",".join(filter(None, [os.environ.get("RPO_SCHEDULE", ""), "force_dist=1"]))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    tr = bench.make_trainer(args.lanes * world, dev, 10 ** 9, capacity=64, workload="cart_sac", torch_seed=123 + rank)
    tr.vec.reset()
    tr.run_steps(ITERS)
    tr._harvest(final=True)
    torch.cuda.synchronize()
    out = {"rank": rank, "world": world, "lanes_per_rank": tr.n_local, "data_parallel_path": bool(tr.dist.on),
           "backend": dist.get_backend() if grouped else None,
           "rccl_ranks": dist.get_world_size() if (grouped and dist.get_backend() == "nccl") else 0,
           "collectives_in_graph": bool(tr.dist.on and tr.dist.in_graph),
           "graph_window_iterations": tr._cycle, "graphs_enabled": bool(tr._graphs.enabled),
           "graph_capture_fell_back_to_eager": bool(tr._graphs.capture_failed),
           "graphs_captured": sorted(str(k) for k, e in tr._graphs.entries.items() if e["graph"] is not None)}
    if world > 1:
        # replicas: every rank's parameters / targets / multipliers after ITERS iterations, and a checksum of its lanes
        blob = torch.cat([tr.agent.flat.data.reshape(-1), tr.agent.critic_target_flat.reshape(-1), tr.agent.nju.weight.data.reshape(-1)])
        blobs = [torch.empty_like(blob) for _ in range(world)]
        dist.all_gather(blobs, blob)
        lanes = tr.vec.internal.double().sum().reshape(1)
        sums = [torch.empty_like(lanes) for _ in range(world)]
        dist.all_gather(sums, lanes)
        replayed = tr.dist.replaying_everywhere(tr._graphs, dev)
        out.update(replicas_bit_equal=bool(all(torch.equal(blobs[0], b) for b in blobs[1:])),
                   parameters_finite=bool(torch.isfinite(blob).all()),
                   lanes_differ_between_ranks=bool(len({float(s) for s in sums}) == world),
                   graphs_replayed_on_all_ranks=bool(replayed))
    # timing: windows of ITERS iterations between fences, max over ranks, median of 5
    regions = []
    for _ in range(1 if args.quick else 5):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        tr.run_steps(ITERS)
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        regions.append(float(dt.item()))
    out["us_per_iteration"] = float(np.median(regions)) / ITERS * 1e6
    out["collectives_per_update"] = (1.0 + 1.0 / tr.policy_fre) if tr.dist.on else 0.0
    tr._harvest(final=True)                                     # (the failure flags once more: a NaN replica must not pass)
    if rank == 0:
        print("PREFLIGHT " + json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------ the legs
def child(cmd, env_extra, budget):
    """Run `cmd` as a fresh process group; -> (rc or None on timeout, stdout, stderr tail)."""
    import signal
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("RPO_VERBOSE", "0")
    log("preflight: " + " ".join(cmd))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=budget)
        return p.returncode, out, err[-3000:]
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)                         # exactly the process group started above
        out, err = p.communicate()
        return None, out, err[-3000:]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_worker(world, backend, lanes, quick, one_rank_group=False, budget=600, env_extra=None):
    me = os.path.abspath(__file__)
    tail = ["--worker", "--backend", backend, "--lanes", str(lanes)] + (["--quick"] if quick else []) + \
        (["--one-rank-group"] if one_rank_group else [])
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
               "127.0.0.1", "--master-port", str(free_port()), me] + tail
    else:
        cmd = [sys.executable, me] + tail
        env_extra = dict(env_extra or {}, MASTER_PORT=str(free_port()))
    rc, out, err = child(cmd, env_extra or {}, budget)
    rec = None
    for line in out.splitlines():
        if line.startswith("PREFLIGHT "):
            rec = json.loads(line[len("PREFLIGHT "):])
    return {"rc": rc, "result": rec, **({"stderr_tail": err} if rc != 0 or rec is None else {})}


def run_bench(world, backend, graph_cycle, budget):
    env = {"RPO_BENCH_BACKEND": backend}
    if graph_cycle is not None:
        env["RPO_GRAPH_CYCLE"] = str(graph_cycle)
    rc, out, err = child([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline"], env, budget)
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    line = json.loads(lines[-1]) if lines else None
    keep = None
    if line is not None:
        cfg = line["config"]
        keep = {"value": line["value"], "n_gpus": line["n_gpus"], "ms_per_step": line["ms_per_step"],
                "rccl_ranks": cfg["rccl_ranks"], "collective_backend": cfg["collective_backend"],
                "collectives_in_graph": cfg["collectives_in_graph"], "graphs_replayed_on_all_ranks": cfg["graphs_replayed_on_all_ranks"],
                "graph_capture_fell_back_to_eager": cfg["graph_capture_fell_back_to_eager"], "hip_graph": cfg["hip_graph"],
                "cart_sac_env_steps_per_s": line.get("cart_sac_env_steps_per_s"), "one_json_line_on_stdout": len(lines) == 1,
                "roofline_kernel": (line.get("roofline") or {}).get("kernel"),
                "cart_sac_roofline_kernel": (line.get("cart_sac_roofline") or {}).get("kernel")}
    return {"rc": rc, "line": keep, **({"stderr_tail": err} if rc != 0 or keep is None else {})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks (default: every visible GPU, at least 2)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--lanes", type=int, default=4096, help="env lanes per rank (config 4: 4096)")
    ap.add_argument("--quick", action="store_true", help="one timed window per leg, no bench legs (the -m gpu test)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--one-rank-group", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import torch                                                # (device_count does not initialise the GPU)
    have = torch.cuda.device_count()
    n = args.gpus or max(2, have)
    verdict = {"gpus_visible": have, "ranks": n, "backend": args.backend, "lanes_per_rank": args.lanes, "iterations": ITERS,
               "what": "pre-flight of the N > 1 path; outcomes are explained in DESIGN.md 7"}
    if args.backend == "nccl" and have < n:
        verdict.update(verdict="not run: %d ranks over RCCL need %d GPUs, %d visible (one rank per GPU; --backend gloo shares a GPU "
                               "for the control-flow form)" % (n, n, have))
        print(json.dumps(verdict))
        return 3
    t0 = time.time()
    legs = {}
    legs["plain_process"] = run_worker(1, args.backend, args.lanes, args.quick)
    if args.backend == "nccl":
        legs["one_rank_group"] = run_worker(1, "nccl", args.lanes, args.quick, one_rank_group=True)
    legs["n_ranks"] = run_worker(n, args.backend, args.lanes, args.quick)
    legs["n_ranks_eager"] = run_worker(n, args.backend, args.lanes, True, env_extra={"RPO_GRAPH_CYCLE": "0"})
    if not args.quick:
        legs["bench"] = run_bench(n, args.backend, None, 1500)
        legs["bench_eager"] = run_bench(n, args.backend, 0, 1500)
        rc, out, err = child([sys.executable, "-m", "pytest", "tests/test_trainer_gpu.py", "-q", "-m", "gpu", "-k",
                              "test_rccl_two_ranks_one_per_gpu", "-rs"], {}, 600)
        legs["rccl_two_rank_test"] = {"rc": rc, "summary": (out.strip().splitlines() or [""])[-1]}
    verdict["legs"] = legs
    res = {k: (v.get("result") or {}) for k, v in legs.items()}
    nr, plain = res["n_ranks"], res["plain_process"]
    problems = []
    for name in ("plain_process", "n_ranks", "n_ranks_eager") + (("one_rank_group",) if "one_rank_group" in legs else ()):
        if legs[name]["rc"] != 0 or not res[name]:
            problems.append("%s: child failed (rc %s)" % (name, legs[name]["rc"]))
    if nr:
        if not nr.get("replicas_bit_equal"):
            problems.append("replicas DIVERGED: ranks hold different parameters after %d iterations" % ITERS)
        if not nr.get("parameters_finite"):
            problems.append("non-finite parameters")
        if not nr.get("lanes_differ_between_ranks"):
            problems.append("ranks stepped the SAME lanes (env id sharding broken)")
        if args.backend == "nccl" and not (nr.get("collectives_in_graph") and nr.get("graphs_replayed_on_all_ranks")):
            problems.append("collectives were NOT replayed from hipGraphs on every rank (fell back to eager launches)")
        if args.backend == "nccl" and nr.get("rccl_ranks") != n:
            problems.append("RCCL saw %s ranks, expected %d" % (nr.get("rccl_ranks"), n))
    if res.get("n_ranks_eager") and not res["n_ranks_eager"].get("replicas_bit_equal"):
        problems.append("replicas diverged with EAGER collectives too: not a graph-capture problem")
    for name in ("bench", "bench_eager"):
        if name in legs and (legs[name]["rc"] != 0 or not legs[name]["line"]):
            problems.append("%s: bench.py --gpus %d failed (rc %s)" % (name, n, legs[name]["rc"]))
    # microseconds per collective: what N ranks add over the plain process, per collective of the iteration, next to the
    # one-rank group's (the launch floor of RCCL's kernel + rpo_absmax_slots, DESIGN 7: ~5.2 us)
    if nr and plain and nr.get("collectives_per_update"):
        per = nr["collectives_per_update"]
        verdict["us_per_iteration"] = {k: res[k].get("us_per_iteration") for k in res if res[k]}
        verdict["us_per_collective"] = {"n_ranks": (nr["us_per_iteration"] - plain["us_per_iteration"]) / per}
        if res.get("one_rank_group"):
            verdict["us_per_collective"]["one_rank_group"] = (res["one_rank_group"]["us_per_iteration"] - plain["us_per_iteration"]) / per
        verdict["weak_scaling_efficiency_estimate"] = plain["us_per_iteration"] / nr["us_per_iteration"]
    verdict["replicas_bit_equal_after_%d_iterations" % ITERS] = bool(nr.get("replicas_bit_equal")) if nr else None
    verdict["rccl_ranks_seen"] = nr.get("rccl_ranks") if nr else None
    verdict["collectives_in_graph"] = nr.get("collectives_in_graph") if nr else None
    verdict["graphs_replayed_on_all_ranks"] = nr.get("graphs_replayed_on_all_ranks") if nr else None
    verdict["problems"] = problems
    verdict["seconds"] = time.time() - t0
    verdict["verdict"] = "ok" if not problems else "FAILED"
    if args.backend == "gloo":
        verdict["note"] = "gloo: control-flow form (host-driven collectives between hipGraph segments; ranks may share a GPU) -- says " \
                          "nothing about RCCL, xGMI or scaling"
    text = json.dumps(verdict)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)
    return 0 if not problems else 1


if __name__ == "__main__":
    sys.exit(main() or 0)
