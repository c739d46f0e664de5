#!/usr/bin/env python
"""Learning quality at the VECTORISED cadences: cart-RPODDPG (scripts/cart_exp.py hyper-parameters) at
num_envs = 4096 in (a) the reference cadence -- one batch-256 update per vector step, bench.py's headline -- and (b) the
large-batch mode -- one batch-2^20 update per vector step, 256 sampled transitions per env step like the reference --, to a
fixed budget of UPDATES (3000, the budget of tests/golden/training_stats_ddpg_cart.npz).

Every lane is an env instance of its own, so the reference's per-run statistics (tests/golden/make_golden.py _stats_run:
violation rate = fraction of logged env steps with max(max_ineq, max_eq) > 1e-3; `reward` row = return of the episode a
step belongs to, back-filled, only completed episodes are logged; mean over the run and over its second half) are computed
PER LANE from the transitions the step kernel wrote into the replay ring (capacity = the whole run), exactly as the
reference's Logger would have recorded that lane, then averaged over the 4096 lanes of a run.  Lanes of one run share one
policy, so a RUN is one sample; the spread is over seeds.

    python tools/cadence_learning.py [seeds=128] [steps=3000] [large_batch_seeds=32]   # writes gpurun_out/cadence_learning.json

Round 5: 128 seeds at the reference cadence (0.12 s of GPU each) and 32 at the large batch (17 s each)
-- 8 + 8 before, whose standard error of 1.8e-3 on the violation rate could not see the north_star's 1e-3.  The per-seed rows
are kept in the JSON; tests/test_host_utils.py asserts the committed file's resolution and bounds on the CPU.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("RPO_VERBOSE", "0")
import bench  # noqa: E402

LANES = 4096


def lane_statistics(tr, steps, chunk=100):
    """Per-lane Logger statistics from the replay ring of a finished run -> (dict of seed-level means, curves per `chunk`)."""
    k, n = tr.kernels, tr.n_local
    c = k.cols
    rows = tr.buffer.rows[: steps * n].view(steps, n, -1)
    r = rows[:, :, c["reward"][0]].double()
    done = rows[:, :, c["done"][0]] > 0.5
    eq = rows[:, :, c["eq_viol"][0]:c["eq_viol"][1]].abs().amax(dim=2)
    ineq = rows[:, :, c["ineq_viol"][0]:c["ineq_viol"][1]].amax(dim=2)
    viol = (torch.maximum(eq, ineq) > 1e-3).double()
    ep = torch.cumsum(done.long(), dim=0) - done.long()          # episodes of the lane completed before step t
    n_done = done.long().sum(dim=0, keepdim=True)                # completed episodes per lane
    logged = (ep < n_done)                                       # the reference logs a step when its episode ends (rpo_ddpg.py:135-137)
    ids = ep * n + torch.arange(n, device=rows.device)[None, :]
    ret = torch.zeros(int(ids.max()) + 1, dtype=torch.float64, device=rows.device).scatter_add_(0, ids.view(-1), r.view(-1))
    rew_row = ret[ids] * logged                                  # `reward` row of (t, lane)
    m = logged.double()
    n_l = m.sum(dim=0)                                           # logger.pointer of the lane
    t_idx = torch.arange(steps, device=rows.device)[:, None].double()
    pos = torch.cumsum(m, dim=0) - m                             # index of the step among the lane's logged steps
    second = m * (pos >= torch.floor(n_l / 2)[None, :]).double()
    ok = n_l > 0
    lane = dict(logged_steps=n_l, viol_rate=(viol * m).sum(0) / n_l.clamp(min=1), mean_max_ineq=(ineq.double() * m).sum(0) / n_l.clamp(min=1),
                mean_return_per_step=rew_row.sum(0) / n_l.clamp(min=1),
                mean_return_second_half=(rew_row * second).sum(0) / second.sum(0).clamp(min=1))
    seed_level = {kk: float(v[ok].mean()) for kk, v in lane.items()}
    seed_level["max_nu"] = float(tr.agent.nju.weight.detach().abs().max())
    # curves over the run (all lanes, per chunk of vector steps): violation fraction, mean `reward` row of the logged steps
    T = steps // chunk * chunk
    vm = (viol * m)[:T].view(-1, chunk, n).sum(dim=(1, 2)) / m[:T].view(-1, chunk, n).sum(dim=(1, 2)).clamp(min=1)
    rm = rew_row[:T].view(-1, chunk, n).sum(dim=(1, 2)) / m[:T].view(-1, chunk, n).sum(dim=(1, 2)).clamp(min=1)
    del t_idx
    return seed_level, dict(viol_rate=vm.cpu().tolist(), reward_row=rm.cpu().tolist())


def run_mode(mode, seeds, steps, device, lanes=LANES):
    """``lanes``: env lanes per run (4096 = the bench; smaller counts interpolate towards the reference's single env: the
    `lanes` sweep of round 5 shows how the statistics move with the number of independent histories an update samples from)."""
    out, curves = [], []
    LANES = lanes                                                 # noqa: N806  (shadows the module default below)
    for seed in range(seeds):
        extra = dict(batch_size=256 * LANES) if mode == "large_batch" else {}
        t0 = time.perf_counter()
        tr = bench.make_trainer(LANES, device, steps, capacity=steps, workload="cart_ddpg", torch_seed=123 + seed,
                                seed=7000 + seed, **extra)
        tr.vec.reset()
        tr.run_steps(steps)
        tr._harvest(final=True)
        torch.cuda.synchronize()
        s, cv = lane_statistics(tr, steps)
        s["device_viol_rate"] = float(tr.viol_rate)
        s["seconds"] = time.perf_counter() - t0
        out.append(s)
        curves.append(cv)
        print(mode, "seed", seed, json.dumps(s), file=sys.stderr, flush=True)
        del tr, s, cv
        import gc
        gc.collect()                                              # (the 1.5 GB ring of the finished run goes before the next one comes)
        torch.cuda.empty_cache()
    keys = [k for k in out[0] if k != "seconds"]
    arr = {k: np.array([o[k] for o in out]) for k in keys}
    return dict(mode=mode, seeds=seeds, updates=steps, lanes=LANES,
                mean={k: float(v.mean()) for k, v in arr.items()},
                se={k: float(v.std(ddof=1) / np.sqrt(len(v))) if len(v) > 1 else None for k, v in arr.items()},
                per_seed=out, curve_chunk=100,
                curves_mean={k: np.mean([c[k] for c in curves], axis=0).tolist() for k in curves[0]})


def reference_row():
    g = np.load(os.path.join(ROOT, "tests", "golden", "training_stats_ddpg_cart.npz"))
    ref = g["stats"]
    cols = [str(c) for c in g["columns"]]
    return dict(seeds=int(len(ref)), updates=int(g["steps"]), mean=dict(zip(cols, ref.mean(0).tolist())),
                se=dict(zip(cols, (ref.std(0, ddof=1) / np.sqrt(len(ref))).tolist())))


def sweep(lane_counts, seeds, steps, device):
    """Reference cadence at several lane counts -> gpurun_out/cadence_lanes.json (means, standard errors, z against the
    reference's runs)."""
    ref = reference_row()
    rows = []
    for n in lane_counts:
        m = run_mode("reference_cadence", seeds, steps, device, lanes=n)
        m.pop("per_seed")
        m.pop("curves_mean")
        m["z_vs_reference"] = {k: (m["mean"][k] - ref["mean"][k]) / float(np.sqrt(m["se"][k] ** 2 + ref["se"][k] ** 2))
                               for k in ("viol_rate", "mean_max_ineq", "mean_return_per_step", "mean_return_second_half") if k in ref["mean"]}
        rows.append(m)
        print(n, json.dumps(m["mean"]), json.dumps(m["z_vs_reference"]), flush=True)
    import provenance
    with open(os.path.join(ROOT, "gpurun_out", "cadence_lanes.json"), "w") as f:
        json.dump(dict(reference=ref, sweep=rows, **provenance.stamp()), f, indent=1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "lanes":             # python tools/cadence_learning.py lanes 1,16,256 128 [steps]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        sweep([int(x) for x in sys.argv[2].split(",")], int(sys.argv[3]) if len(sys.argv) > 3 else 128,
              int(sys.argv[4]) if len(sys.argv) > 4 else 3000, torch.device("cuda"))
        return
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    seeds_lb = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, seeds // 4)
    device = torch.device("cuda")
    import provenance
    res = dict(reference=reference_row(), modes=[run_mode(m, n, steps, device)
                                                 for m, n in (("reference_cadence", seeds), ("large_batch", seeds_lb)) if n > 0],
               **provenance.stamp())
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "cadence_learning.json"), "w") as f:
        json.dump(res, f, indent=1)
    for m in res["modes"]:
        print(m["mode"], json.dumps(m["mean"]), json.dumps(m["se"]))
    print("reference", json.dumps(res["reference"]["mean"]))


if __name__ == "__main__":
    main()
