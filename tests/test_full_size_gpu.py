"""Size-independent properties at BASELINE.json's full sizes for the configs that have no oracle run at that size:
config 3 (SpringPendulum-v0, RPOSAC, 4096 lanes) and config 5 (EVOPF-v0, RPODDPG, 1024 lanes) -- the analogues of
tests/test_kernels_gpu.py::test_cart_full_size_properties, through the shipped trainers with bench.py's exact
hyper-parameters (scripts/pen_exp_sac.py, scripts/evopf_exp.py) and the hipGraph windows of the bench.

Checked over 80 whole iterations (rollout -> projection -> step -> scatter -> sample -> update):
  * the replay ring holds exactly the transitions that were stepped: row (t, lane) continues row (t - 1, lane) -- its
    state equals the previous next_state -- unless the lane's episode ended there, in which case it restarts inside the
    env's reset box; rows are finite and the ring is full up to the step counter and untouched beyond it;
  * every stored action satisfies the equality constraints to the solver's tolerance (the stored eq_viol column is what
    env.step recomputes from (state, action): rpo/env/.../pendulum.py:128, evopf.py:366), and the stored inequality
    violations are the distances of the stored action recomputed by the constraint kernel;
  * the device-side statistics (violation counter, reward sums, episode counts) equal what the ring rows give;
  * lanes are a pure function of (seed, env id): one rollout of all lanes == two half-size rollouts with env_id_base
    offsets (the rank layout of the data-parallel run), bit for bit;
  * replayed hipGraph windows == eager launches, bit for bit, at this size.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DEV = torch.device("cuda")
T = 80            # 5 windows of 16 iterations: three eager passes, the capture, one replay


def run(workload, n, use_graph, iters=T, capacity=96):
    import bench
    os.environ["RPO_VERBOSE"] = "0"
    tr = bench.make_trainer(n, DEV, 10 ** 9, capacity=capacity, workload=workload, use_graph=use_graph)
    tr.vec.reset()
    first = tr.vec.internal.clone()
    tr.run_steps(iters)
    tr._harvest(final=True)
    torch.cuda.synchronize()
    return tr, first


def ring(tr, iters=T):
    n = tr.n_local
    rows = tr.buffer.rows[: iters * n].view(iters, n, -1)
    c = tr.kernels.cols
    return rows, {k: rows[:, :, lo:hi] for k, (lo, hi) in c.items()}


def check_chain(cols, reset_ok):
    """state[t] == next_state[t - 1] for lanes that did not finish at t - 1; restarted lanes are inside the reset box."""
    s, ns, done = cols["state"], cols["next_state"], cols["done"][:, :, 0]
    cont = done[:-1] == 0
    same = (s[1:] == ns[:-1]).all(dim=2)
    assert bool(same[cont].all()), "a continuing lane's state is not the previous next_state"
    restarted = s[1:][~cont]
    if restarted.numel():
        assert bool(reset_ok(restarted).all()), "a restarted lane is outside the reset box"
    return int((~cont).sum())


def test_pendulum_sac_4096_lanes_properties():
    from rpo_amd import ops
    n = 4096
    g, first = run("pen_sac", n, use_graph=True)
    assert g.fused is not None and g._split_state() is not None and any(
        k[0] == "cycle" and e["graph"] is not None for k, e in g._graphs.entries.items() if isinstance(k, tuple))
    rows, cols = ring(g)
    assert bool(torch.isfinite(rows).all()) and float(g.buffer.rows[T * n:].abs().max()) == 0.0
    assert int(g.vec.ctrl[0]) == T
    # trajectory continuity; pendulum obs = (cos, sin, theta_dot, l, l_dot): reset box of pendulum.py:131-133
    def reset_ok(s):
        th = torch.atan2(s[:, 1], s[:, 0])
        return (th.abs() <= np.pi / 12 + 1e-6) & (s[:, 2].abs() <= 1 + 1e-6) & (s[:, 3] >= 0.95 - 1e-6) & \
               (s[:, 3] <= 1.05 + 1e-6) & (s[:, 4].abs() <= 0.05 + 1e-6)
    ended = check_chain(cols, reset_ok)
    assert ended > 0                                             # some lanes did finish within 80 steps
    # equalities hold for every stored action; stored violations == the constraint kernel on (state, action)
    assert float(cols["eq_viol"].abs().max()) < 2e-5
    flat_s = cols["state"].reshape(-1, 5).contiguous()
    flat_a = cols["action"].reshape(-1, 2).contiguous()
    eq = torch.zeros(flat_a.shape[0], device=DEV)
    ineq = torch.zeros(flat_a.shape[0], device=DEV)
    g.kernels.resid(flat_s, flat_a, eq, ineq)
    np.testing.assert_allclose(eq.cpu().numpy(), cols["eq_viol"].reshape(-1).cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(ineq.clamp(min=0).cpu().numpy(), cols["ineq_viol"].reshape(-1).cpu().numpy(), atol=2e-5)
    # projection: |a|^2 <= 32 up to what 10 GRG steps of lr 2e-3 leave; never a NaN, and mostly feasible
    assert float((cols["ineq_viol"] <= 1e-3).float().mean()) > 0.95
    # statistics == ring
    S = ops.STAT
    st = ops.reduce_stats(g.vec.stats[:T]).cpu().numpy()
    viol = (torch.maximum(cols["ineq_viol"].max(dim=2).values, cols["eq_viol"].abs().max(dim=2).values) > 1e-3)
    np.testing.assert_array_equal(st[:, S["viol_count"]], viol.sum(dim=1).cpu().numpy().astype(np.float32))
    np.testing.assert_allclose(st[:, S["reward_sum"]], cols["reward"][:, :, 0].sum(dim=1).cpu().numpy(), rtol=1e-5)
    np.testing.assert_array_equal(st[:, S["episodes"]], cols["done"][:, :, 0].sum(dim=1).cpu().numpy())
    assert abs(g.viol_rate - float(viol.double().mean())) < 1e-9 and g.env_steps == T * n
    # hipGraph windows == eager launches, at this size
    e, _ = run("pen_sac", n, use_graph=False)
    assert torch.equal(g.buffer.rows, e.buffer.rows) and torch.equal(g.agent.flat.data, e.agent.flat.data)
    assert torch.equal(g.vec.internal, e.vec.internal) and torch.equal(g.agent.nju.weight, e.agent.nju.weight)
    # rank layout: the first vector step of lanes [0, n) == two launches of n / 2 lanes with env_id_base offsets
    from rpo_amd.env.vec import VecEnv
    k = g.kernels
    whole = VecEnv(k, n, DEV, seed=g.seed, env_id_base=0, max_episode_steps=200)
    whole.reset()
    assert torch.equal(whole.internal, first)
    halves = [VecEnv(k, n // 2, DEV, seed=g.seed, env_id_base=b, max_episode_steps=200) for b in (0, n // 2)]
    for h in halves:
        h.reset()
    assert torch.equal(torch.cat([h.internal for h in halves]), first)
    ap = torch.linspace(-7, 7, n, device=DEV)
    for v, lo in ((whole, 0), (halves[0], 0), (halves[1], n // 2)):
        k.act_project(v.obs, ap[lo:lo + v.n].contiguous(), None, v.action, None, ops.NOISE_PHILOX, 0.5, 0.5, 0.0, -6.0, 6.0,
                      10, 2e-3, 1e-5, 0.0, v.seed, v.env_id_base, v.ctrl, None)
        v.step(v.action)
    assert torch.equal(whole.internal, torch.cat([h.internal for h in halves]))
    assert torch.equal(whole.action, torch.cat([h.action for h in halves]))


def test_evopf_ddpg_1024_lanes_properties():
    from rpo_amd import ops
    n = 1024
    g, first = run("evopf_ddpg", n, use_graph=True)
    rows, cols = ring(g)
    assert bool(torch.isfinite(rows).all()) and float(g.buffer.rows[T * n:].abs().max()) == 0.0
    assert int(g.vec.ctrl[0]) == T
    # 24-hour episodes: every lane finishes exactly at step 23 (0-based) and restarts with fresh batteries (soc 0.2)
    done = cols["done"][:, :, 0]
    ends = [23, 47, 71]
    assert all(bool((done[t] == 1).all()) for t in ends) and float(done.sum()) == 3.0 * n
    s, ns = cols["state"], cols["next_state"]
    cont = torch.ones(T - 1, dtype=torch.bool, device=DEV)
    cont[ends] = False
    assert bool((s[1:][cont] == ns[:-1][cont]).all())
    assert float((s[24, :, 28:33] - 0.2).abs().max()) < 1e-6
    # Newton power flow: the 28 equalities of every stored action hold to the solver's tolerance (GRG drift included:
    # DESIGN 4b E1 reports <= 3e-3 after 10 projection steps)
    assert float(cols["eq_viol"].abs().max()) < 5e-3
    assert float(cols["eq_viol"].abs().mean()) < 2e-4
    # stored violations == the constraint kernel on (state, action)
    flat_s = s.reshape(-1, s.shape[2]).contiguous()
    flat_a = cols["action"].reshape(-1, cols["action"].shape[2]).contiguous()
    m = flat_a.shape[0]
    eq = torch.zeros(m, 28, device=DEV)
    ineq = torch.zeros(m, 58, device=DEV)
    g.kernels.resid(flat_s, flat_a, eq, ineq)
    np.testing.assert_allclose(eq.cpu().numpy(), cols["eq_viol"].reshape(m, 28).cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(ineq.clamp(min=0).cpu().numpy(), cols["ineq_viol"].reshape(m, 58).cpu().numpy(), atol=2e-5)
    # statistics == ring
    S = ops.STAT
    st = ops.reduce_stats(g.vec.stats[:T]).cpu().numpy()
    viol = (torch.maximum(cols["ineq_viol"].max(dim=2).values, cols["eq_viol"].abs().max(dim=2).values) > 1e-3)
    np.testing.assert_array_equal(st[:, S["viol_count"]], viol.sum(dim=1).cpu().numpy().astype(np.float32))
    np.testing.assert_allclose(st[:, S["reward_sum"]], cols["reward"][:, :, 0].sum(dim=1).cpu().numpy(), rtol=2e-5)
    np.testing.assert_array_equal(st[:, S["episodes"]], done.sum(dim=1).cpu().numpy())
    # hipGraph replay == eager at this size
    e, _ = run("evopf_ddpg", n, use_graph=False)
    assert torch.equal(g.buffer.rows, e.buffer.rows) and torch.equal(g.agent.flat.data, e.agent.flat.data)
    # rank layout: lanes are a pure function of (seed, env id)
    from rpo_amd.env.vec import VecEnv
    k = g.kernels
    whole = VecEnv(k, n, DEV, seed=g.seed, env_id_base=0)
    whole.reset()
    assert torch.equal(whole.internal, first)
    halves = [VecEnv(k, n // 2, DEV, seed=g.seed, env_id_base=b) for b in (0, n // 2)]
    for h in halves:
        h.reset()
    assert torch.equal(torch.cat([h.internal for h in halves]), first)
    for v in (whole, halves[0], halves[1]):
        k.act_project(v.obs, None, None, v.action, None, ops.NOISE_UNIFORM, 0.0, 0.0, 0.0, 0.0, 0.0, 10, 1e-4, 1e-5, 0.0,
                      v.seed, v.env_id_base, v.ctrl, None)
        v.step(v.action)
    assert torch.equal(whole.internal, torch.cat([h.internal for h in halves]))
    assert torch.equal(whole.action, torch.cat([h.action for h in halves]))


def test_cart_sac_4096_lanes_ridden_windows_equal_eager():
    """Config 4's algorithm at its per-GPU size (CartSafe-v0 RPOSAC, 4096 lanes, scripts/cart_exp_sac.py): the hipGraph
    windows in which the rollout rides on the critic update's launches (rpo_split_*_ride) leave the same ring, lanes and
    parameters as eager launches in the serial order; the ring is a chain of the stepped transitions; the device-side
    statistics equal what the ring rows give."""
    from rpo_amd import ops
    n = 4096
    g, first = run("cart_sac", n, use_graph=True)
    assert g._ride_ok(True) and any(len(k) == 4 and k[3] == "ride" and e["graph"] is not None
                                    for k, e in g._graphs.entries.items() if isinstance(k, tuple))
    e, _ = run("cart_sac", n, use_graph=False)
    assert torch.equal(g.buffer.rows, e.buffer.rows) and torch.equal(g.agent.flat.data, e.agent.flat.data)
    assert torch.equal(g.vec.internal, e.vec.internal) and torch.equal(g.agent.nju.weight, e.agent.nju.weight)
    assert torch.equal(g.vec.ep_len, e.vec.ep_len) and torch.equal(g.vec.ep_count, e.vec.ep_count)
    rows, cols = ring(g)
    assert bool(torch.isfinite(rows).all()) and float(g.buffer.rows[T * n:].abs().max()) == 0.0
    assert int(g.vec.ctrl[0]) == T and int(g._uctrl[0]) == T + 1

    def reset_ok(s):                                             # cartpole.py reset: every state entry uniform in (-0.05, 0.05)
        return (s.abs() <= 0.05 + 1e-6).all(dim=1)
    assert check_chain(cols, reset_ok) > 0
    # the equality holds for every stored action (Complete + Proj), violations == the constraint kernel on (state, action)
    assert float(cols["eq_viol"].abs().max()) < 2e-5
    S = ops.STAT
    st = ops.reduce_stats(g.vec.stats[:T]).cpu().numpy()
    viol = (torch.maximum(cols["ineq_viol"].max(dim=2).values, cols["eq_viol"].abs().max(dim=2).values) > 1e-3)
    np.testing.assert_array_equal(st[:, S["viol_count"]], viol.sum(dim=1).cpu().numpy().astype(np.float32))
    np.testing.assert_array_equal(st[:, S["episodes"]], cols["done"][:, :, 0].sum(dim=1).cpu().numpy())
    np.testing.assert_allclose(st[:, S["reward_sum"]], cols["reward"][:, :, 0].sum(dim=1).cpu().numpy(), rtol=1e-6)
    assert abs(g.viol_rate - float(viol.double().mean())) < 1e-9 and g.env_steps == T * n


def test_cart_ddpg_4096_lanes_properties(monkeypatch):
    """Config 2, the headline (CartSafe-v0 RPODDPG, 4096 lanes, scripts/cart_exp.py with its shared state embedding): the one
    configuration where the rollout does NOT ride and the policy front takes pol_a itself.  80 iterations through the
    shipped trainer with the bench's hyper-parameters: ring chain, statistics == ring, hipGraph windows == eager launches,
    fused front launches == separate launches (RPO_SCHEDULE=front=0), bit for bit."""
    from rpo_amd import ops
    n = 4096
    g, first = run("cart_ddpg", n, use_graph=True)
    assert g.fused is not None and g._split_state() is not None and g.agent.flat.sizes[1] > 0      # shared embedding
    assert not g._ride_ok(True) and g._front_ok()
    assert any(k[0] == "cycle" and e["graph"] is not None for k, e in g._graphs.entries.items() if isinstance(k, tuple))
    rows, cols = ring(g)
    assert bool(torch.isfinite(rows).all()) and float(g.buffer.rows[T * n:].abs().max()) == 0.0
    assert int(g.vec.ctrl[0]) == T

    def reset_ok(s):                                             # cartpole.py:233: every state entry uniform in (-0.05, 0.05)
        return (s.abs() <= 0.05 + 1e-6).all(dim=1)
    assert check_chain(cols, reset_ok) > 0
    assert float(cols["eq_viol"].abs().max()) < 2e-5             # Complete + Proj keep the equality (cartpole.py:369-376)
    S = ops.STAT
    st = ops.reduce_stats(g.vec.stats[:T]).cpu().numpy()
    viol = (torch.maximum(cols["ineq_viol"].max(dim=2).values, cols["eq_viol"].abs().max(dim=2).values) > 1e-3)
    np.testing.assert_array_equal(st[:, S["viol_count"]], viol.sum(dim=1).cpu().numpy().astype(np.float32))
    np.testing.assert_array_equal(st[:, S["episodes"]], cols["done"][:, :, 0].sum(dim=1).cpu().numpy())
    np.testing.assert_allclose(st[:, S["reward_sum"]], cols["reward"][:, :, 0].sum(dim=1).cpu().numpy(), rtol=1e-6)
    assert abs(g.viol_rate - float(viol.double().mean())) < 1e-9 and g.env_steps == T * n
    e, _ = run("cart_ddpg", n, use_graph=False)
    monkeypatch.setenv("RPO_SCHEDULE", "front=0")
    s, _ = run("cart_ddpg", n, use_graph=True)
    assert not s._front_ok()
    for other in (e, s):
        assert torch.equal(g.buffer.rows, other.buffer.rows) and torch.equal(g.agent.flat.data, other.agent.flat.data)
        assert torch.equal(g.vec.internal, other.vec.internal) and torch.equal(g.agent.nju.weight, other.agent.nju.weight)
        assert torch.equal(g.agent.critic_target_flat, other.agent.critic_target_flat)
        assert torch.equal(g.agent.critic_optim.exp_avg_sq, other.agent.critic_optim.exp_avg_sq)


def _forced_rccl_worker(rank, port, out_dir, n, iters):
    """Config 4's per-rank workload (CartSafe-v0 RPOSAC, 4096 lanes per GPU) on the data-parallel code path over RCCL with a
    forced one-rank group: the gradient all-reduces are captured inside the ridden 16-iteration windows."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RPO_VERBOSE="0", RPO_SCHEDULE="force_dist=1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import bench
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    tr = bench.make_trainer(n, torch.device("cuda", 0), 10 ** 9, capacity=96, workload="cart_sac", use_graph=True)
    assert tr.dist.on and tr.dist.in_graph and tr._ride_ok(True)
    tr.vec.reset()
    tr.run_steps(iters)
    tr._harvest(final=True)
    torch.cuda.synchronize()
    captured = [k for k, e in tr._graphs.entries.items() if e["graph"] is not None]
    assert ("cycle", 16, True, "ride") in captured, captured
    torch.save(dict(flat=tr.agent.flat.data.cpu(), nju=tr.agent.nju.weight.data.cpu(), state=tr.vec.internal.cpu(),
                    rows=tr.buffer.rows[:16 * n].cpu(), uctrl=int(tr._uctrl[0]), env_steps=float(tr.env_steps)),
               os.path.join(out_dir, "forced.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_cart_sac_4096_lanes_forced_rccl_in_ridden_windows(tmp_path):
    """SCALE config (CartSafe-v0 RPOSAC, 4096 lanes per rank, RCCL gradient all-reduce): what ONE GPU can check -- the
    data-parallel iteration with its collectives captured inside the ridden hipGraph windows (schedule force_dist=1: a one-rank
    RCCL group, the mean over ranks is the identity) leaves the same parameters, lanes and ring as the plain single-process
    run, bit for bit; the inf-norm then comes from rpo_absmax_slots behind the all-reduce instead of the backward kernels."""
    import socket
    import time
    import torch.multiprocessing as mp
    n = 4096
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.spawn(_forced_rccl_worker, args=(port, str(tmp_path), n, T), nprocs=1, join=False)
    deadline = time.time() + 300
    while not ctx.join(timeout=5):
        if time.time() > deadline:
            for proc in ctx.processes:
                if proc.is_alive():
                    proc.kill()
            pytest.fail("the forced-RCCL rank did not finish")
    r = torch.load(os.path.join(str(tmp_path), "forced.pt"), weights_only=False)
    g, _ = run("cart_sac", n, use_graph=True)
    assert torch.equal(g.agent.flat.data.cpu(), r["flat"]) and torch.equal(g.agent.nju.weight.data.cpu(), r["nju"])
    assert torch.equal(g.vec.internal.cpu(), r["state"]) and torch.equal(g.buffer.rows[:16 * n].cpu(), r["rows"])
    assert int(g._uctrl[0]) == r["uctrl"] and g.env_steps == r["env_steps"] == T * n


@pytest.mark.parametrize("workload", ["cart_ddpg", "pen_sac"])
def test_lost_producer_fails_loudly_and_leaves_the_device_usable(workload, monkeypatch):
    """The in-launch hand-overs of the fused fronts wait with a bound (kNsSpinMax / kPmSpinMax polls).  A test-only debug
    bit makes ONE producer workgroup withhold its hand-over: the consumers give up, raise the flag word, and the trainer
    turns that into a RuntimeError -- at the next statistics harvest, at save(), and (asynchronously) within about one graph
    window.  The launch itself completes, leaves its hand-over words clean, and the device runs the next trainer normally."""
    import bench
    from rpo_amd import ops
    os.environ["RPO_VERBOSE"] = "0"
    monkeypatch.setenv("RPO_GRAPH_CYCLE", "4")
    tr = bench.make_trainer(256, DEV, 10 ** 9, capacity=64, workload=workload, use_graph=False)
    tr.vec.reset()
    tr.run_steps(8)
    tr._harvest(final=True)                                      # healthy: no flag
    su = tr._split_state()
    assert su is not None and tr._front_ok() and (workload != "pen_sac" or tr._pfront)
    su.set(debug=1)
    tr.run_steps(1)
    torch.cuda.synchronize()                                     # the launch with the lost producer completes
    front, proj = tr._handover_flags()
    assert int((front if workload == "cart_ddpg" else proj)[0]) == 1
    with pytest.raises(RuntimeError, match="gave up"):
        tr._harvest(final=True)
    with pytest.raises(RuntimeError, match="gave up"):
        tr.save(replay=False)
    # every hand-over word except the flag is back at zero, the granule epoch moved on, the readers' count is zero
    sync = su._held["tile_sync"].clone()
    sync[-32] = 0
    assert int(sync.abs().sum()) == 0
    if workload == "pen_sac":
        ws = su._held["proj_ws"]
        assert int(ws[ops.PROJ_WS_GAVE_UP + 1]) == 0 and int(ws[ops.PROJ_WS_GAVE_UP - 1]) == 9
    # the asynchronous poll of the graph windows raises too
    g = bench.make_trainer(256, DEV, 10 ** 9, capacity=64, workload=workload, use_graph=True)
    g.vec.reset()
    g.run_steps(40)
    g._split_state().set(debug=1)
    g._graphs.entries.clear()                                    # (the struct is read at capture: capture again with the bit)
    with pytest.raises(RuntimeError, match="gave up"):
        for _ in range(40):
            g.run_steps(4)
            torch.cuda.synchronize()
    # ... and the device is fine: a fresh trainer runs, finite and flag-free
    h = bench.make_trainer(256, DEV, 10 ** 9, capacity=64, workload=workload, use_graph=True)
    h.vec.reset()
    h.run_steps(40)
    h._harvest(final=True)
    assert bool(torch.isfinite(h.vec.internal).all()) and bool(torch.isfinite(h.agent.flat.data).all())


def test_eight_way_rank_layout_is_invisible_in_the_lanes():
    """Config 4's rank layout on ONE GPU (rank r owns env ids [4096 r / 8 ...)): eight 512-lane launches of the fused
    rollout kernel with env_id_base = 512 r leave, over 260 vector steps (the 200-step TimeLimit ends every first episode,
    terminations restart lanes in between), exactly the lanes, actions, episode bookkeeping and ring rows of ONE 4096-lane
    launch per step -- CartSafe-v0 with RPOSAC's squashed-Gaussian actor (scripts/cart_exp_sac.py), every random draw keyed by
    (seed, global env id, step).  Next to the 2-way checks above."""
    import bench
    from rpo_amd import ops
    from rpo_amd.env.vec import VecEnv
    os.environ["RPO_VERBOSE"] = "0"
    n, R, steps, cap = 4096, 8, 260, 4
    tr = bench.make_trainer(n, DEV, 10 ** 9, capacity=4, workload="cart_sac", use_graph=False)
    k, f = tr.kernels, tr.fused
    scale, base = tr._box_affine

    def roll(v, rows):
        k.rollout(f.descs["actor"], True, scale, base, v.internal, None, v.action, v.ep_len, v.ep_ret, v.ep_count, rows, cap,
                  v.stats, v.ctrl, ops.NOISE_NONE, tr.eps_start, tr.eps, tr.decay_value, tr._box_lo, tr._box_hi, tr.max_steps,
                  tr.corr_lr, tr.corr_eps, tr.corr_momentum, 200, True, 1e-3, v.seed, v.env_id_base)
    whole = VecEnv(k, n, DEV, seed=tr.seed, env_id_base=0, max_episode_steps=200)
    parts = [VecEnv(k, n // R, DEV, seed=tr.seed, env_id_base=r * (n // R), max_episode_steps=200) for r in range(R)]
    rows_w = torch.zeros(cap * n, k.ring_floats, device=DEV)
    rows_p = [torch.zeros(cap * (n // R), k.ring_floats, device=DEV) for _ in range(R)]
    for v in [whole] + parts:
        v.reset()
    assert torch.equal(whole.internal, torch.cat([p.internal for p in parts]))
    restarts = 0
    for t in range(steps):
        roll(whole, rows_w)
        for p, rp in zip(parts, rows_p):
            roll(p, rp)
        if t % 20 == 19 or t == steps - 1:
            assert torch.equal(whole.internal, torch.cat([p.internal for p in parts])), t
            assert torch.equal(whole.action, torch.cat([p.action for p in parts])), t
            assert torch.equal(whole.ep_len, torch.cat([p.ep_len for p in parts]))
            assert torch.equal(whole.ep_count, torch.cat([p.ep_count for p in parts]))
            assert torch.equal(whole.ep_ret, torch.cat([p.ep_ret for p in parts]))
            slot = t % cap                                         # the ring slot of this step: [slot n, (slot + 1) n)
            mine = rows_w[slot * n:(slot + 1) * n]
            theirs = torch.cat([rp[slot * (n // R):(slot + 1) * (n // R)] for rp in rows_p])
            assert torch.equal(mine, theirs), t
    restarts = int(whole.ep_count.sum())
    assert restarts >= n and int(whole.ctrl[0]) == steps and all(int(p.ctrl[0]) == steps for p in parts)
    # per-step statistics: the shards' counters add up to the whole launch's (float sums to rounding)
    S = ops.STAT
    sw = ops.reduce_stats(whole.stats[:steps]).cpu().numpy()
    sp = sum(ops.reduce_stats(p.stats[:steps]).cpu().numpy() for p in parts)
    for key in ("episodes", "viol_count", "length_sum", "terminated", "proj_iters"):
        np.testing.assert_array_equal(sw[:, S[key]], sp[:, S[key]])
    np.testing.assert_allclose(sw[:, S["reward_sum"]], sp[:, S["reward_sum"]], rtol=2e-6)
