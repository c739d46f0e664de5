"""Host-side pieces of the reference's surface that are not kernels: Logger, gym stand-in, checkpoints (CPU)."""
import os
import pickle

import numpy as np
import pytest
import torch

import oracle_backend as ob
from rpo_amd import gym_shim
from rpo_amd.utils.logger import Logger
from test_train_step_golden import build_trainer


def test_logger_matches_reference_semantics(tmp_path):
    """rpo/utils/logger.py:5-25: preallocated float64 tracks, StopIteration when full, pickles itself with a timestamp."""
    lg = Logger(("epoch", "reward", "max_ineq", "max_eq"), epochs=3, times=2, name="cart_ddpg")
    assert lg.tracker["reward"].shape == (6,) and lg.tracker["reward"].dtype == np.float64 and lg.pointer == 0
    for i in range(6):
        lg.add(epoch=i, reward=2.0 * i, max_ineq=0.1, max_eq=0.0)
    with pytest.raises(StopIteration):
        lg.add(epoch=6, reward=0, max_ineq=0, max_eq=0)
    assert lg.pointer == 6 and lg.tracker["reward"][5] == 10.0
    lg.save(os.path.join(tmp_path, "cart_ddpg"))
    files = os.listdir(tmp_path)
    assert len(files) == 1 and files[0].startswith("cart_ddpg_")
    with open(os.path.join(tmp_path, files[0]), "rb") as f:
        back = pickle.load(f)
    np.testing.assert_array_equal(back.tracker["epoch"], np.arange(6))
    lg2 = Logger(("a",), epochs=4, times=1)
    lg2.add_rows(a=np.arange(3))
    with pytest.raises(StopIteration):
        lg2.add_rows(a=np.arange(2))


def test_logger_pickle_loads_without_rpo_amd(tmp_path):
    """On-disk compatibility (rpo/utils/logger.py:23-25): the file names ``rpo.utils.logger.Logger`` and carries only
    the reference's attributes, so a process that can import a package called ``rpo`` with a plain ``Logger`` class --
    the reference's layout -- and NOT ``rpo_amd`` reads it back.  The stand-in package below is written for this
    test (a bare class with the same name); the interpreter runs isolated from this repository."""
    import pickletools
    import subprocess
    import sys
    lg = Logger(("epoch", "reward", "max_ineq", "max_eq"), epochs=4, times=1, name="cart_ddpg")
    for i in range(3):
        lg.add(epoch=i, reward=1.5 * i, max_ineq=0.25, max_eq=1e-7)
    out = tmp_path / "logs"
    out.mkdir()
    lg.save(str(out / "cart_ddpg"))
    path = str(out / os.listdir(out)[0])
    with open(path, "rb") as f:
        raw = f.read()
    names = [arg for op, arg, _ in pickletools.genops(raw) if op.name in ("GLOBAL", "STACK_GLOBAL", "SHORT_BINUNICODE",
                                                                          "BINUNICODE", "UNICODE")]
    assert "rpo.utils.logger" in names and not any("rpo_amd" in str(n) for n in names if n)
    pkg = tmp_path / "site" / "rpo" / "utils"
    pkg.mkdir(parents=True)
    (tmp_path / "site" / "rpo" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("")
    (pkg / "logger.py").write_text("class Logger(object):\n    pass\n")
    code = ("import sys, pickle\n"
            "sys.path[:] = [p for p in sys.path if 'repo' not in p]\n"
            "sys.path.insert(0, %r)\n"
            "lg = pickle.load(open(%r, 'rb'))\n"
            "assert 'rpo_amd' not in sys.modules and type(lg).__module__ == 'rpo.utils.logger'\n"
            "assert lg.pointer == 3 and lg.epochs == 4 and lg.times == 1 and lg.name == 'cart_ddpg'\n"
            "assert sorted(lg.__dict__) == ['epochs', 'name', 'pointer', 'times', 'tracker']\n"
            "assert lg.tracker['reward'][2] == 3.0 and lg.tracker['reward'].dtype.name == 'float64'\n"
            "print('ok')\n") % (str(tmp_path / "site"), path)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert res.returncode == 0 and res.stdout.strip() == "ok", res.stderr


def test_gym_stand_in_time_limit_and_spaces():
    env = gym_shim.make("CartSafe-v0") if "CartSafe-v0" in gym_shim._REGISTRY else None
    import rpo_amd.env  # noqa: F401  registers the ids
    env = env or gym_shim.make("CartSafe-v0")
    assert env._max_episode_steps == 200 and env.unwrapped is env.env
    with pytest.raises(AttributeError):
        env._no_such_private                                     # keeps copy.deepcopy from recursing
    box = gym_shim.Box(np.array([-1, -2], dtype=np.float32), np.array([1, 2], dtype=np.float32))
    assert box.contains(np.array([0.5, -2.0], dtype=np.float32))
    assert not box.contains(np.array([0.5, -2.1], dtype=np.float32))
    assert not box.contains(np.array([0.5, 0.0]))                # float64 is not castable to float32 (gym 0.19)
    assert not box.contains(np.array([0.5], dtype=np.float32))


def test_checkpoint_roundtrip(tmp_path):
    """save()/load() restore parameters, targets, multipliers and optimiser moments (the reference's save_model stores
    parameter generators and cannot be loaded back, agent/ddpg_pa.py:92-99)."""
    torch.set_num_threads(1)
    torch.manual_seed(1)
    a = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=4)
    a.work_dir = str(tmp_path)
    a.vec.reset()
    a.run_steps(6)
    a.save()
    torch.manual_seed(2)
    b = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=4)
    b.work_dir = str(tmp_path)
    assert not torch.equal(a.agent.flat.data, b.agent.flat.data)
    b.load()
    assert torch.equal(a.agent.flat.data, b.agent.flat.data)
    assert torch.equal(a.agent.nju.weight, b.agent.nju.weight)
    assert torch.equal(a.agent.critic_optim.exp_avg, b.agent.critic_optim.exp_avg)
    assert int(b.agent.actor_optim.step_dev[0]) == int(a.agent.actor_optim.step_dev[0]) == 1
    # exact resume: the targets are restored as they were (not hard-updated like agent/ddpg_pa.py:96-99 does)
    assert torch.equal(b.agent.critic_target_flat, a.agent.critic_target_flat)
    assert torch.equal(b.agent.actor_target_flat, a.agent.actor_target_flat)


def test_harvest_bookkeeping_matches_per_row_loop(capsys):
    """Trainer._harvest turns statistics rows into Logger rows with whole-array operations; compare with the per-row
    bookkeeping of the reference loop (rpo_ddpg.py:120-137: every step's max violations are logged with the return of
    the episode that step belongs to, once it has finished) over several harvests, a logger that fills up, and rows
    left pending across harvests."""
    from rpo_amd import ops as hip_ops
    S = hip_ops.STAT
    torch.manual_seed(0)
    tr = build_trainer("ddpg", "cart", ob, torch.device("cpu"), num_envs=1)
    rng = np.random.default_rng(3)
    cap = tr.vec.stats.shape[0]
    total = 700
    stats = np.zeros((total, hip_ops.STATS_SUB, hip_ops.STATS_LEN), dtype=np.float32)
    stats[:, 0, S["max_ineq_sum"]] = rng.random(total)
    stats[:, 0, S["max_eq_sum"]] = rng.random(total) * 0.1
    ends = np.sort(rng.choice(np.arange(5, total - 40), size=9, replace=False))
    stats[ends, 0, S["episodes"]] = 1
    stats[ends, 0, S["return_sum"]] = rng.random(9) * 100
    stats[ends, 0, S["length_sum"]] = rng.integers(1, 200, 9)
    tr.logger = Logger(("epoch", "reward", "max_ineq", "max_eq"), epochs=int(ends[-2]) + 5, times=1)   # fills up early
    want, pending = [], []
    for i in range(total):                                     # the per-row form
        pending.append((i, stats[i, 0, S["max_ineq_sum"]], stats[i, 0, S["max_eq_sum"]]))
        if stats[i, 0, S["episodes"]] > 0:
            want.extend((ep, stats[i, 0, S["return_sum"]], mi, me) for ep, mi, me in pending)
            pending = []
    want = np.array(want[:tr.logger.capacity], dtype=np.float64)
    for lo, hi in ((0, 3), (3, 250), (250, 251), (251, 640), (640, 700)):
        tr.vec.stats[torch.arange(lo, hi) % cap] = torch.tensor(stats[lo:hi])
        tr._t = hi
        tr._harvest()
    assert tr.logger.pointer == len(want) == tr.logger.capacity
    for col, key in enumerate(("epoch", "reward", "max_ineq", "max_eq")):
        np.testing.assert_allclose(tr.logger.tracker[key][:len(want)], want[:, col], rtol=1e-6)
    assert [int(p[0]) for p in tr._pending] == list(range(int(ends[-1]) + 1, total))
    out = capsys.readouterr().out
    assert out.count("episode") == 9 and ("episode %d ends." % (ends[0] + 1)) in out


def test_entropy_tuning_steps_log_alpha_in_the_flat_buffer():
    """rpo_sac.py:210-216 with `automatic_entropy_tuning=True` (no script enables it; the reference's target entropy is
    read from uninitialised memory, sac_pa.py:60): log_alpha is a word of the flat parameter buffer behind the
    multipliers, its gradient -(mean log pi + H_target) travels in the policy-step bucket and the fused Adam steps it;
    the first steps of Adam move it by lr each, against the sign of the gradient.  The losses keep the agent's fixed
    alpha, like the reference (rpo_sac.py:331,347)."""
    torch.set_num_threads(1)
    for fused in (True, False):
        torch.manual_seed(5)
        tr = build_trainer("sac", "cart", ob, torch.device("cpu"), fused=fused, num_envs=8, use_graph=False,
                           automatic_entropy_tuning=True, lr_alpha=1e-3)
        ag, fl = tr.agent, tr.agent.flat
        lo, hi = fl.policy_bucket
        assert fl.offset[id(ag.log_alpha)] == hi - 4 and ag.log_alpha.data_ptr() == fl.data[hi - 4:].data_ptr()
        tr.vec.reset()
        tr.run_steps(12)                                           # policy steps at t = 4, 8, 12
        assert int(ag.alpha_optim.step_dev[0]) == 3
        sign = -float(torch.sign(ag.log_alpha.grad)[0])
        np.testing.assert_allclose(float(ag.log_alpha), 3e-3 * sign, rtol=5e-3)
        np.testing.assert_allclose(float(tr.alpha), np.exp(3e-3 * sign), rtol=1e-4)
        assert ag.alpha == 0.1                                     # what the losses use


def test_bench_refuses_to_print_a_line_for_fewer_gpus_than_asked():
    """`python bench.py --gpus N` without a launcher starts its N ranks itself -- and when fewer than N GPUs are visible it
    exits non-zero with a message instead of printing a 1-GPU line (this container has none)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_bench_timing_and_regime_helpers():
    """bench.py (round 4): value = median of the repeated timed regions with the spread next to it; an HBM-priced launch is
    labelled by the working set its back-to-back replays touch (LLC-resident below the 256 MiB Infinity Cache -- never an
    'HBM fraction' -- else streaming, also priced against the 6.29 TB/s a copy achieves)."""
    import bench
    st = bench.region_stats([0.0010, 0.0008, 0.0009, 0.0030, 0.00085], 20)
    assert st["timed_regions"] == 5 and abs(st["median"] - 0.0009) < 1e-12
    assert abs(st["ms_per_step"] - 0.045) < 1e-9 and abs(st["ms_per_step_min"] - 0.04) < 1e-9 and abs(st["ms_per_step_max"] - 0.15) < 1e-9
    llc = bench.hbm_regime(42e6, 4400.0)
    assert llc["regime"].startswith("LLC-resident") and llc["frac_of_achievable"] is None
    hbm = bench.hbm_regime(1.07e9, 4400.0)
    assert hbm["regime"].startswith("HBM streaming") and abs(hbm["frac_of_achievable"] - 4400.0 / 6290.0) < 1e-9
    assert bench.hbm_regime(1.07e9, 9000.0)["frac_of_achievable"] == 1.0


def test_evopf_static_order_is_refused_for_another_network():
    """The EVOPF solver's static elimination order has case14's branch pattern compiled in (csrc/evopf_dev.h: kAdjMask): a
    constant table whose Ybus has an entry outside that pattern -- or RPO_EVOPF_PIVOT=dynamic -- selects partial pivoting
    (RPO_EVOPF_C_FLAGS), decided on the host when the kernel set is built (no GPU involved)."""
    from rpo_amd import ops
    from rpo_amd.env.electrical_grid import case14
    C = ops.CONST
    table = case14.kernel_constants(C, C["RPO_EVOPF_CONSTS_LEN"])
    k = ops.EvopfKernels(table)
    assert table[C["RPO_EVOPF_C_FLAGS"]] == 0.0                  # an unvalidated table asks for partial pivoting (the safe default)
    assert k.static_order and k.consts[C["RPO_EVOPF_C_FLAGS"]] == 1.0
    yr = table[C["RPO_EVOPF_C_YR"]:C["RPO_EVOPF_C_YR"] + 196].reshape(14, 14)
    inside = np.array([[(m >> j) & 1 for j in range(14)] for m in ops.EvopfKernels.CASE14_ADJ], dtype=bool)
    yi = table[C["RPO_EVOPF_C_YI"]:C["RPO_EVOPF_C_YI"] + 196].reshape(14, 14)
    np.testing.assert_array_equal((yr != 0) | (yi != 0), inside)   # the compiled-in masks ARE case14's pattern
    other = table.copy()
    other[C["RPO_EVOPF_C_YR"] + 0 * 14 + 13] = 0.5              # a branch 1 - 14 that case14 does not have
    k2 = ops.EvopfKernels(other)
    assert not k2.static_order and k2.consts[C["RPO_EVOPF_C_FLAGS"]] == 0.0
    os.environ["RPO_EVOPF_PIVOT"] = "dynamic"
    try:
        assert not ops.EvopfKernels(table).static_order
    finally:
        os.environ.pop("RPO_EVOPF_PIVOT")


def test_schedule_parsing_and_ring_validation(monkeypatch):
    """Round 5: ONE structured switch for the optional parts of the launch schedule (RPO_SCHEDULE / schedule=), read at
    construction; unknown parts are refused.  And the replay ring's width is validated where tensors become pointers (ADVICE
    r04: the ring stride is compiled into the step / rollout / rider / fused-sampling kernels)."""
    import torch
    from rpo_amd import ops
    from rpo_amd.algo.trainer import SCHEDULE_DEFAULTS, parse_schedule
    monkeypatch.delenv("RPO_SCHEDULE", raising=False)
    assert parse_schedule() == SCHEDULE_DEFAULTS and SCHEDULE_DEFAULTS["front"] == 1 and SCHEDULE_DEFAULTS["force_dist"] == 0
    monkeypatch.setenv("RPO_SCHEDULE", "front=0, ride=0,force_dist")
    s = parse_schedule(dict(split=0))
    assert (s["front"], s["ride"], s["force_dist"], s["split"], s["fused_mlp"]) == (0, 0, 1, 0, 1)
    monkeypatch.setenv("RPO_SCHEDULE", "frnot=0")
    with pytest.raises(ValueError):
        parse_schedule()
    monkeypatch.delenv("RPO_SCHEDULE")
    with pytest.raises(ValueError):
        parse_schedule(dict(nope=1))
    # a [rows, 24] ring (ABI <= 3) where the kernels stride 32 floats: refused before anything is launched
    k = ops.CartSafeKernels(np.zeros(ops.CONST["RPO_CART_CONSTS_LEN"], dtype=np.float32), 1)
    with pytest.raises(ops.RpoHipError, match="ring"):
        ops._ring(torch.zeros(8, k.row_floats), k.ring_floats)
    with pytest.raises(ops.RpoHipError, match="ring"):
        ops.RolloutRider(ring_floats=k.ring_floats, rows=torch.zeros(8, k.row_floats))
    assert ops._ring(None, k.ring_floats) is None


def test_tuning_switches_round_trip():
    """rpo_tuning (the library reads no environment variable): set / restore through the context manager, unknown keys refused,
    defaults = what ships (streaming forward on 16 waves, streaming backward)."""
    from rpo_amd import ops
    assert ops.tuning.get("fwd_stream") == 1 and ops.tuning.get("fwd_stream_waves") == 16 and ops.tuning.get("bwd_stream") == 1
    with ops.tuning(fwd_stream=0, bwd_onepass=0):
        assert ops.tuning.get("fwd_stream") == 0 and ops.tuning.get("bwd_onepass") == 0
    assert ops.tuning.get("fwd_stream") == 1 and ops.tuning.get("bwd_onepass") == 1
    with pytest.raises(ops.RpoHipError):
        ops.tuning(no_such_key=1)
