"""Parity of the HIP kernels (through the C ABI) with the oracle and the reference's golden vectors.  Needs an MI355X.

Tolerances: the reference's env dynamics are float64 (numpy), the kernels float32 -- per-step agreement from identical
inputs is required to 1e-4 relative / absolute on accelerations (|value| up to ~60) and 1e-5 on positions and
velocities; float32 constraint arithmetic agrees to a few ulp (2e-6 .. 1e-5); integer work (Philox words, sampled row
indices, gathered rows, iteration counts away from the 1e-5 stop threshold) is bit-exact.
"""
import numpy as np
import pytest
import torch

from oracle import cartsafe as cs
from oracle import pendulum as pd
from oracle import philox, train_ops

pytestmark = pytest.mark.gpu

DEV = "cuda"


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(DEV)


@pytest.fixture(scope="module")
def ops():
    from rpo_amd import ops as _ops
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _ops


def cart_kernels(ops, partial):
    return ops.CartSafeKernels(cs.Constants(partial).as_array(), partial)


def env_buffers(n):
    return (torch.zeros(n, dtype=torch.int32, device=DEV), torch.zeros(n, device=DEV),
            torch.zeros(n, dtype=torch.int32, device=DEV))


# ------------------------------------------------------------------------------------------------------- Philox

def test_philox_bit_exact(ops):
    n = 5000
    out = torch.zeros(n, 4, dtype=torch.int32, device=DEV)
    for seed, base, index, tag in [(123, 0, 0, 1), (0xDEADBEEFCAFE1234, 4096, 77, 2), (1, 2 ** 31, 2 ** 32 - 1, 3)]:
        ops.philox_fill(out, seed, base, index, tag)
        got = out.cpu().numpy().view(np.uint32)
        want = philox.draw(seed, (np.arange(n, dtype=np.uint64) + base).astype(np.uint32), index, tag)
        np.testing.assert_array_equal(got, want)


# ----------------------------------------------------------------------------------------------------- CartSafe

@pytest.mark.parametrize("partial", [1, 0])
def test_cart_step_matches_reference(ops, golden, partial):
    g = golden("cart_env_p%d" % partial)
    k = cart_kernels(ops, partial)
    n = g["states"].shape[0]
    state = dev(g["states"])
    action = dev(g["actions"])
    ep_len, ep_ret, ep_count = env_buffers(n)
    rows = torch.zeros(n, k.ring_floats, device=DEV)          # ring stride 32: one 96-byte transition per 128-byte line
    stats = ops.new_stats(4, DEV)
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    k.step(state, state, action, ep_len, ep_ret, ep_count, rows, 1, stats, ctrl, 200, False, 1e-3, 7, 0)
    torch.cuda.synchronize()
    r = rows.cpu().numpy()
    nxt = g["next_states"]
    np.testing.assert_array_equal(r[:, 0:6], g["states"].astype(np.float32))
    np.testing.assert_array_equal(r[:, 6:8], g["actions"])
    np.testing.assert_allclose(r[:, [8, 9, 11, 12]], nxt[:, [0, 1, 3, 4]], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r[:, [10, 13]], nxt[:, [2, 5]], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(r[:, 14], g["reward"].astype(np.float32))
    # termination flags: identical except where the float64 next state sits within 1e-5 of a threshold
    edge = (np.abs(np.abs(nxt[:, 0]) - cs.X_THRESHOLD) < 1e-5) | (np.abs(np.abs(nxt[:, 3]) - cs.THETA_THRESHOLD) < 1e-5)
    np.testing.assert_array_equal(r[~edge, 15] > 0.5, g["done"][~edge])
    np.testing.assert_allclose(r[:, 16:17], g["eq_viol"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(r[:, 17:23], g["ineq_viol"], rtol=2e-6, atol=4e-6)
    np.testing.assert_array_equal(r[:, 23], 0)
    # state updated in place (no auto reset requested), episode bookkeeping, statistics row, step counter
    np.testing.assert_array_equal(state.cpu().numpy(), r[:, 8:14])
    assert int(ctrl[0]) == 1 and int(ctrl[1]) == 0
    np.testing.assert_array_equal(ep_len.cpu().numpy(), 1)
    s = ops.reduce_stats(stats[0]).cpu().numpy()
    assert s[ops.STAT["reward_sum"]] == n
    assert s[ops.STAT["episodes"]] == r[:, 15].sum() == s[ops.STAT["terminated"]]
    np.testing.assert_allclose(s[ops.STAT["max_ineq_sum"]], g["ineq_viol"].max(axis=1).sum(), rtol=1e-5)
    np.testing.assert_allclose(s[ops.STAT["max_eq_sum"]], np.abs(g["eq_viol"]).max(axis=1).sum(), rtol=1e-5)
    np.testing.assert_allclose(s[ops.STAT["max_ineq_max"]], g["ineq_viol"].max(), rtol=1e-6)
    viol = np.maximum(g["ineq_viol"].max(axis=1), np.abs(g["eq_viol"]).max(axis=1)) > 1e-3
    assert abs(s[ops.STAT["viol_count"]] - viol.sum()) <= 2


def test_cart_rollout_bookkeeping(ops):
    """TimeLimit, auto-reset through Philox, ring wrap-around and per-step statistics over several vector steps,
    against a float64 oracle simulation driven with the same actions."""
    rng = np.random.RandomState(5)
    n, cap, T, max_len, seed, base = 1500, 3, 7, 4, 99, 1000
    consts = cs.Constants(1)
    k = cart_kernels(ops, 1)
    state = torch.zeros(n, 6, device=DEV)
    ep_len, ep_ret, ep_count = env_buffers(n)
    rows = torch.zeros(cap * n, k.ring_floats, device=DEV)
    stats = ops.new_stats(16, DEV)
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    k.reset(state, state, ep_len, ep_ret, ep_count, seed, base)
    ids = np.arange(n) + base
    o_state = philox.cart_reset(seed, ids, 0).astype(np.float64)
    np.testing.assert_allclose(state.cpu().numpy(), o_state, rtol=0, atol=1e-8)
    o_state[: n // 3, 0] = 2.39          # a third of the lanes start next to the track edge -> early termination
    o_state[: n // 3, 1] = 3.0
    state.copy_(dev(o_state))
    o_len = np.zeros(n, dtype=int)
    o_ret = np.zeros(n)
    o_cnt = np.zeros(n, dtype=int)
    ring = np.zeros((cap * n, 24), dtype=np.float32)
    for t in range(T):
        act = rng.uniform(-12, 12, size=(n, 2)).astype(np.float32)
        k.step(state, state, dev(act), ep_len, ep_ret, ep_count, rows, cap, stats, ctrl, max_len, True, 1e-3, seed, base)
        nxt, rew, term, ineq, eq = cs.step(o_state, act, consts)
        o_len += 1
        o_ret += rew
        done = term | (o_len >= max_len)
        row = np.concatenate([o_state, act, nxt, rew[:, None], done[:, None], eq, ineq, np.zeros((n, 1))], axis=1)
        ring[(t % cap) * n:(t % cap + 1) * n] = row
        srow = ops.reduce_stats(stats[t]).cpu().numpy()
        assert srow[ops.STAT["episodes"]] == done.sum()
        assert srow[ops.STAT["terminated"]] == term.sum()
        np.testing.assert_allclose(srow[ops.STAT["return_sum"]], o_ret[done].sum(), rtol=1e-6)
        np.testing.assert_allclose(srow[ops.STAT["length_sum"]], o_len[done].sum(), rtol=1e-6)
        o_cnt += done
        fresh = philox.cart_reset(seed, ids, o_cnt).astype(np.float64)
        o_state = np.where(done[:, None], fresh, nxt)
        o_len[done] = 0
        o_ret[done] = 0
        # resynchronise the float32 device state with the float64 oracle so that errors do not accumulate
        got = state.cpu().numpy()
        np.testing.assert_allclose(got, o_state, rtol=1e-4, atol=1e-4)
        o_state = got.astype(np.float64)
        np.testing.assert_array_equal(ep_len.cpu().numpy(), o_len)
        np.testing.assert_array_equal(ep_count.cpu().numpy(), o_cnt)
    assert int(ctrl[0]) == T
    assert o_cnt.sum() > n            # resets through both the termination test and the TimeLimit happened
    np.testing.assert_allclose(rows.cpu().numpy()[:, :24], ring, rtol=1e-4, atol=1e-4)
    assert float(rows[:, 24:].abs().max()) == 0.0                # the padding of the 128-byte ring rows stays zero
    np.testing.assert_array_equal(stats[T].cpu().numpy(), 0)      # next row pre-cleared by the last launch


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_act_project_matches_reference(ops, golden, partial):
    g = golden("cart_grad_steps_p%d" % partial)
    k = cart_kernels(ops, partial)
    n = g["ap"].shape[0]
    ap = dev(g["ap"].reshape(-1))
    action = torch.zeros(n, 2, device=DEV)
    iters = torch.zeros(n, dtype=torch.int32, device=DEV)
    # complete_partial only (max_steps = 0)
    k.act_project(None, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -10, 10, 0, 2e-2, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["completed"], rtol=2e-6, atol=2e-6)
    # training projection: K = 10, lr = 2e-2 (scripts/cart_exp.py:26-27)
    k.act_project(None, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -10, 10, 10, 2e-2, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["train_b1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(action.cpu().numpy(), g["train_batched"], rtol=1e-5, atol=1e-5)
    # evaluation projection: K = 50, returns the iteration count (rpo_ddpg.py:305)
    k.act_project(None, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -10, 10, 50, 2e-2, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["eval_b1"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(iters.cpu().numpy(), g["eval_b1_iters"])
    # momentum 0.5 (the trainer's default)
    k.act_project(None, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -10, 10, 10, 2e-2, 1e-5, 0.5)
    np.testing.assert_allclose(action.cpu().numpy(), g["train_b1_mom"], rtol=1e-5, atol=1e-5)


def test_cart_exploration_modes(ops):
    n, seed, base = 4096, 31, 512
    c = cs.Constants(1)
    k = cart_kernels(ops, 1)
    rng = np.random.RandomState(1)
    ap = rng.uniform(-10, 10, size=n).astype(np.float32)
    noise = rng.randn(n).astype(np.float32)
    action = torch.zeros(n, 2, device=DEV)
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    ctrl[0] = 40
    stats = ops.new_stats(64, DEV)
    iters = torch.zeros(n, dtype=torch.int32, device=DEV)
    # explicit noise, eps decays with t: eps_t = max(0.1, 1.0 - 0.01 * 40) = 0.6
    k.act_project(None, dev(ap), dev(noise), action, iters, ops.NOISE_EXPLICIT, 1.0, 0.1, 0.01, -10, 10, 10, 2e-2, 1e-5,
                  0.0, seed, base, ctrl, stats)
    ap_n = np.clip(ap + np.float32(0.6) * noise, -10, 10).astype(np.float32)
    want, it = cs.grad_steps(cs.complete_partial(ap_n, c), c, 2e-2, 10)
    np.testing.assert_allclose(action.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert float(ops.reduce_stats(stats[40])[ops.STAT["proj_iters"]]) == it.sum() == int(iters.sum())
    # Philox normal noise: reproducible from (seed, env id, t); statistics of a standard normal
    k.act_project(None, dev(ap), None, action, None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -1e9, 1e9, 0, 0, 1e-5, 0.0, seed,
                  base, ctrl, None)
    r = philox.draw(seed, np.arange(n) + base, 40, philox.STREAM_ACT)
    z = philox.normal(r[:, 0], r[:, 1])
    got = action.cpu().numpy()[:, 1]
    np.testing.assert_allclose(got, ap + z, rtol=1e-5, atol=2e-5)
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05
    # warm-up: uniform in the box (BoxConstraint.sample)
    k.act_project(None, None, None, action, None, ops.NOISE_UNIFORM, 0, 0, 0, -10, 10, 0, 0, 1e-5, 0.0, seed, base, ctrl,
                  None)
    u = philox.u01(r[:, 0])
    np.testing.assert_allclose(action.cpu().numpy()[:, 1], 10 * (2 * u - 1), rtol=1e-6, atol=1e-5)
    # env-id keyed streams: splitting the lanes over two launches ("two ranks") gives bit-identical actions
    a2 = torch.zeros(n, 2, device=DEV)
    h = n // 2
    k.act_project(None, dev(ap[:h]), None, a2[:h], None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10, 10, 10, 2e-2, 1e-5, 0.0,
                  seed, base, ctrl, None)
    k.act_project(None, dev(ap[h:]), None, a2[h:], None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10, 10, 10, 2e-2, 1e-5, 0.0,
                  seed, base + h, ctrl, None)
    k.act_project(None, dev(ap), None, action, None, ops.NOISE_PHILOX, 1.0, 1.0, 0.0, -10, 10, 10, 2e-2, 1e-5, 0.0, seed,
                  base, ctrl, None)
    assert torch.equal(a2, action)


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_constraint_kernels(ops, golden, partial):
    g = golden("cart_env_p%d" % partial)
    c = cs.Constants(partial)
    k = cart_kernels(ops, partial)
    a = dev(g["any_actions"])
    n = a.shape[0]
    eq = torch.zeros(n, device=DEV)
    ineq = torch.zeros(n, 6, device=DEV)
    k.resid(None, a, eq, ineq)
    np.testing.assert_allclose(eq.cpu().numpy()[:, None], g["eq_resid_any"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(ineq.cpu().numpy(), g["ineq_resid_any"], rtol=2e-6, atol=4e-6)
    stp = torch.zeros(n, 2, device=DEV)
    k.ineq_partial_grad(None, a, stp)
    np.testing.assert_allclose(stp.cpu().numpy(), g["ipg_any"], rtol=2e-6, atol=2e-6)
    k.ineq_partial_grad(None, dev(g["completed"]), stp[: g["completed"].shape[0]])
    np.testing.assert_allclose(stp.cpu().numpy(), g["ipg_completed"], rtol=2e-6, atol=2e-6)
    # backward of complete_partial == finite difference of the oracle's complete_partial
    ga = dev(np.random.RandomState(0).randn(n, 2))
    gap = torch.zeros(n, device=DEV)
    k.complete_bwd(None, ga, gap)
    jac = cs.complete_partial(np.ones((1, 1), np.float32), c) - cs.complete_partial(np.zeros((1, 1), np.float32), c)
    np.testing.assert_allclose(gap.cpu().numpy(), ga.cpu().numpy() @ jac[0], rtol=1e-6, atol=1e-6)
    # Lagrangian forward + backward
    nu = np.array([0.3, 0.0, 1.5, 0.2, 0.7, 0.05], dtype=np.float32)
    loss = torch.zeros(1, device=DEV)
    g_a = torch.zeros(n, 2, device=DEV)
    g_nu = torch.zeros(6, device=DEV)
    k.lagrangian(a, dev(nu), 1.0 / n, loss, g_a, g_nu)
    w_loss, w_ga, w_gnu = train_ops.lagrangian_cart(g["any_actions"], nu, c, 1.0 / n)
    np.testing.assert_allclose(float(loss), w_loss, rtol=1e-5)
    np.testing.assert_allclose(g_a.cpu().numpy(), w_ga, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(g_nu.cpu().numpy(), w_gnu, rtol=1e-5, atol=1e-7)


# ----------------------------------------------------------------------------------------------- SpringPendulum

def test_pendulum_step_matches_reference(ops, golden):
    g = golden("pendulum_env")
    k = ops.PendulumKernels()
    n = g["internal"].shape[0]
    internal = dev(g["internal"])
    obs = torch.zeros(n, 5, device=DEV)
    ep_len, ep_ret, ep_count = env_buffers(n)
    rows = torch.zeros(n, 16, device=DEV)
    stats = ops.new_stats(4, DEV)
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    k.step(internal, obs, dev(g["actions"]), ep_len, ep_ret, ep_count, rows, 1, stats, ctrl, 200, False, 1e-3, 7, 0)
    r = rows.cpu().numpy()
    pre_obs = pd.get_obs(g["internal"])
    np.testing.assert_allclose(r[:, 0:5], pre_obs, rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(r[:, 5:7], g["actions"])
    np.testing.assert_allclose(r[:, 7:12], g["next_obs"], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(r[:, 12], g["reward"], rtol=2e-5, atol=1e-6)
    nxt = g["next_internal"]
    edge = (np.abs(nxt[:, 2] - 0.5) < 1e-5) | (np.abs(nxt[:, 2] - 1.5) < 1e-5) | (np.abs(np.abs(nxt[:, 0]) - np.pi / 12) < 1e-5)
    np.testing.assert_array_equal(r[~edge, 13] > 0.5, g["done"][~edge])
    np.testing.assert_allclose(r[:, 14:15], g["eq_viol"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(r[:, 15:16], g["ineq_viol"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(internal.cpu().numpy(), nxt, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(obs.cpu().numpy(), g["next_obs"], rtol=2e-5, atol=2e-5)
    assert int(ctrl[0]) == 1
    s = ops.reduce_stats(stats[0]).cpu().numpy()
    np.testing.assert_allclose(s[ops.STAT["reward_sum"]], g["reward"].sum(), rtol=1e-5)
    assert s[ops.STAT["episodes"]] == r[:, 13].sum()


def test_pendulum_reset_and_autoreset(ops):
    n, seed, base = 3000, 17, 64
    k = ops.PendulumKernels()
    internal = torch.zeros(n, 4, device=DEV)
    obs = torch.zeros(n, 5, device=DEV)
    ep_len, ep_ret, ep_count = env_buffers(n)
    k.reset(internal, obs, ep_len, ep_ret, ep_count, seed, base)
    want = philox.pendulum_reset(seed, np.arange(n) + base, 0)
    np.testing.assert_allclose(internal.cpu().numpy(), want, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs.cpu().numpy(), pd.get_obs(want.astype(np.float64)), rtol=1e-6, atol=1e-6)
    # max_episode_steps = 1: every lane is done after one step and re-drawn with episode counter 1
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    k.step(internal, obs, torch.zeros(n, 2, device=DEV), ep_len, ep_ret, ep_count, None, 1, None, ctrl, 1, True, 1e-3,
           seed, base)
    want1 = philox.pendulum_reset(seed, np.arange(n) + base, 1)
    np.testing.assert_allclose(internal.cpu().numpy(), want1, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(ep_count.cpu().numpy(), 1)
    np.testing.assert_array_equal(ep_len.cpu().numpy(), 0)


def test_pendulum_act_project_matches_reference(ops, golden):
    g = golden("pendulum_grad_steps")
    k = ops.PendulumKernels()
    obs = dev(g["obs32"])
    n = obs.shape[0]
    ap = dev(g["ap"].reshape(-1))
    action = torch.zeros(n, 2, device=DEV)
    iters = torch.zeros(n, dtype=torch.int32, device=DEV)
    k.act_project(obs, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -6, 6, 0, 2e-3, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["completed"], rtol=1e-5, atol=1e-5)
    k.act_project(obs, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -6, 6, 10, 2e-3, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["train_b1"], rtol=1e-4, atol=1e-4)
    k.act_project(obs, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -6, 6, 50, 2e-3, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), g["eval_b1"], rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(iters.cpu().numpy(), g["eval_b1_iters"])
    k.act_project(obs, ap, None, action, iters, ops.NOISE_NONE, 0, 0, 0, -6, 6, 50, 2e-2, 1e-5, 0.0)
    same = iters.cpu().numpy() == g["eval_b1_lr2e2_iters"]
    assert same.mean() > 0.97         # a 1-ulp difference at the 1e-5 stop threshold can add one iteration
    np.testing.assert_allclose(action.cpu().numpy()[same], g["eval_b1_lr2e2"][same], rtol=1e-3, atol=1e-3)
    # obs given as columns of a wider (batch) matrix
    wide = torch.zeros(n, 16, device=DEV)
    wide[:, 7:12] = obs
    a2 = torch.zeros(n, 2, device=DEV)
    k.act_project(wide[:, 7:12], ap, None, a2, None, ops.NOISE_NONE, 0, 0, 0, -6, 6, 10, 2e-3, 1e-5, 0.0)
    k.act_project(obs, ap, None, action, None, ops.NOISE_NONE, 0, 0, 0, -6, 6, 10, 2e-3, 1e-5, 0.0)
    assert torch.equal(a2, action)


@pytest.mark.parametrize("n", [1, 3, 17, 100, 255, 256, 300, 1000])
def test_pendulum_batched_projection_matches_reference_arithmetic(ops, golden, n):
    """rpo_pendulum_project_batchref = RPODDPG.grad_steps on a batch, literally (rpo_ddpg.py:266-286 with the [B,1] @ [1,B]
    coupling of pendulum.py:337-339).  After ONE iteration the selected set 1[a_x,i dgp_j - bgp_i > 0] is a function of the
    inputs alone, so kernel and oracle must agree to summation round-off (the reference's own sum is a BLAS matmul without a
    defined order); over 10 iterations a 1-ulp difference can flip a predicate that sits on its threshold, which moves
    that row by one lr * dgp_j ~ 1e-2 -- the tolerance the CPU test of the oracle against the reference uses as well.
    n <= 256 runs the register-tiled workgroup (project_batchref_wide), larger batches one thread per sample."""
    g = golden("pendulum_grad_steps")
    rng = np.random.default_rng(n)
    pick = rng.integers(0, 256, n)
    obs_n = g["obs32"][pick].astype(np.float32)
    ap_n = (g["ap"].reshape(-1)[pick] + rng.normal(0, 0.5, n)).astype(np.float32)
    k = ops.PendulumKernels()
    obs, ap = dev(obs_n), dev(ap_n)
    action = torch.zeros(n, 2, device=DEV)
    iters = torch.zeros(1, dtype=torch.int32, device=DEV)
    a0 = pd.complete_partial(obs_n, ap_n)
    k.project_batchref(obs, ap, action, iters, 0, 2e-3, 1e-5, 0.0)
    np.testing.assert_allclose(action.cpu().numpy(), a0, rtol=2e-6, atol=2e-6)
    for lr in (2e-3, 0.05):
        want, it = pd.grad_steps(obs_n, a0, lr=lr, max_steps=1, batch_global_stop=True, batched_reference=True)
        k.project_batchref(obs, ap, action, iters, 1, lr, 1e-5, 0.0)
        scale = np.abs(want).max() + 1.0
        np.testing.assert_allclose(action.cpu().numpy(), want, rtol=0, atol=4e-6 * scale * max(1.0, lr * n))
        assert int(iters.item()) == 1
    lr10 = 2e-3 * min(1.0, 256.0 / n)                            # (the coupled step grows with the batch: keep it contracting)
    want, it = pd.grad_steps(obs_n, a0, lr=lr10, max_steps=10, batch_global_stop=True, batched_reference=True)
    k.project_batchref(obs, ap, action, iters, 10, lr10, 1e-5, 0.0)
    got = action.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=2e-2)
    assert (np.abs(got - want).max(axis=1) < 1e-4).mean() > 0.9
    assert int(iters.item()) == int(it.max())
    # momentum, and the batch-global stop: a feasible batch stops after its first iteration
    want, it = pd.grad_steps(obs_n, a0, lr=lr10, max_steps=10, momentum=0.5, batch_global_stop=True, batched_reference=True)
    k.project_batchref(obs, ap, action, iters, 10, lr10, 1e-5, 0.5)
    np.testing.assert_allclose(action.cpu().numpy(), want, rtol=1e-3, atol=4e-2)
    small = dev(np.zeros(n, dtype=np.float32))
    want, it = pd.grad_steps(obs_n, pd.complete_partial(obs_n, np.zeros(n, np.float32)), lr=0.0, max_steps=10,
                             batch_global_stop=True, batched_reference=True)
    k.project_batchref(obs, small, action, iters, 10, 0.0, 1e-5, 0.0)
    assert int(iters.item()) == int(it.max())
    if n == 256:                                                  # the reference's own batched result
        k.project_batchref(dev(g["obs32"]), dev(g["ap"].reshape(-1)), action, iters, 10, 2e-3, 1e-5, 0.0)
        np.testing.assert_allclose(action.cpu().numpy(), g["train_batched"], rtol=1e-3, atol=1e-2)
    if n <= 256:
        # the eight-workgroup form (tagged-granule all-gather per GRG iteration) gives the same bits, with agent-scope granule
        # stores (0), with the in-launch XCD check (1) and with plain stores forced (2); its workspace survives launches
        ws = torch.zeros(ops.PROJ_WS_WORDS, dtype=torch.int64, device=DEV)
        a2, it2 = torch.zeros(n, 2, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
        for steps, lr, mom in [(10, 2e-3, 0.0), (1, 0.05, 0.0), (0, 2e-3, 0.0), (10, 2e-3, 0.5), (30, 2e-2, 0.0)]:
            k.project_batchref(obs, ap, action, iters, steps, lr, 1e-5, mom)
            for mode in (0, 1, 2, 1):
                a2.fill_(-1.0)
                k.project_batchref_ws(obs, ap, a2, it2, steps, lr, 1e-5, mom, ws, mode)
                assert torch.equal(a2, action), (steps, lr, mom, mode)
                assert int(it2.item()) == int(iters.item())
        assert int(ws[ops.PROJ_WS_GAVE_UP]) == 0
        assert int(ws[ops.PROJ_WS_GAVE_UP - 1]) == 20             # the launch counter of the tags (one per launch)


def test_pendulum_constraint_kernels(ops, golden):
    g = golden("pendulum_env")
    k = ops.PendulumKernels()
    obs = dev(g["obs32"])
    a = dev(g["any_actions"])
    n = a.shape[0]
    eq = torch.zeros(n, device=DEV)
    ineq = torch.zeros(n, device=DEV)
    k.resid(obs, a, eq, ineq)
    np.testing.assert_allclose(eq.cpu().numpy()[:, None], g["eq_resid_any"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ineq.cpu().numpy()[:, None], g["ineq_resid_any"], rtol=1e-5, atol=2e-5)
    stp = torch.zeros(n, 2, device=DEV)
    k.ineq_partial_grad(obs, a, stp)
    np.testing.assert_allclose(stp.cpu().numpy(), g["ipg_any_b1"], rtol=1e-5, atol=1e-4)
    ga = dev(np.random.RandomState(0).randn(n, 2))
    gap = torch.zeros(n, device=DEV)
    k.complete_bwd(obs, ga, gap)
    o = g["obs32"]
    np.testing.assert_allclose(gap.cpu().numpy(), ga.cpu().numpy()[:, 0] - ga.cpu().numpy()[:, 1] * o[:, 1] / o[:, 0],
                               rtol=1e-5, atol=1e-5)
    nu = np.array([0.37], dtype=np.float32)
    loss = torch.zeros(1, device=DEV)
    g_a = torch.zeros(n, 2, device=DEV)
    g_nu = torch.zeros(1, device=DEV)
    k.lagrangian(a, dev(nu), 1.0 / n, loss, g_a, g_nu)
    w_loss, w_ga, w_gnu = train_ops.lagrangian_pendulum(g["any_actions"], nu, 1.0 / n)
    np.testing.assert_allclose(float(loss), w_loss, rtol=1e-5)
    np.testing.assert_allclose(g_a.cpu().numpy(), w_ga, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(g_nu.cpu().numpy(), w_gnu, rtol=1e-5)


# ------------------------------------------------------------------------------------------------ replay ring

@pytest.mark.parametrize("row_floats,ring_floats", [(24, 32), (24, 24), (16, 16)])
def test_replay_gather_and_sample(ops, row_floats, ring_floats):
    """(24, 32) = CartSafe: a 96-byte transition per 128-byte line of the ring, gathered into 24-float batch rows."""
    rng = np.random.RandomState(3)
    n_envs, cap = 257, 5
    ring_np = rng.randn(cap * n_envs, ring_floats).astype(np.float32)
    rows_np = ring_np[:, :row_floats]
    rows = dev(ring_np)
    idx = rng.randint(0, cap * n_envs, size=1000)
    out = torch.zeros(1000, row_floats, device=DEV)
    ops.replay_gather(rows, dev(idx, torch.int64), out)
    np.testing.assert_array_equal(out.cpu().numpy(), rows_np[idx])            # byte-exact copy
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    for t, salt in [(1, 0), (3, 0), (3, 1), (5, 0), (12, 7)]:                 # partly filled, full, wrapped
        ctrl[0] = t
        got_idx = torch.zeros(256, dtype=torch.int64, device=DEV)
        batch = torch.zeros(256, row_floats, device=DEV)
        ops.replay_sample_gather(rows, cap, n_envs, batch, got_idx, 42, salt, ctrl)
        n_valid = min(t, cap) * n_envs
        want = philox.sample_indices(42, 256, t, salt, n_valid)
        np.testing.assert_array_equal(got_idx.cpu().numpy(), want)
        assert want.max() < n_valid
        np.testing.assert_array_equal(batch.cpu().numpy(), rows_np[want])


def test_replay_sampling_is_uniform(ops):
    n_envs, cap, B = 64, 16, 1 << 16
    rows = torch.arange(cap * n_envs, device=DEV, dtype=torch.float32).repeat_interleave(16).reshape(-1, 16).contiguous()
    ctrl = torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV)
    ctrl[0] = 100
    batch = torch.zeros(B, 16, device=DEV)
    ops.replay_sample_gather(rows, cap, n_envs, batch, None, 9, 0, ctrl)
    counts = np.bincount(batch[:, 0].cpu().numpy().astype(int), minlength=cap * n_envs)
    expect = B / (cap * n_envs)
    chi2 = ((counts - expect) ** 2 / expect).sum()
    assert chi2 < 1.25 * cap * n_envs          # ~N(1024, 45): a biased index map fails by a wide margin


# ------------------------------------------------------------------------------------------------- update step

@pytest.mark.parametrize("sac", [False, True])
def test_td_huber(ops, sac):
    rng = np.random.RandomState(11)
    n = 1000
    q1, q2, qn1, qn2 = [(3 * rng.randn(n)).astype(np.float32) for _ in range(4)]
    logp = rng.randn(n).astype(np.float32)
    batch = rng.randn(n, 24).astype(np.float32)
    batch[:, 15] = rng.rand(n) < 0.2
    bt = dev(batch)
    loss = torch.zeros(1, device=DEV)
    g1 = torch.zeros(n, device=DEV)
    g2 = torch.zeros(n, device=DEV)
    y = torch.zeros(n, device=DEV)
    if sac:
        ops.td_huber(dev(q1), dev(q2), dev(qn1), dev(qn2), dev(logp), 0.1, bt[:, 14:15], bt[:, 15:16], 0.95, loss, g1, g2, y)
        w = train_ops.td_huber(q1, qn1, batch[:, 14], batch[:, 15], 0.95, q2=q2, qn2=qn2, logp=logp, alpha=0.1)
        np.testing.assert_allclose(g2.cpu().numpy(), w[3], rtol=1e-6, atol=1e-9)
    else:
        ops.td_huber(dev(q1), None, dev(qn1), None, None, 0.0, bt[:, 14:15], bt[:, 15:16], 0.95, loss, g1, None, y)
        w = train_ops.td_huber(q1, qn1, batch[:, 14], batch[:, 15], 0.95)
    np.testing.assert_allclose(float(loss), w[0], rtol=1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), w[1], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g1.cpu().numpy(), w[2], rtol=1e-6, atol=1e-9)
    # and against torch's own smooth_l1_loss + autograd on the GPU
    tq = dev(q1).requires_grad_()
    l = torch.nn.functional.smooth_l1_loss(tq, y)
    if sac:
        tq2 = dev(q2).requires_grad_()
        l = l + torch.nn.functional.smooth_l1_loss(tq2, y)
    l.backward()
    np.testing.assert_allclose(float(loss), float(l.detach()), rtol=1e-5)
    np.testing.assert_allclose(g1.cpu().numpy(), tq.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("mode", ["adam_clip_polyak", "dual"])
def test_adam_step_matches_torch(ops, mode):
    rng = np.random.RandomState(2)
    n = 67842
    p0 = rng.randn(n).astype(np.float32) * 0.1
    dual = mode == "dual"
    if dual:
        n = 6
        p0 = np.abs(rng.randn(n)).astype(np.float32) * 0.05
    param = dev(p0)
    target = dev(p0 * 0.5)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    step = torch.zeros(ops.CONST["RPO_ADAM_STATE_LEN"], dtype=torch.int32, device=DEV)
    gmax = torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV)      # 16 slots, 64 bytes apart; rpo_absmax fills slot 0
    # torch reference on the CPU (the reference trainer's own optimiser classes, rebuilt from torch.optim.Adam)
    tp = torch.nn.Parameter(torch.tensor(p0))
    topt = torch.optim.Adam([tp], lr=0.2 if dual else 3e-4, maximize=dual)
    tt = torch.tensor(p0 * 0.5)
    for it in range(5):
        gnp = (rng.randn(n) * (0.05 if it % 2 else 1.0)).astype(np.float32)
        grad = dev(gnp)
        tp.grad = torch.tensor(gnp)
        if dual:
            ops.adam_step(param, grad, m, v, step, 0.2, maximize=True, clamp_min0=True)
            topt.step()
            with torch.no_grad():
                tp.clamp_(0)                                           # DualAdam, model/dual.py:41-43
        else:
            ops.absmax(grad, gmax)
            np.testing.assert_allclose(float(gmax.max()), np.abs(gnp).max(), rtol=0)
            ops.adam_step(param, grad, m, v, step, 3e-4, clip_thres=0.2, gradmax=gmax, target=target, tau=0.005)
            torch.nn.utils.clip_grad_norm_([tp], 0.2, "inf")
            np.testing.assert_allclose(grad.cpu().numpy(), tp.grad.numpy(), rtol=1e-6, atol=1e-9)   # clipped in place
            topt.step()
            with torch.no_grad():
                tt.copy_(tt * (1.0 - 0.005) + tp.data * 0.005)
            assert float(gmax.abs().max()) == 0.0
        assert int(step[0]) == it + 1 and int(step[2]) == 0
        np.testing.assert_allclose(param.cpu().numpy(), tp.detach().numpy(), rtol=2e-6, atol=2e-7)
        if not dual:
            np.testing.assert_allclose(target.cpu().numpy(), tt.numpy(), rtol=2e-6, atol=2e-7)
    if dual:
        assert (param.cpu().numpy() >= 0).all()
    pol = dev(p0)
    ops.polyak(param, pol, 0.25)
    np.testing.assert_allclose(pol.cpu().numpy(), p0 * 0.75 + param.cpu().numpy() * 0.25, rtol=1e-6, atol=1e-8)


# -------------------------------------------------------------------------- full-size, size-independent properties

def test_cart_full_size_properties(ops):
    """BASELINE config 2 size (4096 lanes) and a 1M-lane launch: after complete+project every lane satisfies the
    equality to float32 round-off and no inequality got worse; the replay ring holds exactly the transitions that
    were stepped; trajectories are a pure function of (seed, env id) -- independent of how lanes are split."""
    c = cs.Constants(1)
    k = cart_kernels(ops, 1)
    for n in (4096, 1 << 20):
        torch.manual_seed(0)
        ap = (torch.rand(n, device=DEV) * 24 - 12)
        a = torch.zeros(n, 2, device=DEV)
        it = torch.zeros(n, dtype=torch.int32, device=DEV)
        k.act_project(None, ap, None, a, it, ops.NOISE_CLIP_ONLY, 0, 0, 0, -10, 10, 50, 2e-2, 1e-5, 0.0)
        eq = torch.zeros(n, device=DEV)
        ineq = torch.zeros(n, 6, device=DEV)
        k.resid(None, a, eq, ineq)
        assert float(eq.abs().max()) < 1e-5
        a0 = torch.zeros(n, 2, device=DEV)
        k.act_project(None, ap, None, a0, None, ops.NOISE_CLIP_ONLY, 0, 0, 0, -10, 10, 0, 2e-2, 1e-5, 0.0)
        ineq0 = torch.zeros(n, 6, device=DEV)
        k.resid(None, a0, eq, ineq0)
        assert bool((ineq.clamp(min=0).max(dim=1).values <= ineq0.clamp(min=0).max(dim=1).values + 1e-5).all())
        # each iteration lowers the active reduced inequality by corr_lr * G_r^2 = 0.02 * 1.1547^2 = 0.02667
        # (SURVEY 8c anchor), so 50 iterations remove up to 1.333 of violation
        worst0, worst = ineq0.max(dim=1).values, ineq.max(dim=1).values
        assert bool((worst <= (worst0 - 1.30).clamp(min=0) + 0.03).all())
        assert float((worst <= 1e-5).float().mean()) > 0.6 and float((worst0 <= 1e-5).float().mean()) < 0.6
        feasible0 = ineq0.max(dim=1).values <= 1e-5
        assert bool((it[feasible0] == 1).all())         # the loop always takes its first iteration (rpo_ddpg.py:270)

    n, T, cap = 4096, 12, 8
    def rollout(splits):
        state = torch.zeros(n, 6, device=DEV)
        ep_len, ep_ret, ep_count = env_buffers(n)
        rows = torch.zeros(cap * n, k.ring_floats, device=DEV)
        ctrls = [torch.zeros(ops.CTRL_LEN, dtype=torch.int64, device=DEV) for _ in splits]
        for (lo, hi) in splits:
            k.reset(state[lo:hi], state[lo:hi], ep_len[lo:hi], ep_ret[lo:hi], ep_count[lo:hi], 5, lo)
        act = torch.zeros(n, 2, device=DEV)
        for t in range(T):
            for ci, (lo, hi) in enumerate(splits):
                k.act_project(None, state[lo:hi, 3].contiguous() * 40, None, act[lo:hi], None, ops.NOISE_PHILOX, 3.0, 3.0,
                              0.0, -10, 10, 10, 2e-2, 1e-5, 0.0, 5, lo, ctrls[ci], None)
                k.step(state[lo:hi], state[lo:hi], act[lo:hi], ep_len[lo:hi], ep_ret[lo:hi], ep_count[lo:hi], None, cap,
                       None, ctrls[ci], 6, True, 1e-3, 5, lo)
        return state.clone(), ep_count.clone()
    s1, c1 = rollout([(0, n)])
    s2, c2 = rollout([(0, n // 2), (n // 2, n)])
    assert torch.equal(s1, s2) and torch.equal(c1, c2) and int(c1.sum()) >= 2 * n


def test_adam_step_multi_equals_single_launches(ops):
    """rpo_adam_step_multi (gridDim.y = slice): actor Adam with two Polyak targets (the second over a prefix: the shared
    embedding's copy in the critic target) | DualAdam | Polyak-only slice, bitwise equal to rpo_adam_step / rpo_polyak."""
    rng = np.random.RandomState(5)

    def fresh():
        st = {}
        rs = np.random.RandomState(6)
        for name, n in (("p", 34180), ("d", 6), ("c", 33796)):
            st[name] = dict(param=dev(rs.randn(n).astype(np.float32) * 0.1), m=torch.zeros(n, device=DEV),
                            v=torch.zeros(n, device=DEV), step=torch.zeros(ops.CONST["RPO_ADAM_STATE_LEN"], dtype=torch.int32, device=DEV),
                            gmax=torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV))
        st["p"]["target"] = dev(rs.randn(34180).astype(np.float32))
        st["p"]["target2"] = dev(rs.randn(768).astype(np.float32))
        st["c"]["target"] = dev(rs.randn(33796).astype(np.float32))
        return st
    a, b = fresh(), fresh()
    for it in range(3):
        gp, gd = rng.randn(34180).astype(np.float32), rng.randn(6).astype(np.float32)
        for st, multi in ((a, False), (b, True)):
            grad_p, grad_d = dev(gp), dev(gd)
            ops.absmax(grad_p, st["p"]["gmax"])
            if not multi:
                ops.adam_step(st["p"]["param"], grad_p, st["p"]["m"], st["p"]["v"], st["p"]["step"], 1e-4, clip_thres=0.2,
                              gradmax=st["p"]["gmax"], target=st["p"]["target"], tau=0.005, zero_grad=True)
                ops.polyak(st["p"]["param"][:768], st["p"]["target2"], 0.005)
                ops.adam_step(st["d"]["param"], grad_d, st["d"]["m"], st["d"]["v"], st["d"]["step"], 0.2, maximize=True,
                              clamp_min0=True, zero_grad=True)
                ops.polyak(st["c"]["param"], st["c"]["target"], 0.005)
            else:
                ops.adam_step_multi([
                    dict(param=st["p"]["param"], grad=grad_p, exp_avg=st["p"]["m"], exp_avg_sq=st["p"]["v"],
                         step_dev=st["p"]["step"], lr=1e-4, clip_thres=0.2, gradmax=st["p"]["gmax"], target=st["p"]["target"],
                         tau=0.005, zero_grad=True, target2=st["p"]["target2"], n2=768),
                    dict(param=st["d"]["param"], grad=grad_d, exp_avg=st["d"]["m"], exp_avg_sq=st["d"]["v"],
                         step_dev=st["d"]["step"], lr=0.2, maximize=True, clamp_min0=True, zero_grad=True,
                         gradmax=st["d"]["gmax"]),
                    dict(polyak_only=True, param=st["c"]["param"], target=st["c"]["target"], tau=0.005)])
            assert float(grad_p.abs().max()) == 0.0 and float(grad_d.abs().max()) == 0.0      # consumed
    for name in a:
        for key in a[name]:
            assert torch.equal(a[name][key], b[name][key]), (name, key)
    assert int(b["p"]["step"][0]) == 3 and int(b["d"]["step"][0]) == 3 and int(b["p"]["step"][2]) == 0


def test_xcc_probe_and_front_placement(ops):
    """rpo_xcc_probe reports the XCD of every workgroup; the fused front launches (rpo_split_critic_front*) are
    used only where every workgroup of a row tile -- in all planes -- shares one.  On an 8-XCD MI355X the dispatcher deals
    workgroups round-robin in block order: 16 row tiles (batch 256) keep a tile on XCD tile % 8, 7 row tiles (batch 100) do not."""
    import ctypes
    from rpo_amd import _lib
    out = torch.full((5, 16, 8), -1, dtype=torch.int32, device="cuda")
    rc = _lib.load().rpo_xcc_probe(8, 16, 5, 256, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    ids = out.cpu().view(5, 128)
    n_xcd = int(ids.max()) + 1
    assert int(ids.min()) >= 0 and n_xcd in (1, 2, 4, 8)
    assert ops.front_launch_ok(256, False) and ops.front_launch_ok(256, True)
    if n_xcd == 8:
        lin = torch.arange(128)
        assert all(len(torch.unique(ids[:, lin % 16 == t])) == 1 for t in range(16))
        assert not ops.front_launch_ok(100, False)              # 7 tiles: a tile's column groups land on different XCDs


@pytest.mark.parametrize("n", [257, 5000, 300000])
def test_lagrangian_beyond_one_workgroup_is_deterministic(ops, n):
    """Batches beyond 256 rows (the large-batch update mode): the per-workgroup sums of rpo_*_lagrangian meet in a scratch area
    (the head of the grad_action buffer) and ONE workgroup adds them in a fixed order before the elementwise launch overwrites
    it -- two calls give the same bits (round 3 added the partials with float atomics in arrival order) and match the oracle;
    without a grad_action buffer one workgroup recomputes and sums the rows (same property)."""
    rng = np.random.RandomState(n)
    c = cs.Constants(1)
    kc = cart_kernels(ops, 1)
    kp = ops.PendulumKernels()
    a_np = (rng.randn(n, 2) * 9.0).astype(np.float32)
    a = dev(a_np)
    for k, nu_np, oracle in ((kc, np.array([0.3, 0.0, 1.5, 0.2, 0.7, 0.05], np.float32),
                              lambda: train_ops.lagrangian_cart(a_np, np.array([0.3, 0.0, 1.5, 0.2, 0.7, 0.05], np.float32), c, 1.0 / n)),
                             (kp, np.array([0.37], np.float32),
                              lambda: train_ops.lagrangian_pendulum(a_np, np.array([0.37], np.float32), 1.0 / n))):
        outs = []
        for with_ga in (True, True, False):
            loss = torch.zeros(1, device=DEV)
            g_a = torch.full((n, 2), float("nan"), device=DEV) if with_ga else None
            g_nu = torch.zeros(len(nu_np), device=DEV)
            k.lagrangian(a, dev(nu_np), 1.0 / n, loss, g_a, g_nu)
            outs.append((loss.clone(), g_nu.clone(), g_a))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
        w_loss, w_ga, w_gnu = oracle()
        for loss, g_nu, g_a in outs:
            np.testing.assert_allclose(float(loss), w_loss, rtol=2e-5)
            np.testing.assert_allclose(g_nu.cpu().numpy(), w_gnu, rtol=2e-5, atol=1e-7)
            if g_a is not None:
                np.testing.assert_allclose(g_a.cpu().numpy(), w_ga, rtol=1e-5, atol=1e-8)
