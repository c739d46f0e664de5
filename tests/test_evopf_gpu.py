"""EVOPF-v0 HIP kernels (one wavefront per lane) against oracle/evopf.py and the reference cross-check fixtures.

Tolerances (float32 kernels vs float64 oracle, per-unit quantities of order 1): observations / residuals 2e-5;
Newton-solved actions 5e-5; products with the inverse of the 28x28 Jacobian block (condition ~1e3) 2e-3 of the largest
entry; Philox-driven episode data goes through logf / tanf(acosf()) / Box-Muller: 1e-5 relative, and 2e-5 absolute because tan(acos(pf)) is ill-conditioned for a power factor -> 1.
"""
import os

import numpy as np
import pytest
import torch

from oracle import evopf as oe

pytestmark = pytest.mark.gpu
G = oe.GRID
HERE = os.path.dirname(os.path.abspath(__file__))


def golden(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.fixture(scope="module")
def env():
    from rpo_amd.env import EVOPFEnv
    assert torch.cuda.is_available()
    return EVOPFEnv(device="cuda")


def dev(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x), dtype=dtype, device="cuda")


def states(n, seed=3):
    ids = np.arange(n)
    rng = np.random.RandomState(seed)
    hours = rng.randint(0, 24, size=n)
    s = np.concatenate([oe.episode_demand(seed, ids, 1, hours), rng.uniform(0.1, 0.8, size=(n, 5)),
                        oe.episode_price(seed, ids, 1, hours)], axis=1)
    return s.astype(np.float32), rng


def partials(s, rng):
    low, high = oe.partial_box(s.astype(np.float64))
    ap = low + rng.uniform(0.05, 0.95, size=low.shape) * (high - low)
    ap[:, :4] = rng.uniform(0.0, 0.6, size=(s.shape[0], 4))
    ap[:, 4:9] = rng.uniform(1.0, 1.06, size=(s.shape[0], 5))
    return ap.astype(np.float32)


def test_constants_table(env):
    t = env._table
    from rpo_amd import ops
    C = ops.CONST
    np.testing.assert_allclose(t[C["RPO_EVOPF_C_YR"]:C["RPO_EVOPF_C_YR"] + 196].reshape(14, 14), G.Yr, atol=1e-6)
    np.testing.assert_allclose(t[C["RPO_EVOPF_C_YI"]:C["RPO_EVOPF_C_YI"] + 196].reshape(14, 14), G.Yi, atol=2e-6)
    np.testing.assert_array_equal(env.partial_actions, G.partial_actions)
    np.testing.assert_array_equal(env.other_vars, G.other_vars)
    np.testing.assert_allclose(env.action_space.low, G.action_low, atol=1e-6)
    assert (env.state_dim, env.action_dim, env.eq_num, env.ineq_num, env.volatile) == (57, 43, 28, 58, True)


def test_reset_matches_oracle(env):
    n, seed = 300, 77
    v = env.make_vec(n, seed=seed, env_id_base=5)
    v.ep_count.copy_(torch.arange(n, dtype=torch.int32, device="cuda") % 7)
    v.reset()
    want = oe.reset(seed, 5 + np.arange(n), np.arange(n) % 7)
    np.testing.assert_allclose(v.obs.cpu().numpy(), want, rtol=1e-5, atol=2e-5)
    assert int(v.ep_len.abs().sum()) == 0


def test_resid_and_box_match_oracle(env):
    s, rng = states(64)
    a = (oe.complete_partial(s.astype(np.float64), partials(s, rng).astype(np.float64)) + 0.02 * rng.randn(64, 43)).astype(np.float32)
    eq = env.eq_resid(dev(s), dev(a)).cpu().numpy()
    ineq = env.ineq_resid(dev(s), dev(a)).cpu().numpy()
    np.testing.assert_allclose(eq, oe.eq_resid(s.astype(np.float64), a.astype(np.float64)), atol=2e-5)
    np.testing.assert_allclose(ineq, oe.ineq_resid(s.astype(np.float64), a.astype(np.float64)), atol=2e-6)
    lo, hi = env.update(dev(s))
    wl, wh = oe.partial_box(s.astype(np.float64))
    np.testing.assert_allclose(lo.cpu().numpy(), wl, atol=1e-6)
    np.testing.assert_allclose(hi.cpu().numpy(), wh, atol=1e-6)
    lo2, hi2 = env.update(s)
    np.testing.assert_allclose(lo2, wl, atol=1e-6)


def test_equation_solver_matches_oracle_and_reference(env):
    s, rng = states(200)
    ap = partials(s, rng)
    a = env.complete_partial(dev(s), dev(ap)).cpu().numpy()
    want = oe.complete_partial(s.astype(np.float64), ap.astype(np.float64))
    np.testing.assert_allclose(a, want, atol=5e-5)
    assert np.abs(oe.eq_resid(s.astype(np.float64), a.astype(np.float64))).max() < 2e-5
    fx = golden("evopf_env")                                    # the reference's own float32 results
    a = env.complete_partial(dev(fx["S"]), dev(fx["AP"])).cpu().numpy()
    np.testing.assert_allclose(a, fx["A"], atol=5e-5)


def test_equation_solver_backward(env):
    fx = golden("evopf_env")
    ap = dev(fx["AP"]).requires_grad_(True)
    a = env.complete_partial(dev(fx["S"]), ap)
    a.backward(dev(fx["DY"]))
    ref = fx["DZ"]
    np.testing.assert_allclose(ap.grad.cpu().numpy(), ref, atol=1e-3 * np.abs(ref).max())
    s, rng = states(40, seed=9)
    z = partials(s, rng)
    w = rng.randn(40, 43).astype(np.float32)
    ap = dev(z).requires_grad_(True)
    env.complete_partial(dev(s), ap).backward(dev(w))
    act, jac, jn, _ = oe.complete_partial(s.astype(np.float64), z.astype(np.float64), return_aux=True)
    want = oe.complete_partial_bwd(w.astype(np.float64), jac, jn)
    np.testing.assert_allclose(ap.grad.cpu().numpy(), want, atol=1e-3 * np.abs(want).max())


def test_ineq_partial_grad_matches_oracle_and_reference(env):
    fx = golden("evopf_env")
    got = env.ineq_partial_grad(dev(fx["S"]), dev(fx["AX"])).cpu().numpy()
    ref = fx["ineq_partial_grad"]
    np.testing.assert_allclose(got, ref, atol=2e-3 * np.abs(ref).max())
    s, rng = states(100, seed=4)
    a = (oe.complete_partial(s.astype(np.float64), partials(s, rng).astype(np.float64)) + 0.03 * rng.randn(100, 43)).astype(np.float32)
    got = env.ineq_partial_grad(dev(s), dev(a)).cpu().numpy()
    want = oe.ineq_partial_grad(s.astype(np.float64), a.astype(np.float64))
    np.testing.assert_allclose(got, want, atol=2e-3 * np.abs(want).max())


@pytest.mark.parametrize("tag", ["script", "large"])
def test_projection_matches_reference_fixture(env, tag):
    fx = golden("evopf_project")
    lr = float(fx[tag + "_lr"])
    a = env.project(dev(fx["S"]), dev(fx["AP"]), 10, lr).cpu().numpy()
    np.testing.assert_allclose(a, fx[tag + "_train"], atol=1e-4)
    a, it = env.project(dev(fx["S"]), dev(fx["AP"]), 50, lr, return_iters=True)
    np.testing.assert_allclose(a.cpu().numpy(), fx[tag + "_eval"], atol=5e-4)
    np.testing.assert_array_equal(it.cpu().numpy(), fx[tag + "_eval_iters"])


def test_projection_matches_oracle(env):
    s, rng = states(96, seed=12)
    ap = partials(s, rng)
    ap[:48, 4:9] = 1.07
    _, hi = oe.partial_box(s.astype(np.float64))
    ap[48:, 9:] = (hi[48:, 9:] + 0.05).astype(np.float32)
    a, it = env.project(dev(s), dev(ap), 10, 1e-4, return_iters=True)
    want, wit = oe.project(s.astype(np.float64), ap.astype(np.float64), 10, 1e-4)
    np.testing.assert_allclose(a.cpu().numpy(), want, atol=1e-4)
    assert (it.cpu().numpy() == wit).mean() > 0.95           # the 1e-5 stop test can flip on float32 noise


def test_env_constraint_api_matches_reference(env):
    """eq_jac / ineq_jac / eq_grad / ineq_grad / ineq_grad_new / ineq_dist_np / eq_resid_np (evopf.py:564-594,614-707)
    through rpo_evopf_eq_vjp / rpo_evopf_resid against the reference's own outputs."""
    from test_evopf_oracle import check_constraint_api
    check_constraint_api(env, golden("evopf_env"), dev)


def test_static_elimination_order_equals_partial_pivoting(env, monkeypatch):
    """The solver eliminates in a compiled-in static order with case14's sparsity (csrc/evopf_dev.h) and falls back to
    partial pivoting per solve when its pivots fail the acceptance test; RPO_EVOPF_PIVOT=dynamic forces partial pivoting
    everywhere.  Both are backward-stable evaluations of the same inverse: equation solver, its backward, the reduced
    gradient and the whole projection agree to float32 round-off amplified by the Jacobian's condition (~3e2), on
    well-posed points AND on grossly perturbed ones where the static order hands over to the fallback (there the two
    runs execute the same code, so they agree bit for bit or both produce the same non-finite rows)."""
    from rpo_amd.env import EVOPFEnv
    assert env.kernels.static_order
    monkeypatch.setenv("RPO_EVOPF_PIVOT", "dynamic")
    dyn = EVOPFEnv(device="cuda")
    assert not dyn.kernels.static_order
    s, rng = states(256, seed=21)
    ap = partials(s, rng)
    a_s, a_d = (e.complete_partial(dev(s), dev(ap)).cpu().numpy() for e in (env, dyn))
    np.testing.assert_allclose(a_s, a_d, atol=2e-5)
    w = rng.randn(256, 43).astype(np.float32)
    grads = []
    for e in (env, dyn):
        z = dev(ap).requires_grad_(True)
        e.complete_partial(dev(s), z).backward(dev(w))
        grads.append(z.grad.cpu().numpy())
    np.testing.assert_allclose(grads[0], grads[1], atol=2e-5 * np.abs(grads[1]).max())
    ax = (a_d + 0.03 * rng.randn(256, 43)).astype(np.float32)
    g_s, g_d = (e.ineq_partial_grad(dev(s), dev(ax)).cpu().numpy() for e in (env, dyn))
    np.testing.assert_allclose(g_s, g_d, atol=2e-5 * np.abs(g_d).max())
    ap2 = ap.copy()
    ap2[:128, 4:9] = 1.07                                       # violated bounds: all 10 GRG iterations
    (p_s, it_s), (p_d, it_d) = (e.project(dev(s), dev(ap2), 10, 1e-4, return_iters=True) for e in (env, dyn))
    np.testing.assert_allclose(p_s.cpu().numpy(), p_d.cpu().numpy(), atol=2e-5)
    assert (it_s == it_d).float().mean() > 0.98
    # near-singular Jacobians (voltages far off any operating point): the acceptance test must hand these to the fallback
    bad = a_d.copy()
    bad[:, G.vm0:G.vm0 + 14] += 0.25 * rng.randn(256, 14).astype(np.float32)
    bad[:, G.va0:G.va0 + 14] += 0.8 * rng.randn(256, 14).astype(np.float32)
    g_s, g_d = (e.ineq_partial_grad(dev(s), dev(bad)).cpu().numpy() for e in (env, dyn))
    want = oe.ineq_partial_grad(s.astype(np.float64), bad.astype(np.float64))
    scale = np.abs(want).max(axis=1, keepdims=True)
    err_s, err_d = (np.abs(g - want) / scale for g in (g_s, g_d))
    assert np.nanmax(err_s) <= max(4.0 * np.nanmax(err_d), 2e-3), (np.nanmax(err_s), np.nanmax(err_d))


def test_diverging_newton_input_stays_finite(env):
    """The recorded basic action for which Newton has no solution to find (tests/golden/evopf_newton_divergence.npz; the float64
    oracle does not converge on it either, test_evopf_oracle.py): the kernels return garbage there like the reference would,
    but FINITE garbage -- a NaN action would poison the replay ring and, one update later, every parameter."""
    fx = golden("evopf_newton_divergence")
    s = dev(fx["s"][None])
    ap = dev(fx["a"][None][:, G.partial_actions])
    a = env.complete_partial(s, ap)
    p, it = env.project(s, ap, 10, 1e-4, return_iters=True)
    g = env.ineq_partial_grad(s, a.detach())
    assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(p).all()) and bool(torch.isfinite(g).all())
    assert float(env.eq_resid(s, a.detach()).abs().max()) > 0.1      # (not converged: that is the point of the fixture)


def test_step_matches_oracle_including_episode_end(env):
    n, seed = 128, 31
    ids = np.arange(n)
    v = env.make_vec(n, seed=seed)
    v.reset()
    rows = torch.zeros(3 * n, 248, device="cuda")
    rng = np.random.RandomState(1)
    s = v.obs.cpu().numpy().astype(np.float64)
    length, count = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    c = env.kernels.cols
    for t in range(26):
        ap = partials(s.astype(np.float32), rng)
        a = oe.complete_partial(s, ap.astype(np.float64))
        a[:, 38:] += rng.uniform(-0.2, 0.2, size=(n, 5))
        a = a.astype(np.float32)
        want = oe.step(s, a.astype(np.float64), length, count, seed, ids)
        v.step(dev(a), rows=rows, cap_steps=3, auto_reset=True)
        row = rows[(t % 3) * n:(t % 3 + 1) * n].cpu().numpy()
        np.testing.assert_allclose(row[:, c["state"][0]:c["state"][1]], s, atol=1e-6)
        np.testing.assert_array_equal(row[:, c["action"][0]:c["action"][1]], a)
        np.testing.assert_allclose(row[:, c["next_state"][0]:c["next_state"][1]], want["next_state"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(row[:, c["reward"][0]], want["reward"], rtol=2e-5, atol=2e-5)
        np.testing.assert_array_equal(row[:, c["done"][0]] > 0.5, want["done"])
        np.testing.assert_allclose(row[:, c["eq_viol"][0]:c["eq_viol"][1]], want["eq_viol"], atol=2e-5)
        np.testing.assert_allclose(row[:, c["ineq_viol"][0]:c["ineq_viol"][1]], want["ineq_viol"], atol=2e-6)
        assert want["done"].all() == (t == 23)
        s, length, count = want["state"], want["ep_len"], want["ep_count"]
        np.testing.assert_allclose(v.obs.cpu().numpy(), s, rtol=1e-5, atol=2e-5)
        s = v.obs.cpu().numpy().astype(np.float64)              # continue from the device state (no drift)
        np.testing.assert_array_equal(v.ep_len.cpu().numpy(), length)
        np.testing.assert_array_equal(v.ep_count.cpu().numpy(), count)
    assert int(v.ctrl[0]) == 26
    from rpo_amd import ops
    st = ops.reduce_stats(v.stats[:26]).cpu().numpy()
    assert st[23, ops.STAT["episodes"]] == n and st[:23, ops.STAT["episodes"]].sum() == 0
    assert st[23, ops.STAT["length_sum"]] == 24 * n


def test_step_matches_reference_fixture(env):
    fx = golden("evopf_step")
    n = len(fx["hour"])
    k = env.kernels
    state = dev(fx["state"])
    ep_len = dev(fx["hour"], torch.int32)
    ep_ret = torch.zeros(n, device="cuda")
    ep_count = torch.full((n,), 2, dtype=torch.int32, device="cuda")
    rows = torch.zeros(n, 248, device="cuda")
    ctrl = torch.zeros(288, dtype=torch.int64, device="cuda")
    k.step(state, state, dev(fx["action"]), ep_len, ep_ret, ep_count, rows, 1, None, ctrl, 2 ** 31 - 1, False, 1e-3,
           int(fx["seed"]), 0)
    row, c = rows.cpu().numpy(), k.cols
    np.testing.assert_allclose(row[:, c["next_state"][0]:c["next_state"][1]], fx["next_state"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(row[:, c["reward"][0]], fx["reward"], rtol=2e-5, atol=2e-5)
    np.testing.assert_array_equal(row[:, c["done"][0]] > 0.5, fx["done"])
    np.testing.assert_allclose(row[:, c["eq_viol"][0]:c["eq_viol"][1]], fx["eq_viol"], atol=2e-5)
    np.testing.assert_allclose(row[:, c["ineq_viol"][0]:c["ineq_viol"][1]], fx["ineq_viol"], atol=2e-6)


def test_equation_solver_backward_adds_its_terms_in_the_kernel(env):
    """rpo_evopf_complete_bwd(grad_action_b, grad_action2): dL/dy = (grad_action + grad_action_b) + grad_action2 inside the kernel
    == the same sums made by elementwise launches first, bit for bit (ragged n: the last workgroup is partly empty); and
    rpo_min_q_bwd == torch.min's backward of -min(q1, q2) / B, ties split."""
    from rpo_amd import ops
    s, rng = states(37, seed=12)
    a = dev(oe.complete_partial(s.astype(np.float64), partials(s, rng).astype(np.float64)).astype(np.float32))
    g1, g2, g3 = (dev(rng.randn(37, 43).astype(np.float32)) for _ in range(3))
    k = env.kernels
    want, got = torch.zeros(37 * 14, device="cuda"), torch.zeros(37 * 14, device="cuda")
    k.complete_bwd(None, (g1 + g2) + g3, want, action=a)
    k.complete_bwd(None, g1, got, action=a, grad_action_b=g2, grad_action2=g3)
    assert torch.equal(want, got) and float(got.abs().max()) > 0
    k.complete_bwd(None, g1 + g3, want, action=a)
    k.complete_bwd(None, g1, got, action=a, grad_action2=g3)
    assert torch.equal(want, got)
    q1 = dev(rng.randn(300, 1).astype(np.float32))
    q2 = q1.clone()
    q2[::3] += 0.5
    q2[1::3] -= 0.5                                                 # a third each: q1 smaller, q2 smaller, equal
    dq1, dq2 = torch.empty(300, 1, device="cuda"), torch.empty(300, 1, device="cuda")
    ops.min_q_bwd(q1, q2, -1.0 / 300, dq1, dq2)
    a1, a2 = q1.clone().requires_grad_(True), q2.clone().requires_grad_(True)
    (-torch.min(a1, a2)).mean().backward()
    assert torch.equal(dq1, a1.grad) and torch.equal(dq2, a2.grad)


def test_lagrangian(env):
    s, rng = states(256, seed=6)
    a = (oe.complete_partial(s.astype(np.float64), partials(s, rng).astype(np.float64)) + 0.05 * rng.randn(256, 43)).astype(np.float32)
    nu = rng.uniform(0, 1, size=58).astype(np.float32)
    loss = torch.zeros(1, device="cuda")
    g_a = torch.empty(256, 43, device="cuda")
    g_nu = torch.zeros(58, device="cuda")
    env.kernels.lagrangian(dev(a), dev(nu), 1.0 / 256, loss, g_a, g_nu, obs=dev(s))
    dist = oe.ineq_dist(s.astype(np.float64), a.astype(np.float64))
    np.testing.assert_allclose(float(loss), (dist @ nu).mean(), rtol=1e-5)
    np.testing.assert_allclose(g_nu.cpu().numpy(), dist.mean(axis=0), rtol=1e-5, atol=1e-8)
    want = ((dist > 0) * nu) @ oe.ineq_jac() / 256
    np.testing.assert_allclose(g_a.cpu().numpy(), want, rtol=1e-5, atol=1e-9)


def test_gym_api_runs_one_day(env):
    from rpo_amd.env.base import gym
    e = gym.make("EVOPF-v0")
    obs = e.reset()
    assert obs.shape == (57,) and np.allclose(obs[28:33], 0.2)
    rng = np.random.RandomState(0)
    total, steps, done = 0.0, 0, False
    while not done:
        ap = partials(obs[None].astype(np.float32), rng)
        a = e.complete_partial(torch.tensor(obs[None], dtype=torch.float32), torch.tensor(ap)).cpu().numpy()[0]
        obs, r, done, info = e.step(a)
        assert info["eq_viol"].shape == (1, 28) and info["ineq_viol"].shape == (1, 58)
        assert np.abs(info["eq_viol"]).max() < 2e-5
        total += r
        steps += 1
    assert steps == 24 and (obs[:28] == 0).all()
    obs2 = e.reset()
    assert not np.allclose(obs2[:28], 0)
    e.close()


# ------------------------------------------------------------------------------------------------ trainer
def _golden(name):
    return golden(name)


@pytest.mark.parametrize("envname,fused", [("evopf", False), ("evopf256", True), ("evopf256", False)],
                         ids=["torch_mlp_64", "fused_mlp_256", "torch_mlp_256"])
def test_update_matches_reference_on_gpu(envname, fused):
    """RPODDPG.train for t = 1..4 on EVOPF (critic steps, one policy + multiplier step, Polyak) against the fixtures
    recorded from the reference (make_evopf_golden.gen_train_steps), every random draw replayed; through the hand-written
    MLP kernels (multi-output MFMA head) or the torch modules.  Tolerance: float32 GEMMs + 4 Adam steps 5e-6 on
    parameters; the actor gradient passes through the 22x22 Newton inverse -> 2e-5."""
    import test_train_step_golden as tsg
    from rpo_amd import ops
    out = tsg.run_product_update(_golden, "ddpg", envname, ops, torch.device("cuda"), fused=fused)
    tsg.check_evopf_update(*out)


@pytest.mark.parametrize("fused", [True, False], ids=["fused_mlp_256", "torch_mlp_256"])
def test_sac_update_matches_reference_on_gpu(fused):
    """RPOSAC.train for t = 1..4 on EVOPF (scripts/evopf_exp_sac.py hyper-parameters; 14-dimensional squashed-Gaussian head
    in the state-dependent box) against the reference fixture, every random draw replayed."""
    import test_train_step_golden as tsg
    from rpo_amd import ops
    out = tsg.run_product_update(_golden, "sac", "evopf256", ops, torch.device("cuda"), fused=fused)
    tsg.check_evopf_update(*out, algo="sac")


def _run(n_envs, iters, use_graph, seed_all=5, algo="ddpg", fused=False, **extra):
    import test_train_step_golden as tsg
    from rpo_amd import ops
    torch.manual_seed(seed_all)
    tr = tsg.build_trainer(algo, "evopf256" if fused else "evopf", ops, torch.device("cuda"), fused=fused,
                           num_envs=n_envs, use_graph=use_graph, **extra)
    tr.vec.reset()
    tr.run_steps(iters)
    torch.cuda.synchronize()
    return tr


_SAC_KW = dict(lr_actor=1e-4, lr_critic=3e-4, grad_eps=0.1, init_lamb=0.0, init_nju=0.0, alpha=0.001,
               automatic_entropy_tuning=False, fixed=False)            # scripts/evopf_exp_sac.py:30-33


@pytest.mark.parametrize("fused,algo", [(True, "ddpg"), (False, "ddpg"), (True, "sac")], ids=["fused_mlp", "torch_mlp", "fused_mlp_sac"])
def test_training_iterations_graph_equals_eager_and_episodes_roll_over(fused, algo, monkeypatch):
    """30 iterations through hipGraph windows of one policy_fre period (RPO_GRAPH_CYCLE=4: eager passes on the side stream, the
    capture at the fourth window, replays, a ragged tail) == 30 eager iterations, bit for bit.  The windows of this workload
    have TWO captured branches: rollout t+1 beside update t, and (round 5, fused networks) the actor-only prefix of the policy
    step -- pi(s), noise, Complete, Lagrangian -- beside the critic update of a policy iteration; the serial order
    (`_policy_prefix_enabled = False`) gives the same bits again.  Since the re-capture of round 5 (the update chain first, the
    second branch behind an event: DESIGN 4b) this is also the test that the event form carries the same dependencies; RPOSAC
    (its prefix draws into a buffer of its own) goes through the same windows."""
    from rpo_amd.algo.trainer import RPOTrainerBase
    monkeypatch.setenv("RPO_GRAPH_CYCLE", "4")
    kw = dict(algo=algo, **(_SAC_KW if algo == "sac" else {}))
    a = _run(64, 30, use_graph=False, fused=fused, **kw)
    b = _run(64, 30, use_graph=True, fused=fused, **kw)
    key = ("cycle", 4, True, "overlap") if fused else ("cycle", 4, True)      # (torch modules: no update clock, serial windows)
    assert b._graphs.entries[key]["graph"] is not None and not b._graphs.capture_failed
    assert b._policy_prefix_ok() == fused
    monkeypatch.setattr(RPOTrainerBase, "_policy_prefix_enabled", False, raising=False)
    c = _run(64, 30, use_graph=True, fused=fused, **kw)
    assert not c._policy_prefix_ok() and c._graphs.entries[key]["graph"] is not None
    monkeypatch.undo()
    for other in (b, c):
        assert torch.equal(a.vec.internal, other.vec.internal) and torch.equal(a.buffer.rows, other.buffer.rows)
        assert torch.equal(a.agent.flat.data, other.agent.flat.data) and torch.equal(a.agent.nju.weight, other.agent.nju.weight)
        assert torch.equal(a.agent.critic_target_flat, other.agent.critic_target_flat)
    assert int(a.vec.ctrl[0]) == 30 and (a.vec.ep_count == 1).all() and (a.vec.ep_len == 6).all()
    a._harvest()
    assert a.env_steps == 64 * 30 and 0.0 <= a.viol_rate <= 1.0
    res = a.eval()
    assert len(res) == 10 and np.isfinite(res).all()


@pytest.mark.parametrize("lanes,extra,prefix", [(256, {}, True), (64, dict(fixed=True), False)], ids=["lanes_equal_batch", "fixed_multipliers"])
def test_policy_prefix_branch_edge_cases(lanes, extra, prefix, monkeypatch):
    """ADVICE r05 on the actor-only prefix branch: (i) with num_envs == batch_size the rollout's actor output and the
    prefix's pi(s) would be ONE shape-keyed scratch buffer written from two streams -- the prefix now has a buffer of its own
    ("pi.out"); (ii) with `fixed=True` (multipliers not stepped) the prefix only zeroed nu's gradient beside the critic update:
    the serial order is kept there.  hipGraph windows == eager launches, bit for bit, in both."""
    monkeypatch.setenv("RPO_GRAPH_CYCLE", "4")
    a = _run(lanes, 22, use_graph=False, fused=True, **extra)
    b = _run(lanes, 22, use_graph=True, fused=True, **extra)
    assert b._policy_prefix_ok() == prefix and not b._graphs.capture_failed
    assert torch.equal(a.vec.internal, b.vec.internal) and torch.equal(a.buffer.rows, b.buffer.rows)
    assert torch.equal(a.agent.flat.data, b.agent.flat.data) and torch.equal(a.agent.nju.weight, b.agent.nju.weight)
    if prefix:
        assert ("pi.out", 256, 14) in b.fused._scratch and b.fused._scratch[("pi.out", 256, 14)] is not b.fused._scratch[("actor.out", 256, 14)]


def test_sac_on_evopf_runs_and_replays(monkeypatch):
    """scripts/evopf_exp_sac.py's configuration: RPOSAC with a 14-dimensional squashed-Gaussian policy and the
    state-dependent box; hipGraph replay (one graph per iteration) equals the eager run, stored transitions stay near the
    equality manifold."""
    kw = _SAC_KW
    monkeypatch.setenv("RPO_GRAPH_CYCLE", "1")
    a = _run(32, 12, use_graph=False, algo="sac", **kw)
    b = _run(32, 12, use_graph=True, algo="sac", **kw)
    assert any(e["graph"] is not None for e in b._graphs.entries.values())
    assert torch.equal(a.vec.internal, b.vec.internal) and torch.equal(a.agent.flat.data, b.agent.flat.data)
    c = a.kernels.cols
    rows = a.buffer.rows[:12 * 32]
    # GRG steps leave the manifold at second order (and at first order along pe, hazard E1): small, not zero
    assert float(rows[:, c["eq_viol"][0]:c["eq_viol"][1]].abs().max()) < 1e-2
    lo, hi = a.base_env.update(rows[:, c["state"][0]:c["state"][1]].contiguous())
    ap = rows[:, c["action"][0]:c["action"][1]][:, a.base_env.partial_actions]
    assert bool((ap[:, :4] >= lo[:, :4] - 1e-3).all()) and bool((ap[:, :4] <= hi[:, :4] + 1e-3).all())


def test_fused_and_torch_mlp_paths_agree_on_evopf():
    """Same seeds, same Philox draws: the trainer on the hand-written MLP kernels and on the torch modules produce the
    same transitions and (to float32 summation order) the same parameters after 12 iterations."""
    a = _run(48, 12, use_graph=False, fused=True)
    b = _run(48, 12, use_graph=False, fused=False, embed_dim=256, hidden_dim=256)
    np.testing.assert_allclose(a.buffer.rows.cpu().numpy(), b.buffer.rows.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(a.agent.flat.data.cpu().numpy(), b.agent.flat.data.cpu().numpy(), rtol=0, atol=2e-5)


def test_sac_fused_and_torch_paths_agree_on_evopf(monkeypatch):
    """RPOSAC on EVOPF through the MLP kernels (2 x 14 MFMA heads) + rpo_evopf_gauss_head(_bwd) vs the torch modules:
    same Philox draws, same transitions, parameters equal to float32 summation order after 12 iterations; hipGraph
    replay of the fused path equals its eager run."""
    kw = dict(lr_actor=1e-4, lr_critic=3e-4, grad_eps=0.1, init_lamb=0.0, init_nju=0.1, alpha=0.001,
              automatic_entropy_tuning=False, fixed=False)
    a = _run(48, 12, use_graph=False, algo="sac", fused=True, **kw)
    b = _run(48, 12, use_graph=False, algo="sac", fused=False, embed_dim=256, hidden_dim=256, **kw)
    assert a.fused is not None and a.fused.descs["actor"].head_dim == 14
    np.testing.assert_allclose(a.buffer.rows.cpu().numpy(), b.buffer.rows.cpu().numpy(), rtol=0, atol=2e-4)
    np.testing.assert_allclose(a.agent.flat.data.cpu().numpy(), b.agent.flat.data.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(a.agent.nju.weight.detach().cpu().numpy(), b.agent.nju.weight.detach().cpu().numpy(), rtol=1e-3, atol=1e-6)
    # windows of one policy_fre period: rollout t+1 and the actor-only prefix of the policy step on the second captured branch
    monkeypatch.setenv("RPO_GRAPH_CYCLE", "4")
    c = _run(48, 24, use_graph=True, algo="sac", fused=True, **kw)
    d = _run(48, 24, use_graph=False, algo="sac", fused=True, **kw)
    assert c._graphs.entries[("cycle", 4, True, "overlap")]["graph"] is not None and c._policy_prefix_ok()
    assert torch.equal(d.agent.flat.data, c.agent.flat.data) and torch.equal(d.buffer.rows, c.buffer.rows)
    assert torch.equal(d.agent.nju.weight, c.agent.nju.weight)
