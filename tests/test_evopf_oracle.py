"""oracle/evopf.py: (1) self-consistency -- what pins the restatement in the absence of pypower (Newton converges,
finite differences agree with eq_jac and the PFFunction backward, the GRG direction is tangent to the equality manifold);
(2) cross-check against the reference's evopf.py run on the pypower stand-in (tests/golden/make_evopf_golden.py).

Tolerances: the reference computes in float32 torch, the oracle in float64: 2e-5 absolute on per-unit quantities, 2e-3
relative on Jacobian-inverse products (condition number of the 28x28 block ~1e3).
"""
import os

import numpy as np
import pytest

from oracle import evopf as oe

G = oe.GRID
HERE = os.path.dirname(os.path.abspath(__file__))


def golden(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.fixture(scope="module")
def env_fx():
    return golden("evopf_env")


def some_states(n, seed=3):
    ids = np.arange(n)
    rng = np.random.RandomState(seed)
    hours = rng.randint(0, 24, size=n)
    return np.concatenate([oe.episode_demand(seed, ids, 1, hours), rng.uniform(0.1, 0.8, size=(n, 5)),
                           oe.episode_price(seed, ids, 1, hours)], axis=1), rng


def some_partials(state, rng):
    low, high = oe.partial_box(state)
    ap = low + rng.uniform(0.05, 0.95, size=low.shape) * (high - low)
    ap[:, :4] = rng.uniform(0.0, 0.6, size=(state.shape[0], 4))
    ap[:, 4:9] = rng.uniform(1.0, 1.06, size=(state.shape[0], 5))
    return ap


# ------------------------------------------------------------------------------------------------ self-consistency
def test_dimensions_and_indices():
    assert (G.state_dim, G.action_dim, G.eq_num, G.ineq_num) == (57, 43, 28, 58)        # evopf.py:333-337
    assert len(G.partial_actions) == 14 and len(G.partial_vars) == 15 and len(G.other_vars) == 28
    assert list(G.slack) == [0] and list(G.pv) == [1, 2, 5, 7] and len(G.pq) == 9
    assert len(G.keep_constr) == 22 and len(G.newton_vars) == 22


def test_ybus_properties():
    y = oe.make_ybus()
    assert np.allclose(y, y.T)                               # no phase shifters in case14
    rows = np.abs(y.sum(axis=1))
    # rows sum to the shunt terms only (line charging, off-nominal taps, bus 9 capacitor)
    assert rows.max() < 0.3 and np.isclose(y[8, 8].imag - (y[8].sum() - y[8, 8]).imag * -1, y[8].sum().imag)


def test_power_flow_known_answer():
    """Known-answer test for the case data + Ybus + Newton solver: the published power-flow solution of the IEEE 14-bus
    case (the `runpf(case14)` printout of MATPOWER / PYPOWER, 3 decimals) with the case's own loads and set-points."""
    s = np.zeros((1, 57))
    s[0, :14], s[0, 14:28] = oe.BUS[:, oe.PD] / 100, oe.BUS[:, oe.QD] / 100
    ap = np.zeros((1, 14))
    ap[0, :4], ap[0, 4:9] = oe.GEN[1:, oe.GEN_PG] / 100, oe.GEN[:, 5]
    pg, qg, vm, va, _ = oe.split(oe.complete_partial(s, ap))
    vm_pub = [1.060, 1.045, 1.010, 1.018, 1.020, 1.070, 1.062, 1.090, 1.056, 1.051, 1.057, 1.055, 1.050, 1.036]
    va_pub = [0.0, -4.983, -12.725, -10.313, -8.774, -14.221, -13.360, -13.360, -14.939, -15.097, -14.791, -15.076,
              -15.156, -16.034]
    np.testing.assert_allclose(vm[0], vm_pub, atol=5.1e-4)
    np.testing.assert_allclose(np.rad2deg(va[0]), va_pub, atol=5.1e-4)
    np.testing.assert_allclose(pg[0] * 100, [232.39, 40.0, 0.0, 0.0, 0.0], atol=5.1e-3)
    np.testing.assert_allclose(qg[0] * 100, [-16.55, 43.56, 25.08, 12.73, 17.62], atol=5.1e-3)


def test_newton_converges_and_zeroes_the_equations():
    S, rng = some_states(6)
    ap = some_partials(S, rng)
    a, jac, jn, iters = oe.complete_partial(S, ap, return_aux=True)
    assert iters.max() <= 8
    assert np.abs(oe.eq_resid(S, a)).max() < 1e-9
    np.testing.assert_allclose(a[:, G.partial_actions], ap)


def test_eq_jac_matches_finite_differences():
    S, rng = some_states(3)
    a = oe.complete_partial(S, some_partials(S, rng)) + 0.01 * rng.randn(3, 43)
    jac = oe.eq_jac(a)
    h = 1e-6
    for k in range(43):
        e = np.zeros(43)
        e[k] = h
        fd = (oe.eq_resid(S, a + e) - oe.eq_resid(S, a - e)) / (2 * h)
        # the battery columns carry the reference's sign: d(real)/d(pe) = -I at evopf.py:639-640 although eq_resid adds
        # +pe (:532); reproduced literally (hazard E1)
        np.testing.assert_allclose(jac[:, :, k] * (-1.0 if k >= G.pe0 else 1.0), fd, atol=1e-6)


def test_backward_matches_finite_differences():
    S, rng = some_states(2)
    ap = some_partials(S, rng)
    a, jac, jn, _ = oe.complete_partial(S, ap, return_aux=True, tol=1e-12)
    w = rng.randn(2, 43)
    dz = oe.complete_partial_bwd(w, oe.eq_jac(a), oe.eq_jac(a)[:, G.keep_constr][:, :, G.newton_vars])
    h = 1e-6
    for k in range(14):
        e = np.zeros((1, 14))
        e[0, k] = h
        fd = ((oe.complete_partial(S, ap + e, tol=1e-12) - oe.complete_partial(S, ap - e, tol=1e-12)) * w).sum(axis=1) / (2 * h)
        if k < 9:
            np.testing.assert_allclose(dz[:, k], fd, rtol=1e-5, atol=1e-6)
        else:
            # same sign convention in PFFunction.backward (:897-898): the implicit part of d/d(pe) is mirrored
            direct = w[:, G.pe0 + k - 9]
            np.testing.assert_allclose(2 * direct - dz[:, k], fd, rtol=1e-5, atol=1e-6)


def test_grg_direction_is_tangent_to_the_equalities():
    S, rng = some_states(4)
    ap = some_partials(S, rng)
    ap[:, 4:9] = 1.08                                        # above vmax -> violated inequalities
    a = oe.complete_partial(S, ap)
    d = oe.ineq_partial_grad(S, a)
    assert np.abs(d).max() > 0.5
    assert np.abs(np.einsum("nev,nv->ne", oe.eq_jac(a), d)).max() < 1e-9
    moved, iters = oe.grad_steps(S, a, 10, 1e-4)             # corr_lr of scripts/evopf_exp.py:29
    assert (iters == 10).all() and np.abs(moved - a).max() > 1e-4
    assert np.abs(oe.eq_resid(S, moved)).max() < 5e-3        # small drift off the manifold only


def test_episode_data_shapes_and_statistics():
    ids = np.arange(4000)
    d = oe.episode_demand(11, ids, 0, 7)
    assert d.shape == (4000, 28)
    ps = oe.CURVE_DEMAND / oe.CURVE_DEMAND.sum() * G.p_total
    np.testing.assert_allclose(d[:, :14].sum(axis=1) * 100, ps[7], rtol=1e-9)      # shares sum to one (demand.py:55-58)
    assert (d[:, [0, 6, 7]] == 0).all()                                           # buses without load
    assert (d[:, 14 + 3] <= 0).all() and (d[:, 14 + 1] >= 0).all()                # sign of the nominal Qd
    p = oe.episode_price(11, ids, 0, 20)
    assert p.shape == (4000, 24) and (p[:, 4:] == 0).all()
    np.testing.assert_allclose(p[:, 0].mean() * 100, oe.CURVE_PRICE[20], rtol=0.05)
    assert (oe.episode_price(11, ids, 0, 24) == 0).all() and (oe.episode_demand(11, ids, 0, 24) == 0).all()
    r = oe.reset(11, ids[:3], np.zeros(3, dtype=np.int64))
    assert r.shape == (3, 57) and np.allclose(r[:, 28:33], 0.2)


def test_episode_runs_24_steps_and_auto_resets():
    n, seed = 3, 4
    ids = np.arange(n)
    count, length = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    s = oe.reset(seed, ids, count)
    rng = np.random.RandomState(0)
    for t in range(24):
        a = oe.complete_partial(s, some_partials(s, rng))
        out = oe.step(s, a, length, count, seed, ids)
        assert out["done"].all() == (t == 23)
        s, length, count = out["state"], out["ep_len"], out["ep_count"]
    assert (count == 1).all() and (length == 0).all()
    np.testing.assert_allclose(s, oe.reset(seed, ids, count))
    assert (out["next_state"][:, :28] == 0).all() and (out["next_state"][:, 33:] == 0).all()


# ------------------------------------------------------------------------------------------------ reference cross-check
def test_constants_match_reference_on_standin(env_fx):
    np.testing.assert_allclose(G.Yr, env_fx["Yr"], atol=2e-6)
    np.testing.assert_allclose(G.Yi, env_fx["Yi"], atol=2e-6)
    for k in ("partial_actions", "partial_vars", "other_vars"):
        np.testing.assert_array_equal(getattr(G, k), env_fx[k])
    np.testing.assert_allclose(G.action_low, env_fx["action_low"], atol=1e-6)
    np.testing.assert_allclose(G.action_high, env_fx["action_high"], atol=1e-6)


def test_constraint_functions_match_reference(env_fx):
    S, AX = env_fx["S"].astype(np.float64), env_fx["AX"].astype(np.float64)
    np.testing.assert_allclose(oe.eq_resid(S, AX), env_fx["eq_resid"], atol=2e-5)
    np.testing.assert_allclose(oe.ineq_resid(S, AX), env_fx["ineq_resid"], atol=2e-6)
    np.testing.assert_allclose(oe.eq_jac(AX), env_fx["eq_jac"], atol=5e-5)
    np.testing.assert_allclose(oe.obj_fn(AX), env_fx["obj"], rtol=1e-5)
    low, high = oe.partial_box(S)
    np.testing.assert_allclose(low, env_fx["box_low"], atol=1e-6)
    np.testing.assert_allclose(high, env_fx["box_high"], atol=1e-6)
    ref = env_fx["ineq_partial_grad"]
    np.testing.assert_allclose(oe.ineq_partial_grad(S, AX), ref, atol=2e-3 * np.abs(ref).max())
    assert np.abs(ref).max() > 1.0
    np.testing.assert_allclose(oe.ineq_partial_grad(S, env_fx["A"].astype(np.float64)),
                               env_fx["ineq_partial_grad_feasible"], atol=2e-3 * max(1.0, np.abs(env_fx["ineq_partial_grad_feasible"]).max()))


def check_constraint_api(env, fx, t32):
    """The rest of the reference's constraint surface (evopf.py:564-594,614-707: eq_jac, ineq_jac, eq_grad, ineq_grad,
    ineq_grad_new, ineq_dist_np, eq_resid_np) on an EVOPFEnv, against what the unmodified reference returned for the same
    (S, AX) (tests/golden/make_evopf_golden.py gen_env).  Shared by the CPU leg (oracle backend) and the GPU leg."""
    S, AX = t32(fx["S"]), t32(fx["AX"])
    np.testing.assert_allclose(env.eq_jac(AX).cpu().numpy(), fx["eq_jac"], atol=5e-5)
    assert tuple(env.eq_jac(AX[3]).shape) == (1, 28, 43)
    np.testing.assert_array_equal(env.ineq_jac(S, AX).cpu().numpy(), fx["ineq_jac"])
    ref = fx["eq_grad"]
    np.testing.assert_allclose(env.eq_grad(S, AX).cpu().numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(env.ineq_grad(S, AX).cpu().numpy(), fx["ineq_grad"], atol=5e-6)
    np.testing.assert_allclose(env.ineq_grad(S, AX, 0.02).cpu().numpy(), fx["ineq_grad_eps"], atol=5e-6)
    np.testing.assert_array_equal(env.ineq_grad_new(S, AX).cpu().numpy(), fx["ineq_grad_new"])
    assert np.abs(fx["ineq_grad_new"]).sum() > 0 and np.abs(fx["ineq_grad"]).max() > 0
    d = env.ineq_dist_np(fx["S"][3], fx["AX"][3])
    assert d.shape == (1, 58)
    np.testing.assert_allclose(d, fx["ineq_dist_np"], atol=2e-6)
    r = env.eq_resid_np(fx["S"][3], fx["AX"][3])
    assert r.shape == (1, 28)
    np.testing.assert_allclose(r, fx["eq_resid_np"], atol=2e-5)


def test_env_constraint_api_matches_reference_on_oracle_backend(env_fx):
    import torch
    import oracle_backend as ob
    from rpo_amd.env import EVOPFEnv
    env = EVOPFEnv(backend=ob, device="cpu")
    check_constraint_api(env, env_fx, lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32))


def test_equation_solver_matches_reference(env_fx):
    S, AP = env_fx["S"].astype(np.float64), env_fx["AP"].astype(np.float64)
    a, jac, jn, _ = oe.complete_partial(S, AP, return_aux=True)
    np.testing.assert_allclose(a, env_fx["A"], atol=2e-5)
    np.testing.assert_allclose(a, env_fx["A_batch"], atol=2e-5)      # batch-wide stop test: extra iterations only
    dz = oe.complete_partial_bwd(env_fx["DY"].astype(np.float64), jac, jn)
    np.testing.assert_allclose(dz, env_fx["DZ"], atol=2e-4 * np.abs(env_fx["DZ"]).max())


def test_step_matches_reference():
    fx = golden("evopf_step")
    seed = int(fx["seed"])
    n = len(fx["hour"])
    ids = np.arange(n)
    out = oe.step(fx["state"], fx["action"].astype(np.float64), fx["hour"].astype(np.int64), np.full(n, 2, dtype=np.int64),
                  seed, ids, auto_reset=False)
    np.testing.assert_allclose(out["next_state"], fx["next_state"], atol=1e-9)
    np.testing.assert_allclose(out["reward"], fx["reward"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(out["done"], fx["done"])
    assert fx["done"].sum() == 2
    np.testing.assert_allclose(out["eq_viol"], fx["eq_viol"], atol=2e-5)
    np.testing.assert_allclose(out["ineq_viol"], fx["ineq_viol"], atol=2e-6)
    assert (fx["ineq_viol"][:, 48:] > 0).any()


@pytest.mark.parametrize("tag", ["script", "large"])
def test_projection_matches_reference(tag):
    fx = golden("evopf_project")
    S, AP = fx["S"].astype(np.float64), fx["AP"].astype(np.float64)
    lr = float(fx[tag + "_lr"])
    a, it = oe.project(S, AP, 10, lr)
    np.testing.assert_allclose(a, fx[tag + "_train"], atol=5e-5)
    a, it = oe.project(S, AP, 50, lr)
    np.testing.assert_allclose(a, fx[tag + "_eval"], atol=2e-4)
    np.testing.assert_array_equal(it, fx[tag + "_eval_iters"])
    if tag == "large":
        start = oe.complete_partial(S, AP)
        assert np.abs(fx["large_eval"] - start).max() > 1e-2


def test_recorded_newton_divergence_is_a_property_of_the_input():
    """One step in 46 080 of the EVOPF-RPOSAC parity runs (round 4, seed 40, step 617) stored an equality violation of 117: the
    basic action had the generator voltages at the lower edge of their box with almost no generation.  The float64 oracle
    (numpy LAPACK solves, partial pivoting) does not converge on that input either -- PFFunction's Newton iteration
    (evopf.py:819-835) simply has no solution to find from the flat start -- so the event is not an artefact of the kernels'
    float32 static-order elimination (tests/diag_evopf_sac.py: static and pivoted kernels both return garbage there)."""
    fx = golden("evopf_newton_divergence")
    s = fx["s"][None].astype(np.float64)
    ap = fx["a"][None].astype(np.float64)[:, G.partial_actions]
    a, _, _, its = oe.complete_partial(s, ap, return_aux=True)
    assert int(np.asarray(its).max()) == 50                       # max_iters: no convergence
    assert np.abs(oe.eq_resid(s, a)).max() > 0.1
