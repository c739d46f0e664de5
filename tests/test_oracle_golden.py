"""oracle/ == the reference, on the golden vectors generated from the unmodified reference (CPU only)."""
import numpy as np
import pytest

from oracle import cartsafe as cs
from oracle import pendulum as pd

# float32 functions: the reference evaluates them with torch matmul/bmm, the oracle with numpy; the only
# differences are fused-multiply-add / summation-order roundings of 2-term dot products.
F32_TOL = dict(rtol=2e-6, atol=2e-6)
# float64 dynamics: same formulas, same operation order.
F64_TOL = dict(rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_constants(golden, partial):
    g = golden("cart_env_p%d" % partial)
    c = cs.Constants(partial)
    assert int(g["partial"]) == partial
    np.testing.assert_array_equal(c.C, g["C"])
    np.testing.assert_array_equal(c.C_p, g["C_p"])
    np.testing.assert_allclose(c.C_o_inv, g["C_o_inv"], rtol=1e-7)
    np.testing.assert_array_equal(c.b, g["b"])
    np.testing.assert_array_equal(c.G, g["G"])
    np.testing.assert_array_equal(c.d, g["d"])


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_step(golden, partial):
    g = golden("cart_env_p%d" % partial)
    c = cs.Constants(partial)
    nxt, rew, term, ineq, eq = cs.step(g["states"], g["actions"], c)
    np.testing.assert_allclose(nxt, g["next_states"], **F64_TOL)
    np.testing.assert_array_equal(term, g["done"])
    np.testing.assert_array_equal(rew, g["reward"])
    np.testing.assert_allclose(ineq, g["ineq_viol"], **F32_TOL)
    np.testing.assert_allclose(eq, g["eq_viol"], **F32_TOL)
    assert g["done"].sum() > 100 and (~g["done"]).sum() > 100          # both outcomes are covered


def test_cart_known_answer():
    """SURVEY.md §8c anchor observed on the reference."""
    c = cs.Constants(1)
    s = np.array([[0.00488135, 0.02151894, 0.01027634, 0.00448832, -0.00763452, 0.01458941]])
    nxt, rew, term, ineq, eq = cs.step(s, np.array([[1.0, 2.0]], dtype=np.float32), c)
    np.testing.assert_allclose(nxt[0], [5.31172913e-03, 4.67920206e-02, 1.26365420, 4.33562790e-03,
                                        -9.66637048e-02, -4.45145924], rtol=1e-6)
    assert not term[0] and rew[0] == 1.0 and np.all(ineq == 0)
    np.testing.assert_allclose(eq[0, 0], 0.13397461, rtol=1e-6)
    a = cs.complete_partial(np.array([[9.9]], dtype=np.float32), c)
    np.testing.assert_allclose(a[0], [5.71576738, 9.9], rtol=1e-6)
    tr, _ = cs.grad_steps(a, c, lr=2e-2, max_steps=10)
    np.testing.assert_allclose(tr[0], [5.58243418, 9.66905785], rtol=1e-6)
    ev, it = cs.grad_steps(a, c, lr=2e-2, max_steps=50)
    np.testing.assert_allclose(ev[0], [5.04910135, 8.74529076], rtol=1e-6)
    assert it[0] == 50


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_constraint_api(golden, partial):
    g = golden("cart_env_p%d" % partial)
    c = cs.Constants(partial)
    full = cs.complete_partial(g["ap"], c)
    np.testing.assert_allclose(full, g["completed"], **F32_TOL)
    np.testing.assert_allclose(cs.eq_resid(g["completed"], c), g["eq_resid_completed"], **F32_TOL)
    np.testing.assert_allclose(cs.ineq_resid(g["completed"], c), g["ineq_resid_completed"], **F32_TOL)
    np.testing.assert_allclose(cs.ineq_dist(g["completed"], c), g["ineq_dist_completed"], **F32_TOL)
    np.testing.assert_allclose(cs.ineq_partial_grad(g["completed"], c), g["ipg_completed"], **F32_TOL)
    np.testing.assert_allclose(cs.ineq_partial_grad(g["completed"][:64], c), g["ipg_completed_b1"], **F32_TOL)
    np.testing.assert_allclose(cs.eq_resid(g["any_actions"], c), g["eq_resid_any"], **F32_TOL)
    np.testing.assert_allclose(cs.ineq_resid(g["any_actions"], c), g["ineq_resid_any"], rtol=2e-6, atol=4e-6)
    np.testing.assert_allclose(cs.ineq_partial_grad(g["any_actions"], c), g["ipg_any"], **F32_TOL)
    assert np.abs(g["ipg_completed"]).max() > 0                      # some rows do violate


@pytest.mark.parametrize("partial", [1, 0])
def test_cart_grad_steps(golden, partial):
    g = golden("cart_grad_steps_p%d" % partial)
    c = cs.Constants(partial)
    a0 = g["completed"]
    tr, _ = cs.grad_steps(a0, c, lr=2e-2, max_steps=10)
    np.testing.assert_allclose(tr, g["train_b1"], rtol=1e-5, atol=1e-5)
    ev, it = cs.grad_steps(a0, c, lr=2e-2, max_steps=50)
    np.testing.assert_allclose(ev, g["eval_b1"], rtol=1e-5, atol=2e-5)
    np.testing.assert_array_equal(it, g["eval_b1_iters"])
    # the reference's batched call (global stop test) -- for this env identical to per-row, feasible rows get a
    # zero step (cartpole.py:403), except for the iteration count which is the batch maximum
    trb, _ = cs.grad_steps(a0, c, lr=2e-2, max_steps=10, batch_global_stop=True)
    np.testing.assert_allclose(trb, g["train_batched"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(tr, g["train_batched"], rtol=1e-5, atol=1e-5)
    evb, itb = cs.grad_steps(a0, c, lr=2e-2, max_steps=50, batch_global_stop=True)
    np.testing.assert_allclose(evb, g["eval_batched"], rtol=1e-5, atol=2e-5)
    assert itb.max() == int(g["eval_batched_iters"])
    trm, _ = cs.grad_steps(a0, c, lr=2e-2, max_steps=10, momentum=0.5)
    np.testing.assert_allclose(trm, g["train_b1_mom"], rtol=1e-5, atol=1e-5)
    assert (it < 50).any() and (it == 50).any()


def test_pendulum_step(golden):
    g = golden("pendulum_env")
    nxt, obs, rew, term, ineq, eq = pd.step(g["internal"], g["actions"])
    np.testing.assert_allclose(nxt, g["next_internal"], **F64_TOL)
    np.testing.assert_allclose(obs, g["next_obs"], **F64_TOL)
    np.testing.assert_allclose(rew, g["reward"], **F64_TOL)
    np.testing.assert_array_equal(term, g["done"])
    np.testing.assert_allclose(ineq, g["ineq_viol"], rtol=2e-6, atol=1e-5)
    np.testing.assert_allclose(eq, g["eq_viol"], rtol=2e-6, atol=1e-5)
    assert g["done"].sum() > 100 and (~g["done"]).sum() > 100


def test_pendulum_known_answer():
    internal = np.array([[0.1, -0.5, 1.02, 0.03]])
    nxt, obs, rew, term, ineq, eq = pd.step(internal, np.array([[1.5, 4.0]], dtype=np.float32))
    np.testing.assert_allclose(obs[0], [0.99696069, 0.0779063, -0.44029358, 1.0215, -0.0437754], rtol=1e-6)
    np.testing.assert_allclose(rew[0], 0.0909090909, rtol=1e-9)
    np.testing.assert_allclose(eq[0, 0], 0.43775368, rtol=1e-5)
    a = pd.complete_partial(pd.get_obs(internal).astype(np.float32), np.array([[5.9]], dtype=np.float32))
    np.testing.assert_allclose(a[0], [5.9, 3.99847889], rtol=1e-6)
    np.testing.assert_allclose(pd.ineq_resid(a)[0, 0], 18.7978363, rtol=1e-6)


def test_pendulum_constraint_api(golden):
    g = golden("pendulum_env")
    obs = g["obs32"]
    np.testing.assert_allclose(pd.complete_partial(obs, g["ap"]), g["completed"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pd.eq_resid(obs, g["completed"]), g["eq_resid_completed"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pd.ineq_resid(g["completed"]), g["ineq_resid_completed"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pd.eq_resid(obs, g["any_actions"]), g["eq_resid_any"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pd.ineq_resid(g["any_actions"]), g["ineq_resid_any"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(pd.ineq_partial_grad(obs, g["any_actions"]), g["ipg_any_b1"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pd.ineq_partial_grad(obs, g["completed"]), g["ipg_completed_b1"], rtol=1e-5, atol=1e-3)
    # the reference's batched call couples samples (pendulum.py:337-339): reproduced only in compat mode
    np.testing.assert_allclose(pd.ineq_partial_grad(obs[:8], g["any_actions"][:8], batched_reference=True),
                               g["ipg_batched_small"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(pd.ineq_partial_grad(obs, g["any_actions"], batched_reference=True),
                               g["ipg_any_batched"], rtol=1e-4, atol=1e-2)
    assert not np.allclose(g["ipg_any_batched"], g["ipg_any_b1"])     # the hazard is real


def test_pendulum_grad_steps(golden):
    g = golden("pendulum_grad_steps")
    obs, a0 = g["obs32"], g["completed"]
    tr, _ = pd.grad_steps(obs, a0, lr=2e-3, max_steps=10)
    np.testing.assert_allclose(tr, g["train_b1"], rtol=1e-4, atol=1e-4)
    ev, it = pd.grad_steps(obs, a0, lr=2e-3, max_steps=50)
    np.testing.assert_allclose(ev, g["eval_b1"], rtol=1e-4, atol=2e-4)
    np.testing.assert_array_equal(it, g["eval_b1_iters"])
    ev2, it2 = pd.grad_steps(obs, a0, lr=2e-2, max_steps=50)
    same = it2 == g["eval_b1_lr2e2_iters"]
    assert same.mean() > 0.97        # stop test sits on a 1e-5 threshold; a 1-ulp difference can add one step
    np.testing.assert_allclose(ev2[same], g["eval_b1_lr2e2"][same], rtol=1e-3, atol=1e-3)
    trb, _ = pd.grad_steps(obs, a0, lr=2e-3, max_steps=10, batch_global_stop=True, batched_reference=True)
    np.testing.assert_allclose(trb, g["train_batched"], rtol=1e-3, atol=1e-2)
