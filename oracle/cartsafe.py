"""Oracle for CartSafe-v0 (reference: ``rpo/env/classic_control/cartpole.py``).  Test infrastructure only.

Vectorised over a leading batch axis; row ``i`` of every output is exactly what the reference computes for
env ``i`` alone (the reference's batch behaviour for this env has no cross-sample terms).
"""
import math

import numpy as np

F32 = np.float32

# cartpole.py:75-91
GRAVITY = 9.8
MASSCART = 1.0
MASSPOLE = 0.1
TOTAL_MASS = MASSPOLE + MASSCART
LENGTH = 0.5
POLEMASS_LENGTH = MASSPOLE * LENGTH
TAU = 0.02
MU_C = 0.1
MU_P = 0.01
DELTA = np.array([np.pi / 3, -np.pi / 6])
THETA_THRESHOLD = 12 * 2 * math.pi / 360
X_THRESHOLD = 2.4
ACTION_LOW = np.array([-10.0, -10.0], dtype=F32)   # cartpole.py:107
ACTION_HIGH = np.array([10.0, 10.0], dtype=F32)
RESET_LOW, RESET_HIGH = -0.05, 0.05                 # cartpole.py:233
STATE_DIM, ACTION_DIM, EQ_NUM, INEQ_NUM = 6, 2, 1, 6  # cartpole.py:109-112
MAX_EPISODE_STEPS = 200                             # rpo/env/classic_control/__init__.py:9
# torch.sin / torch.cos of float32(delta) exactly as the reference's CPU run produces them (cartpole.py:124,129;
# torch's vectorised sinf is 1 ulp off the correctly-rounded value for sin(pi/3)); pinned bit-for-bit by
# tests/test_oracle_golden.py::test_cart_constants.
SIN_DELTA32 = np.array([0x3F5DB3D7, 0xBF000000], dtype=np.uint32).view(F32)   # ( 0.8660254, -0.5)
COS_DELTA32 = np.array([0x3EFFFFFF, 0x3F5DB3D7], dtype=np.uint32).view(F32)   # ( 0.49999997, 0.8660254)


class Constants(object):
    """Constraint matrices of cartpole.py:124-136 for a given choice of ``partial_actions``.

    The reference draws ``partial_actions`` from the global numpy RNG (cartpole.py:117); with
    ``np.random.seed(123)`` (scripts/cart_exp.py:9) the first env gets ``[1]``.
    """

    def __init__(self, partial=1):
        self.partial = int(partial)
        self.other = 1 - self.partial
        self.C = SIN_DELTA32.reshape(1, 2).copy()                   # diff_eq           :124
        self.C_p = self.C[:, [self.partial]]                        # diff_eq_partial   :125
        self.C_o_inv = (F32(1.0) / self.C[:, [self.other]]).astype(F32)  # 1x1 inverse  :126
        self.b = np.zeros((1, 1), dtype=F32)                        # diff_eq_bias      :127
        cosd = COS_DELTA32
        G = np.zeros((6, 2), dtype=F32)                             # diff_ineq         :129-135
        G[0] = cosd
        G[1] = -cosd
        G[2, 0], G[3, 0], G[4, 1], G[5, 1] = 1.0, -1.0, 1.0, -1.0
        self.G = G
        self.d = np.array([8, 8, 10, 10, 10, 10], dtype=F32)        # diff_ineq_bias    :136
        # reduced (partial-space) inequality of ineq_partial_grad, cartpole.py:397-400
        self.G_r = (G[:, [self.partial]] - G[:, [self.other]] @ (self.C_o_inv @ self.C_p)).astype(F32)
        self.d_r = (self.d - (self.b @ self.C_o_inv.T) @ G[:, [self.other]].T).reshape(6).astype(F32)

    def as_array(self):
        """Flat float32 table handed to the HIP kernels (layout documented in include/rpo_hip.h)."""
        return np.concatenate([self.C.reshape(2), self.C_p.reshape(1), self.C_o_inv.reshape(1), self.b.reshape(1),
                               self.G.reshape(12), self.d, self.G_r.reshape(6), self.d_r]).astype(F32)


# ------------------------------------------------------------------ dynamics (float64, cartpole.py:166-229)

def step(state, action, consts):
    """One env step for every row.

    state  [N,6] float64 = (x, x_dot, xacc, theta, theta_dot, thetaacc); action [N,2] float32, un-clipped.
    Returns next_state [N,6] f64, reward [N] f64, terminated [N] bool, ineq_viol [N,6] f32, eq_viol [N,1] f32.
    ``terminated`` is the env's own ``done`` (cartpole.py:208-213); the 200-step TimeLimit is applied by the caller.
    The violations are those of the PRE-step state and the UN-clipped action (cartpole.py:229).
    """
    state = np.asarray(state, dtype=np.float64)
    action = np.asarray(action, dtype=F32)
    a_fixed = np.clip(action, ACTION_LOW, ACTION_HIGH)              # :170-173 (float32 clip)
    x, x_dot, _xacc, theta, theta_dot, thetaacc = [state[:, i] for i in range(6)]
    force_x = a_fixed.astype(np.float64) @ np.cos(DELTA)             # :178
    force_y = a_fixed.astype(np.float64) @ np.sin(DELTA)             # :179
    force = force_x
    costheta, sintheta = np.cos(theta), np.sin(theta)
    # normal force uses the PREVIOUS step's thetaacc (:183)
    n_c = force_y + TOTAL_MASS * GRAVITY - POLEMASS_LENGTH * (thetaacc * sintheta + theta_dot * theta_dot * costheta)
    sign = np.sign(n_c * x_dot)                                      # :184
    temp = (force + POLEMASS_LENGTH * theta_dot * theta_dot * (sintheta + MU_C * sign * costheta)) / TOTAL_MASS \
        + MU_C * GRAVITY * sign                                      # :185-186
    thetaacc_new = (GRAVITY * sintheta - costheta * temp - MU_P * theta_dot / POLEMASS_LENGTH) / \
        (LENGTH * (4.0 / 3.0 - MASSPOLE * costheta * (costheta - MU_C * GRAVITY * sign) / TOTAL_MASS))  # :187-189
    xacc_new = (force + POLEMASS_LENGTH * (theta_dot * theta_dot * sintheta - thetaacc_new * costheta)
                - MU_C * n_c * sign) / TOTAL_MASS                   # :190-191
    x_new = x + TAU * x_dot                                          # explicit Euler :193-197
    x_dot_new = x_dot + TAU * xacc_new
    theta_new = theta + TAU * theta_dot
    theta_dot_new = theta_dot + TAU * thetaacc_new
    nxt = np.stack([x_new, x_dot_new, xacc_new, theta_new, theta_dot_new, thetaacc_new], axis=1)
    terminated = (x_new < -X_THRESHOLD) | (x_new > X_THRESHOLD) | \
        (theta_new < -THETA_THRESHOLD) | (theta_new > THETA_THRESHOLD)   # :208-213
    reward = np.ones(state.shape[0])                                 # :215-221 (reset follows every done)
    return nxt, reward, terminated, ineq_dist(action, consts), eq_resid(action, consts)


def reset(rng, n):
    """cartpole.py:231-239: U(-0.05, 0.05)^6 from the env's own RandomState."""
    return rng.uniform(low=RESET_LOW, high=RESET_HIGH, size=(n, 6))


# ------------------------------------------------------------------ constraint API (float32, cartpole.py:369-422)

def complete_partial(action_partial, consts):
    """cartpole.py:369-373.  action_partial [N,1] f32 -> action [N,2] f32."""
    ap = np.asarray(action_partial, dtype=F32).reshape(-1, 1)
    out = np.zeros((ap.shape[0], 2), dtype=F32)
    out[:, [consts.partial]] = ap
    out[:, [consts.other]] = (consts.b.T - ap @ consts.C_p.T) @ consts.C_o_inv.T
    return out


def eq_resid(action, consts):
    """cartpole.py:375-376 -> [N,1]."""
    return (consts.b - np.asarray(action, dtype=F32) @ consts.C.T).astype(F32)


def ineq_resid(action, consts):
    """cartpole.py:378-379 -> [N,6]."""
    return (np.asarray(action, dtype=F32) @ consts.G.T - consts.d).astype(F32)


def ineq_dist(action, consts):
    """cartpole.py:385-387."""
    return np.maximum(ineq_resid(action, consts), F32(0))


def ineq_partial_grad(action, consts):
    """cartpole.py:396-408: sign-based reduced (sub)gradient; ``eps`` is ignored by the reference (:402)."""
    action = np.asarray(action, dtype=F32)
    bias_mod = np.maximum(action[:, [consts.partial]] @ consts.G_r.T - consts.d_r, F32(0))
    grad = (bias_mod > 0).astype(F32) @ consts.G_r
    out = np.zeros_like(action)
    out[:, [consts.partial]] = grad
    out[:, [consts.other]] = -(grad @ consts.C_p.T) @ consts.C_o_inv.T
    return out


def grad_steps(action, consts, lr, max_steps, corr_eps=1e-5, momentum=0.0, batch_global_stop=False):
    """GRG projection loop, rpo/algo/rpo_ddpg.py:266-305 (``corr_mode == 0``).

    Returns (action [N,2] f32, iterations [N] int32).  ``batch_global_stop=True`` reproduces the reference's
    stop test literally (one ``torch.max`` over the whole batch, rpo_ddpg.py:271-272); ``False`` applies the same
    test per row, which is what the reference does in its B=1 rollouts and what the HIP kernels do.
    """
    a = np.array(action, dtype=F32, copy=True)
    n = a.shape[0]
    iters = np.zeros(n, dtype=np.int32)
    old = np.zeros_like(a)
    lr, momentum, corr_eps = F32(lr), F32(momentum), F32(corr_eps)
    for k in range(int(max_steps)):
        viol = np.maximum(np.abs(eq_resid(a, consts)).max(axis=1), ineq_dist(a, consts).max(axis=1)) > corr_eps
        if batch_global_stop:
            active = np.full(n, bool(k == 0 or viol.any()))
        else:
            active = viol | (k == 0)
        if not active.any():
            break
        stp = (lr * ineq_partial_grad(a, consts) + momentum * old).astype(F32)
        a = np.where(active[:, None], a - stp, a).astype(F32)
        old = np.where(active[:, None], stp, old)
        iters += active
    return a, iters
