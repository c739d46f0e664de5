"""Oracle for SpringPendulum-v0 (reference: ``rpo/env/classic_control/pendulum.py``).  Test infrastructure only.

The internal state is ``(theta, theta_dot, l, l_dot)``; the observation is
``(cos theta, sin theta, theta_dot, l, l_dot)`` (pendulum.py:138-140).  All constraint functions are pure
functions of (obs, action): the reference caches per-call tensors on the env object (pendulum.py:284-296,
SURVEY H3), which is never replicated here.
"""
import numpy as np

F32 = np.float32

# pendulum.py:15-29
MAX_SPEED = 8.0
MAX_TORQUE = 6.0
MAX_SUMMATION = 32.0
DT = 0.05
G = 10.0
M = 0.5
K = 1.0
L0 = 1.0
M_DT = M / DT
ACTION_LOW = np.array([-MAX_TORQUE, -MAX_TORQUE], dtype=F32)
ACTION_HIGH = np.array([MAX_TORQUE, MAX_TORQUE], dtype=F32)
RESET_LOW = np.array([-np.pi / 12, -1.0, 0.95, -0.05], dtype=F32)   # pendulum.py:131-132
RESET_HIGH = np.array([np.pi / 12, 1.0, 1.05, 0.05], dtype=F32)
STATE_DIM, INTERNAL_DIM, ACTION_DIM, EQ_NUM, INEQ_NUM = 5, 4, 2, 1, 1  # pendulum.py:39-42
PARTIAL, OTHER = 0, 1                                                # pendulum.py:48-49 (fx is the basic action)
MAX_EPISODE_STEPS = 200                                              # classic_control/__init__.py:15


def get_obs(internal):
    th, thdot, l, ldot = [internal[:, i] for i in range(4)]
    return np.stack([np.cos(th), np.sin(th), thdot, l, ldot], axis=1)


def angle_normalize(x):
    return ((x + np.pi) % (2 * np.pi)) - np.pi                       # pendulum.py:367-368


# ------------------------------------------------------------------ dynamics (float64, pendulum.py:80-128)

def step(internal, action):
    """internal [N,4] f64, action [N,2] f32 (un-clipped) ->
    next_internal [N,4] f64, next_obs [N,5] f64, reward [N] f64, terminated [N] bool, ineq_viol [N,1] f32,
    eq_viol [N,1] f32 (violations of the PRE-step observation and UN-clipped action, pendulum.py:128)."""
    internal = np.asarray(internal, dtype=np.float64)
    action = np.asarray(action, dtype=F32)
    obs = get_obs(internal)
    a_fixed = np.clip(action, ACTION_LOW, ACTION_HIGH).astype(np.float64)   # :85-88
    th, thdot, l, ldot = [internal[:, i] for i in range(4)]
    fx, fy = a_fixed[:, 0], a_fixed[:, 1]
    fth = -fy * np.sin(th) + fx * np.cos(th)                         # :95
    fl = fy * np.cos(th) + fx * np.sin(th)                           # :96
    costs = np.abs(angle_normalize(th))                              # :106 (pre-step theta)
    thacc = (fth - M * (G * np.sin(th) + 2 * ldot * thdot)) / (l * M)   # :109
    lacc = (fl - M * G * np.cos(th) + M * l * thdot ** 2 - K * (l - L0)) / M  # :110
    newthdot = thdot + thacc * DT
    newldot = ldot + lacc * DT
    newth = th + newthdot * DT                                       # semi-implicit in theta :119
    newl = l + ldot * DT                                             # explicit in l         :120
    newthdot = np.clip(newthdot, -MAX_SPEED, MAX_SPEED)              # :122
    terminated = (newl <= 0.5) | (newl >= 1.5) | (newth >= np.pi / 12) | (newth <= -np.pi / 12)  # :124
    nxt = np.stack([newth, newthdot, newl, newldot], axis=1)
    reward = 1 / (100 * costs + 1)
    return nxt, get_obs(nxt), reward, terminated, ineq_dist(action), eq_resid(obs, action)


def reset(rng, n):
    return rng.uniform(low=RESET_LOW, high=RESET_HIGH, size=(n, 4))  # pendulum.py:130-136


# ------------------------------------------------------------------ constraint API (pendulum.py:256-343)

def set_eq(obs):
    """pendulum.py:264-288.  Arithmetic runs in the dtype of ``obs`` (float64 when called from ``env.step``'s info
    path, float32 from the trainer) and is then cast to float32, exactly as the reference's ``torch.tensor(...)``
    casts do.  Returns (C [N,2], C_p [N,1], C_o_inv [N,1], b [N,1]), all float32."""
    obs = np.asarray(obs)
    cos_t, sin_t, thdot, l, ldot = [obs[:, i] for i in range(5)]
    C_p = sin_t.astype(F32).reshape(-1, 1)                           # fx is partial -> sin(theta)
    C_o = cos_t.astype(F32).reshape(-1, 1)
    C = np.concatenate([C_p, C_o], axis=1)
    C_o_inv = (F32(1.0) / C_o).astype(F32)
    b = -M_DT * ldot - (l * M * thdot ** 2 - K * (l - L0) - M * G * cos_t)   # :287
    return C, C_p, C_o_inv, b.astype(F32).reshape(-1, 1)


def complete_partial(obs, action_partial):
    """pendulum.py:256-262."""
    _, C_p, C_o_inv, b = set_eq(obs)
    ap = np.asarray(action_partial, dtype=F32).reshape(-1, 1)
    return np.concatenate([ap, ((b - ap * C_p) * C_o_inv).astype(F32)], axis=1)


def eq_resid(obs, action):
    """pendulum.py:298-300 -> [N,1]."""
    C, _, _, b = set_eq(obs)
    action = np.asarray(action, dtype=F32)
    return (b - (action[:, [0]] * C[:, [0]] + action[:, [1]] * C[:, [1]])).astype(F32)


def ineq_resid(action):
    """pendulum.py:302-303: ||a||^2 - 32 -> [N,1]."""
    action = np.asarray(action, dtype=F32)
    return (action[:, [0]] * action[:, [0]] + action[:, [1]] * action[:, [1]] - F32(MAX_SUMMATION)).astype(F32)


def ineq_dist(action):
    return np.maximum(ineq_resid(action), F32(0))                    # pendulum.py:309-311


def ineq_partial_grad(obs, action, batched_reference=False):
    """pendulum.py:331-343.

    ``batched_reference=False``: row-wise semantics (what the reference computes for B=1, i.e. in every rollout).
    ``batched_reference=True``: the literal batched arithmetic of :337-339, in which ``[B,1] @ [1,B]`` couples
    every sample with every other -- kept only so the B=256 fixtures of ``critic_loss`` can be matched.
    """
    action = np.asarray(action, dtype=F32)
    _, C_p, C_o_inv, b = set_eq(obs)
    Gm = (F32(2) * action).astype(F32)                               # set_ineq :296
    dgp = (Gm[:, [0]] - Gm[:, [1]] * (C_o_inv * C_p)).astype(F32)    # :334-335
    bgp = (F32(MAX_SUMMATION) - (b * C_o_inv) * Gm[:, [1]]).astype(F32)   # :336
    if batched_reference:
        bm = np.maximum(action[:, [0]] @ dgp.T - bgp, F32(0))        # [B,B]
        grad = ((bm > 0).astype(F32) @ dgp).astype(F32)
    else:
        bm = np.maximum(action[:, [0]] * dgp - bgp, F32(0))
        grad = ((bm > 0).astype(F32) * dgp).astype(F32)
    return np.concatenate([grad, (-(grad * C_p) * C_o_inv).astype(F32)], axis=1)


def grad_steps(obs, action, lr, max_steps, corr_eps=1e-5, momentum=0.0, batch_global_stop=False,
               batched_reference=False):
    """GRG projection loop of rpo/algo/rpo_ddpg.py:266-305 for this env; see oracle.cartsafe.grad_steps."""
    a = np.array(action, dtype=F32, copy=True)
    n = a.shape[0]
    iters = np.zeros(n, dtype=np.int32)
    old = np.zeros_like(a)
    lr, momentum, corr_eps = F32(lr), F32(momentum), F32(corr_eps)
    for k in range(int(max_steps)):
        viol = np.maximum(np.abs(eq_resid(obs, a)).max(axis=1), ineq_dist(a).max(axis=1)) > corr_eps
        if batch_global_stop:
            active = np.full(n, bool(k == 0 or viol.any()))
        else:
            active = viol | (k == 0)
        if not active.any():
            break
        stp = (lr * ineq_partial_grad(obs, a, batched_reference) + momentum * old).astype(F32)
        a = np.where(active[:, None], a - stp, a).astype(F32)
        old = np.where(active[:, None], stp, old)
        iters += active
    return a, iters
