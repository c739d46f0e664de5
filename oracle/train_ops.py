"""Oracle for the update-step arithmetic of RPODDPG.train / RPOSAC.train.  Test infrastructure only.

numpy float32 restatements of: the TD target + Huber loss (rpo/algo/rpo_ddpg.py:327-337, rpo_sac.py:342-353), the
Lagrangian term of the actor loss (rpo_ddpg.py:312-322 with Dual.forward, model/dual.py:63-65), clip_grad_norm_(inf)
(rpo_ddpg.py:180,193), torch.optim.Adam / DualAdam (model/dual.py:27-45) and the Polyak update (agent/ddpg_pa.py:77-86).
Pinned against the reference by tests/test_oracle_golden.py (fixtures train_step_*.npz).
"""
import numpy as np

from . import cartsafe, pendulum

F32 = np.float32


def huber(d):
    ad = np.abs(d)
    return np.where(ad < 1, F32(0.5) * d * d, ad - F32(0.5)).astype(F32)


def td_huber(q1, qn1, reward, done, gamma, q2=None, qn2=None, logp=None, alpha=0.0):
    """-> (loss, target, grad_q1, grad_q2).  smooth_l1_loss(beta=1, mean) per critic, summed over critics."""
    q1, qn1 = np.asarray(q1, F32).reshape(-1), np.asarray(qn1, F32).reshape(-1)
    qn = qn1 if qn2 is None else np.minimum(qn1, np.asarray(qn2, F32).reshape(-1))
    if logp is not None:
        qn = qn - F32(alpha) * np.asarray(logp, F32).reshape(-1)
    y = (np.asarray(reward, F32).reshape(-1) + F32(gamma) * (F32(1) - np.asarray(done, F32).reshape(-1)) * qn).astype(F32)
    n = q1.shape[0]
    d1 = q1 - y
    loss = huber(d1).astype(np.float64).sum() / n
    g1 = (np.clip(d1, -1, 1) / F32(n)).astype(F32)
    g2 = None
    if q2 is not None:
        d2 = np.asarray(q2, F32).reshape(-1) - y
        loss += huber(d2).astype(np.float64).sum() / n
        g2 = (np.clip(d2, -1, 1) / F32(n)).astype(F32)
    return F32(loss), y, g1, g2


def lagrangian_cart(action, nu, consts, scale):
    """-> (loss, grad_action, grad_nu) of scale * sum_b nu . relu(g(a_b))."""
    dist = cartsafe.ineq_dist(action, consts)                       # [n,6]
    nu = np.asarray(nu, F32).reshape(-1)
    loss = F32(scale) * (dist.astype(np.float64) @ nu.astype(np.float64)).sum()
    active = (cartsafe.ineq_resid(action, consts) > 0).astype(F32)
    grad_a = (F32(scale) * ((active * nu) @ consts.G)).astype(F32)
    grad_nu = (F32(scale) * dist.astype(np.float64).sum(axis=0)).astype(F32)
    return F32(loss), grad_a, grad_nu


def lagrangian_pendulum(action, nu, scale):
    action = np.asarray(action, F32)
    g = pendulum.ineq_resid(action).reshape(-1)
    dist = np.maximum(g, 0)
    nu0 = F32(np.asarray(nu).reshape(-1)[0])
    loss = F32(scale) * nu0 * dist.astype(np.float64).sum()
    grad_a = (F32(scale) * nu0 * (g > 0).astype(F32)[:, None] * F32(2) * action).astype(F32)
    grad_nu = np.array([F32(scale) * dist.astype(np.float64).sum()], dtype=F32)
    return F32(loss), grad_a, grad_nu


def clip_inf_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm, "inf") on a list of arrays (in place); returns the norm."""
    total = max(float(np.abs(g).max()) for g in grads) if grads else 0.0
    coef = min(F32(max_norm) / (F32(total) + F32(1e-6)), F32(1.0))
    for g in grads:
        g *= F32(coef)
    return total


def adam_step(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, maximize=False,
              clamp_min0=False):
    """One torch.optim.Adam step (single-tensor path) in float32, in place; ``step`` is the count BEFORE this step."""
    step = step + 1
    g = -grad if maximize else grad
    if weight_decay != 0:
        g = g + F32(weight_decay) * param
    m += (g - m) * F32(1 - beta1)
    v *= F32(beta2)
    v += F32(1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    step_size = F32(lr / bc1)
    denom = np.sqrt(v) / F32(bc2 ** 0.5) + F32(eps)
    param -= step_size * (m / denom)
    if clamp_min0:
        np.maximum(param, 0, out=param)
    return step


def polyak(param, target, tau):
    target *= F32(1.0 - tau)
    target += param * F32(tau)
