"""A stand-in for ``rpo_amd.ops`` built on ``oracle/`` -- TEST INFRASTRUCTURE, lives under tests/ on purpose.

It implements the same kernel-set interface on CPU tensors with the oracle's numpy arithmetic, so that (a) the host
logic of the trainers (loop cadence, flat parameter layout, optimiser order, logging, sharding, gradient buckets) is
exercised by the CPU suite, and (b) a whole trainer driven by the oracle can be compared with the same trainer driven
by the HIP kernels.  The product never imports this module; ``rpo_amd`` has no CPU fallback.
"""
import numpy as np
import torch

from oracle import cartsafe as cs
from oracle import evopf as oe
from oracle import pendulum as pd
from oracle import philox, train_ops
from rpo_amd._lib import CONST

NOISE_NONE, NOISE_EXPLICIT, NOISE_PHILOX, NOISE_UNIFORM, NOISE_CLIP_ONLY = 0, 1, 2, 3, 4
STREAM_POLICY = CONST["RPO_STREAM_POLICY"]
STATS_LEN = CONST["RPO_STATS_LEN"]
CTRL_LEN = CONST["RPO_CTRL_LEN"]
STATS_SUB = CONST["RPO_STATS_SUB"]
STAT = {k[len("RPO_STAT_"):].lower(): v for k, v in CONST.items()
        if k.startswith("RPO_STAT_") and k not in ("RPO_STATS_LEN", "RPO_STATS_SUB")}


def _np(t):
    return t.detach().numpy()


def _put(t, arr):
    t.copy_(torch.as_tensor(np.asarray(arr), dtype=t.dtype).reshape(t.shape))


class _EnvKernels(object):
    """Vector-step bookkeeping shared by both envs (TimeLimit, auto-reset, ring scatter, statistics)."""

    def _t(self, ctrl):
        return 0 if ctrl is None else int(ctrl[0])

    def _explore(self, ap_raw, noise, n, mode, eps_start, eps_end, eps_decay, lo, hi, seed, base, t):
        eps_t = np.float32(max(eps_end, eps_start - eps_decay * t))
        ids = np.arange(n) + base
        if mode == NOISE_UNIFORM:
            r = philox.draw(seed, ids, t, philox.STREAM_ACT)
            scale = np.float32((hi - lo) * 0.5)
            return (scale * (np.float32(2) * philox.u01(r[:, 0]) - np.float32(1)) + np.float32(lo + scale)).astype(np.float32)
        ap = _np(ap_raw).reshape(-1).astype(np.float32)
        if mode == NOISE_EXPLICIT:
            ap = np.clip(ap + eps_t * _np(noise).reshape(-1), lo, hi)
        elif mode == NOISE_PHILOX:
            r = philox.draw(seed, ids, t, philox.STREAM_ACT)
            ap = np.clip(ap + eps_t * philox.normal(r[:, 0], r[:, 1]), lo, hi)
        elif mode == NOISE_CLIP_ONLY:
            ap = np.clip(ap, lo, hi)
        return ap.astype(np.float32)

    def _finish_step(self, n, pre_obs, action, nxt_obs, reward, terminated, eq, ineq, ep_len, ep_ret, ep_count, rows,
                     cap_steps, stats, ctrl, max_episode_steps, auto_reset, viol_thresh, t):
        length = _np(ep_len) + 1
        done = terminated | (length >= max_episode_steps)
        ret = _np(ep_ret) + reward.astype(np.float32)
        if rows is not None:
            row = np.zeros((n, self.row_floats), dtype=np.float32)
            c = self.cols
            row[:, c["state"][0]:c["state"][1]] = pre_obs
            row[:, c["action"][0]:c["action"][1]] = action
            row[:, c["next_state"][0]:c["next_state"][1]] = nxt_obs
            row[:, c["reward"][0]] = reward
            row[:, c["done"][0]] = done
            row[:, c["eq_viol"][0]:c["eq_viol"][1]] = eq
            row[:, c["ineq_viol"][0]:c["ineq_viol"][1]] = ineq
            base = (t % cap_steps) * n
            rows[base:base + n, :self.row_floats] = torch.as_tensor(row)
        if stats is not None:
            cap = stats.shape[0]
            r = stats[t % cap, 0]                       # everything into sub-row 0 (the reader sums the sub-rows)
            mi, me = ineq.max(axis=1), np.abs(eq).max(axis=1)
            r[STAT["reward_sum"]] += float(reward.sum())
            r[STAT["episodes"]] += float(done.sum())
            r[STAT["return_sum"]] += float(ret[done].sum())
            r[STAT["length_sum"]] += float(length[done].sum())
            r[STAT["max_ineq_sum"]] += float(mi.sum())
            r[STAT["max_eq_sum"]] += float(me.sum())
            r[STAT["viol_count"]] += float((np.maximum(mi, me) > viol_thresh).sum())
            r[STAT["terminated"]] += float((terminated & done).sum())
            r[STAT["max_ineq_max"]] = max(float(r[STAT["max_ineq_max"]]), float(mi.max()))
            r[STAT["max_eq_max"]] = max(float(r[STAT["max_eq_max"]]), float(me.max()))
            if cap > 1 and ctrl is not None:
                stats[(t + 1) % cap].zero_()
        reset_mask = done & bool(auto_reset)
        _put(ep_count, _np(ep_count) + reset_mask)
        _put(ep_len, np.where(reset_mask, 0, length))
        _put(ep_ret, np.where(reset_mask, 0, ret))
        if ctrl is not None:
            # failure detection (include/rpo_hip.h RPO_CTRL_NONFINITE; the reference's assert on a NaN action, cartpole.py:170-174)
            if int(ctrl[CONST["RPO_CTRL_NONFINITE"]]) == 0 and (np.isnan(action).any() or not np.isfinite(nxt_obs).all() or
                                                                not np.isfinite(reward).all()):
                ctrl[CONST["RPO_CTRL_NONFINITE"]] = t + 1
            ctrl[0] = t + 1
        return reset_mask

    def _stat_iters(self, stats, ctrl, iters):
        if stats is not None:
            stats[self._t(ctrl) % stats.shape[0], 0, STAT["proj_iters"]] += float(iters.sum())


class CartSafeKernels(_EnvKernels):
    name = "CartSafe-v0"
    obs_dim, internal_dim, action_dim, eq_num, ineq_num, partial_dim = 6, 6, 2, 1, 6, 1
    row_floats, ring_floats = CONST["RPO_CART_ROW"], CONST["RPO_CART_RING"]
    cols = dict(state=(0, 6), action=(6, 8), next_state=(8, 14), reward=(14, 15), done=(15, 16), eq_viol=(16, 17),
                ineq_viol=(17, 23))

    def __init__(self, consts, partial):
        self.c = cs.Constants(partial)
        assert np.array_equal(np.asarray(consts, dtype=np.float32), self.c.as_array()), "constant table drifted"
        self.partial = int(partial)

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        n = internal.shape[0]
        _put(internal, philox.cart_reset(seed, np.arange(n) + env_id_base, _np(ep_count).astype(np.uint32)))
        ep_len.zero_()
        ep_ret.zero_()

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        n, t = internal.shape[0], self._t(ctrl)
        s, a = _np(internal).astype(np.float64), _np(action)
        nxt, rew, term, ineq, eq = cs.step(s, a, self.c)
        nxt32 = nxt.astype(np.float32)
        mask = self._finish_step(n, s.astype(np.float32), a, nxt32, rew, term, eq, ineq, ep_len, ep_ret, ep_count,
                                 rows, cap_steps, stats, ctrl, max_episode_steps, auto_reset, viol_thresh, t)
        fresh = philox.cart_reset(seed, np.arange(n) + env_id_base, _np(ep_count).astype(np.uint32))
        _put(internal, np.where(mask[:, None], fresh, nxt32))

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None):
        n = action.shape[0]
        ap = self._explore(ap_raw, noise, n, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, seed,
                           env_id_base, self._t(ctrl))
        a, it = cs.grad_steps(cs.complete_partial(ap, self.c), self.c, corr_lr, max_steps, corr_eps, corr_momentum)
        _put(action, a)
        if iters is not None:
            _put(iters, it)
        self._stat_iters(stats, ctrl, it)

    def complete_bwd(self, obs, grad_action, grad_ap, action=None):
        k = -(self.c.C_p * self.c.C_o_inv)[0, 0]
        g = _np(grad_action)
        _put(grad_ap, g[:, self.c.partial] + k * g[:, self.c.other])

    def resid(self, obs, action, eq_out, ineq_out):
        if eq_out is not None:
            _put(eq_out, cs.eq_resid(_np(action), self.c))
        if ineq_out is not None:
            _put(ineq_out, cs.ineq_resid(_np(action), self.c))

    def ineq_partial_grad(self, obs, action, step_out):
        _put(step_out, cs.ineq_partial_grad(_np(action), self.c))

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None):
        loss, ga, gnu = train_ops.lagrangian_cart(_np(action), _np(nu), self.c, scale)
        loss_out += float(loss)
        if grad_action is not None:
            _put(grad_action, ga)
        if grad_nu is not None:
            grad_nu += torch.as_tensor(gnu)


class EvopfKernels(_EnvKernels):
    """oracle/evopf.py behind the kernel-set interface of rpo_amd.ops.EvopfKernels (float64 arithmetic)."""
    name = "EVOPF-v0"
    obs_dim = internal_dim = 57
    action_dim, partial_dim, eq_num, ineq_num = 43, 14, 28, 58
    row_floats = ring_floats = CONST["RPO_EVOPF_ROW"]
    cols = dict(state=(0, 57), action=(57, 100), next_state=(100, 157), reward=(157, 158), done=(158, 159),
                eq_viol=(159, 187), ineq_viol=(187, 245))
    episode_steps = 24
    partial = 0

    def __init__(self, consts):
        self.consts = np.asarray(consts, dtype=np.float32)
        C = CONST
        np.testing.assert_allclose(self.consts[C["RPO_EVOPF_C_YR"]:C["RPO_EVOPF_C_YR"] + 196].reshape(14, 14), oe.GRID.Yr, atol=1e-6)

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        n = internal.shape[0]
        _put(internal, oe.reset(seed, np.arange(n) + env_id_base, _np(ep_count).astype(np.int64)))
        ep_len.zero_()
        ep_ret.zero_()

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        n, t = internal.shape[0], self._t(ctrl)
        s, a = _np(internal).astype(np.float64), _np(action).astype(np.float64)
        ids = np.arange(n) + env_id_base
        out = oe.step(s, a, _np(ep_len).astype(np.int64), _np(ep_count).astype(np.int64), seed, ids, auto_reset=False)
        mask = self._finish_step(n, s.astype(np.float32), _np(action), out["next_state"].astype(np.float32),
                                 out["reward"], out["done"], out["eq_viol"], out["ineq_viol"], ep_len, ep_ret, ep_count,
                                 rows, cap_steps, stats, ctrl, max_episode_steps, auto_reset, viol_thresh, t)
        fresh = oe.reset(seed, ids, _np(ep_count).astype(np.int64))
        _put(internal, np.where(mask[:, None], fresh, out["next_state"]))

    def _explore14(self, obs, ap_raw, noise, n, mode, eps_start, eps_end, eps_decay, seed, base, t):
        lo, hi = oe.partial_box(_np(obs).astype(np.float64))
        eps_t = max(eps_end, eps_start - eps_decay * t)
        ids = np.arange(n) + base
        if mode == NOISE_UNIFORM:
            w = np.concatenate([philox.draw(seed, ids, t, philox.STREAM_ACT, c) for c in range(4)], axis=1)[:, :14]
            scale = (hi - lo) * 0.5
            return scale * (2.0 * philox.u01(w) - 1.0) + (lo + scale)
        ap = _np(ap_raw).reshape(n, 14).astype(np.float64)
        if mode == NOISE_EXPLICIT:
            return np.clip(ap + eps_t * _np(noise).reshape(n, 14), lo, hi)
        if mode == NOISE_PHILOX:
            z = np.zeros((n, 14))
            for c in range(7):
                w = philox.draw(seed, ids, t, philox.STREAM_ACT, c)
                z[:, 2 * c], z[:, 2 * c + 1] = philox.normal(w[:, 0], w[:, 1]), philox.normal(w[:, 2], w[:, 3])
            return np.clip(ap + eps_t * z, lo, hi)
        if mode == NOISE_CLIP_ONLY:
            return np.clip(ap, lo, hi)
        return ap

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None,
                    ap_is_raw=False):
        n = action.shape[0]
        s = _np(obs).astype(np.float64)
        if ap_is_raw and noise_mode != NOISE_UNIFORM:
            lo, hi = oe.partial_box(s)
            scale = (hi - lo) * 0.5
            ap_raw = torch.as_tensor(scale * np.tanh(_np(ap_raw).reshape(n, 14).astype(np.float64)) + (lo + scale))
        ap = self._explore14(obs, ap_raw, noise, n, noise_mode, eps_start, eps_end, eps_decay, seed, env_id_base, self._t(ctrl))
        a, it = oe.project(s, ap, max_steps, corr_lr, corr_eps, corr_momentum)
        _put(action, a)
        if iters is not None:
            _put(iters, it)
        self._stat_iters(stats, ctrl, it)

    def eq_vjp(self, action, grad_eq, grad_action, autograd_sign=True):
        jac = oe.eq_jac(_np(action).astype(np.float64))
        if autograd_sign:
            jac[:, :, oe.GRID.pe0:] *= -1.0
        _put(grad_action, np.einsum("nev,ne->nv", jac, _np(grad_eq).astype(np.float64)))

    def tanh_box_bwd(self, obs, raw, noise, eps_start, eps_end, eps_decay, ctrl, dap, dout):
        n = dout.numel() // 14
        lo, hi = oe.partial_box(_np(obs).astype(np.float64))
        scale = (hi - lo) * 0.5
        th = np.tanh(_np(raw).reshape(n, 14).astype(np.float64))
        g = _np(dap).reshape(n, 14) * scale * (1 - th * th)
        if noise is not None:
            eps_t = max(eps_end, eps_start - eps_decay * self._t(ctrl))
            pre = scale * th + (lo + scale) + eps_t * _np(noise).reshape(n, 14)
            g = np.where((pre < lo) | (pre > hi), 0.0, g)
        _put(dout, g)

    def gauss_head(self, obs, raw, eps, deterministic, ap_out, logp_out):
        """GaussianSharedPolicy.forward (model/policy.py:53-66) with the volatile box (evopf.py:769-783) + the clip of
        PDSAC_PA.take_action (agent/sac_pa.py:111); raw [n,28] = (mean | log-std heads)."""
        n = ap_out.numel() // 14
        lo, hi = oe.partial_box(_np(obs).astype(np.float64))
        scale, r, e = (hi - lo) * 0.5, _np(raw).reshape(n, 28).astype(np.float64), _np(eps).reshape(n, 14).astype(np.float64)
        ls = np.clip(r[:, 14:] - 3.0, -23.0, -2.0)
        y = np.tanh(r[:, :14] + e * np.exp(ls))
        if logp_out is not None:
            _put(logp_out, (-0.5 * e * e - ls - 0.9189385332046727 - np.log(scale * (1 - y * y) + 1e-6)).sum(axis=1))
        a = scale * (np.tanh(r[:, :14]) if deterministic else y) + lo + scale
        _put(ap_out, np.clip(a, lo, hi))

    def gauss_head_bwd(self, obs, raw, eps, dap, dlogp, draw):
        n = draw.numel() // 28
        lo, hi = oe.partial_box(_np(obs).astype(np.float64))
        scale, r, e = (hi - lo) * 0.5, _np(raw).reshape(n, 28).astype(np.float64), _np(eps).reshape(n, 14).astype(np.float64)
        lsr = r[:, 14:] - 3.0
        sd = np.exp(np.clip(lsr, -23.0, -2.0))
        y = np.tanh(r[:, :14] + e * sd)
        omy = 1 - y * y
        a = scale * y + lo + scale
        gx = _np(dap).reshape(n, 14) * scale * omy * ((a >= lo) & (a <= hi)) + dlogp * (2 * scale * y * omy) / (scale * omy + 1e-6)
        dls = (gx * e * sd - dlogp) * ((lsr >= -23.0) & (lsr <= -2.0))
        _put(draw, np.concatenate([gx, dls], axis=1))

    def complete_bwd(self, obs, grad_action, grad_ap, action=None):
        a = _np(action).astype(np.float64)
        jac = oe.eq_jac(a)
        jn = jac[:, oe.GRID.keep_constr][:, :, oe.GRID.newton_vars]
        _put(grad_ap, oe.complete_partial_bwd(_np(grad_action).astype(np.float64), jac, jn))

    def resid(self, obs, action, eq_out, ineq_out):
        s, a = _np(obs).astype(np.float64), _np(action).astype(np.float64)
        if eq_out is not None:
            _put(eq_out, oe.eq_resid(s, a))
        if ineq_out is not None:
            _put(ineq_out, oe.ineq_resid(s, a))

    def ineq_partial_grad(self, obs, action, step_out):
        _put(step_out, oe.ineq_partial_grad(_np(obs).astype(np.float64), _np(action).astype(np.float64)))

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None):
        s, a, v = _np(obs).astype(np.float64), _np(action).astype(np.float64), _np(nu).astype(np.float64)
        g = oe.ineq_resid(s, a)
        dist = np.maximum(g, 0.0)
        loss_out += float(scale * (dist @ v).sum())
        if grad_action is not None:
            _put(grad_action, scale * (((g > 0) * v) @ oe.ineq_jac()))
        if grad_nu is not None:
            grad_nu += torch.as_tensor(scale * dist.sum(axis=0), dtype=grad_nu.dtype)


class PendulumKernels(_EnvKernels):
    name = "SpringPendulum-v0"
    obs_dim, internal_dim, action_dim, eq_num, ineq_num, partial_dim = 5, 4, 2, 1, 1, 1
    row_floats, ring_floats = CONST["RPO_PEND_ROW"], CONST["RPO_PEND_RING"]
    cols = dict(state=(0, 5), action=(5, 7), next_state=(7, 12), reward=(12, 13), done=(13, 14), eq_viol=(14, 15),
                ineq_viol=(15, 16))
    partial = 0

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        n = internal.shape[0]
        s = philox.pendulum_reset(seed, np.arange(n) + env_id_base, _np(ep_count).astype(np.uint32))
        _put(internal, s)
        if obs is not None:
            _put(obs, pd.get_obs(s.astype(np.float64)))
        ep_len.zero_()
        ep_ret.zero_()

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        n, t = internal.shape[0], self._t(ctrl)
        s, a = _np(internal).astype(np.float64), _np(action)
        nxt, nobs, rew, term, ineq, eq = pd.step(s, a)
        mask = self._finish_step(n, pd.get_obs(s).astype(np.float32), a, nobs.astype(np.float32), rew, term, eq, ineq,
                                 ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps, auto_reset,
                                 viol_thresh, t)
        fresh = philox.pendulum_reset(seed, np.arange(n) + env_id_base, _np(ep_count).astype(np.uint32))
        new = np.where(mask[:, None], fresh, nxt.astype(np.float32))
        _put(internal, new)
        if obs is not None:
            _put(obs, pd.get_obs(new.astype(np.float64)))

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None):
        n = action.shape[0]
        ap = self._explore(ap_raw, noise, n, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, seed,
                           env_id_base, self._t(ctrl))
        o = _np(obs).astype(np.float32)
        a, it = pd.grad_steps(o, pd.complete_partial(o, ap), corr_lr, max_steps, corr_eps, corr_momentum)
        _put(action, a)
        if iters is not None:
            _put(iters, it)
        self._stat_iters(stats, ctrl, it)

    def project_batchref(self, obs, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum):
        o = _np(obs).astype(np.float32)
        a, it = pd.grad_steps(o, pd.complete_partial(o, _np(ap).reshape(-1)), corr_lr, max_steps, corr_eps,
                              corr_momentum, batch_global_stop=True, batched_reference=True)
        _put(action, a)
        if iters_out is not None:
            iters_out[0] = int(it.max())

    def complete_bwd(self, obs, grad_action, grad_ap, action=None):
        o, g = _np(obs), _np(grad_action)
        _put(grad_ap, g[:, 0] - g[:, 1] * (o[:, 1] * (np.float32(1) / o[:, 0])))

    def resid(self, obs, action, eq_out, ineq_out):
        if eq_out is not None:
            _put(eq_out, pd.eq_resid(_np(obs).astype(np.float32), _np(action)))
        if ineq_out is not None:
            _put(ineq_out, pd.ineq_resid(_np(action)))

    def ineq_partial_grad(self, obs, action, step_out):
        _put(step_out, pd.ineq_partial_grad(_np(obs).astype(np.float32), _np(action)))

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None):
        loss, ga, gnu = train_ops.lagrangian_pendulum(_np(action), _np(nu), scale)
        loss_out += float(loss)
        if grad_action is not None:
            _put(grad_action, ga)
        if grad_nu is not None:
            grad_nu += torch.as_tensor(gnu)


# ---------------------------------------------------------------------------------------------- shared kernels

def philox_normal(out, seed, id_base, salt, stream_tag, ctrl=None):
    t = 0 if ctrl is None else int(ctrl[0])
    n = out.numel()
    sub = 0 if ctrl is None else int(ctrl[2])
    r = philox.draw(seed, np.arange(n) + id_base, (t + salt) & 0xFFFFFFFF, stream_tag, sub)
    _put(out, philox.normal(r[:, 0], r[:, 1]))


def replay_sample_gather(rows, cap_steps, n_envs, out, idx_out, seed, salt, ctrl):
    t = int(ctrl[0])
    idx = philox.sample_indices(seed, out.shape[0], t, salt, min(t, cap_steps) * n_envs, int(ctrl[2]))
    out.copy_(rows[torch.as_tensor(idx)][:, :out.shape[1]])      # (ring rows may be wider than a transition: ring_floats)
    if idx_out is not None:
        _put(idx_out, idx)


def replay_gather(rows, idx, out):
    out.copy_(rows[idx][:, :out.shape[1]])


def td_huber(q1, q2, qn1, qn2, logp, alpha, reward, done, gamma, loss_out, grad_q1, grad_q2, target_out=None):
    opt = lambda x: None if x is None else _np(x)   # noqa: E731
    loss, y, g1, g2 = train_ops.td_huber(_np(q1), _np(qn1), _np(reward).reshape(-1), _np(done).reshape(-1), gamma,
                                         q2=opt(q2), qn2=opt(qn2), logp=opt(logp), alpha=alpha)
    loss_out += float(loss)
    if grad_q1 is not None:
        _put(grad_q1, g1)
    if grad_q2 is not None and g2 is not None:
        _put(grad_q2, g2)
    if target_out is not None:
        _put(target_out, y)


def absmax(x, max_out):
    max_out[0] = max(float(max_out[0]), float(x.abs().max()))


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              maximize=False, clip_thres=0.0, gradmax=None, reset_gradmax=True, clamp_min0=False, target=None, tau=0.0,
              zero_grad=False):
    g = _np(grad)
    if clip_thres > 0:
        coef = min(np.float32(clip_thres) / (np.float32(float(gradmax.max())) + np.float32(1e-6)), np.float32(1.0))
        g *= np.float32(coef)                         # in place, like clip_grad_norm_
    step = train_ops.adam_step(_np(param), g, _np(exp_avg), _np(exp_avg_sq), int(step_dev[0]), lr, beta1, beta2, eps,
                               weight_decay, maximize, clamp_min0)
    step_dev[0] = step
    if reset_gradmax and gradmax is not None:
        gradmax.zero_()
    if target is not None:
        train_ops.polyak(_np(param), _np(target), tau)
    if zero_grad:
        grad.zero_()


def polyak(param, target, tau):
    train_ops.polyak(_np(param), _np(target), tau)


# ---------------------------------------------------------------------------------------------- MLP kernels (torch-CPU)

def mlp_supported(E, H, cat=False):
    return H == 256 and (2 * E if cat else E) in (128, 256, 512)


class MlpDesc(object):
    FIELDS = ("Ws", "bs", "Wa", "ba", "W0", "b0", "W1", "b1", "W1b", "b1b")

    def __init__(self, tensors, S, A, E, H, n_out, cat, head_dim=1):
        self.tensors = {k: tensors.get(k) for k in self.FIELDS}
        self.S, self.A, self.E, self.H, self.n_out, self.cat = S, A, E, H, n_out, int(bool(cat))
        self.head_dim = int(head_dim)
        self.ein = E * (2 if cat else 1)


def _x0(d, s, a):
    t = {k: (None if v is None else v.detach()) for k, v in d.tensors.items()}
    xs = s @ t["Ws"].T + t["bs"]
    if d.A == 0:
        return xs
    xa = a @ t["Wa"].T + t["ba"]
    return torch.cat([xs, xa], dim=1) if d.cat else xs + xa


def mlp_forward(d, s, a, out, x0_save=None, h1_save=None, out_mode=0, scale=1.0, base=0.0):
    t = {k: (None if v is None else v.detach()) for k, v in d.tensors.items()}
    with torch.no_grad():
        x0 = _x0(d, s, a)
        h1 = torch.relu(x0) @ t["W0"].T + t["b0"]
        hr = torch.relu(h1)
        o = hr @ t["W1"].T + t["b1"]
        if d.n_out > 1:
            o = torch.cat([o, hr @ t["W1b"].T + t["b1b"]], dim=1)
        if out_mode == 1:
            o = torch.cat([scale * torch.tanh(o[:, :1]) + base, o[:, 1:]], dim=1)
        out.copy_(o)
        if x0_save is not None:
            x0_save.copy_(x0)
        if h1_save is not None:
            h1_save.copy_(h1)


class Td(object):
    """Mirror of rpo_amd.ops.Td (rpo_td): TD target + Huber as the prologue of the critic's backward pass."""

    def __init__(self, q, qn1, qn2, logp, reward, done, alpha, gamma, dq_out, loss_partial):
        self.q, self.qn1, self.qn2, self.logp, self.reward, self.done = q, qn1, qn2, logp, reward, done
        self.alpha, self.gamma, self.dq_out, self.loss_partial = float(alpha), float(gamma), dq_out, loss_partial

    def apply(self):
        """rpo_ddpg.py:331-335 / rpo_sac.py:346-353 in float32, loss partials per 16-row tile; returns dout [n, 1]."""
        n = self.dq_out.shape[0]
        qn = self.qn1 if self.qn2 is None else torch.minimum(self.qn1, self.qn2)
        if self.logp is not None:
            qn = qn - self.alpha * self.logp.reshape(-1)
        y = self.reward.reshape(-1) + self.gamma * (1.0 - self.done.reshape(-1)) * qn
        d = self.q - y
        ad = d.abs()
        hub = torch.where(ad < 1.0, 0.5 * d * d, ad - 0.5) / n
        self.dq_out.copy_(torch.clamp(d, -1.0, 1.0) / n)
        tiles = self.loss_partial.shape[0]
        pad = torch.zeros(tiles * 16)
        pad[:n] = hub
        self.loss_partial.copy_(pad.view(tiles, 16).sum(1))
        return self.dq_out.view(n, 1)


def mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da=None, param_grads=True, first_layer_state_only=False, gradmax=None,
                 td=None):
    if td is not None:
        dout = td.apply()
    _mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, param_grads, first_layer_state_only)
    if gradmax is not None and param_grads:
        with torch.no_grad():
            keys = ("Ws", "bs") if first_layer_state_only else tuple(d.tensors)
            m = max(float(d.tensors[k].grad.abs().max()) for k in keys if d.tensors[k] is not None)
            gradmax[0] = max(float(gradmax[0]), m)


def _mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da=None, param_grads=True, first_layer_state_only=False):
    t = {k: (None if v is None else v.detach()) for k, v in d.tensors.items()}
    g = {k: (None if v is None else v.grad) for k, v in d.tensors.items()}
    with torch.no_grad():
        hr, xr = torch.relu(h1), torch.relu(x0)
        w1 = t["W1"] if d.n_out == 1 else torch.cat([t["W1"], t["W1b"]], dim=0)
        dh.copy_((dout @ w1) * (h1 > 0))
        dx0.copy_((dh @ t["W0"]) * (x0 > 0))
        E = d.E
        dxs = dx0[:, :E]
        dxa = dx0[:, E:] if d.cat else dx0
        if da is not None:
            da.copy_(dxa @ t["Wa"])
        if not param_grads:
            return
        g["Ws"] += dxs.T @ s
        g["bs"] += dxs.sum(0)
        if first_layer_state_only:
            return
        if d.A > 0:
            g["Wa"] += dxa.T @ a
            g["ba"] += dxa.sum(0)
        g["W0"] += dh.T @ xr
        g["b0"] += dh.sum(0)
        hd = max(1, getattr(d, "head_dim", 1))
        g["W1"] += dout[:, :hd].T @ hr
        g["b1"] += dout[:, :hd].sum(0) if hd > 1 else dout[:, 0].sum()
        if d.n_out > 1:
            g["W1b"] += dout[:, hd:2 * hd].T @ hr
            g["b1b"] += dout[:, hd:2 * hd].sum(0) if hd > 1 else dout[:, 1].sum()


def tanh_box_bwd(dap, ap_det, noise, eps_start, eps_end, eps_decay, ctrl, lo, hi, scale, base, dout):
    t = 0 if ctrl is None else int(ctrl[0])
    eps_t = max(eps_end, eps_start - eps_decay * t)
    y = (ap_det - base) / scale
    g = dap * scale * (1 - y * y)
    if noise is not None:
        x = ap_det + eps_t * noise
        g = g * ((x >= lo) & (x <= hi))
    dout.copy_(g)


def gauss_head(raw, eps, scale, base, lo, hi, deterministic, ap_out, logp_out=None):
    mean, ls = raw[:, 0], torch.clamp(raw[:, 1] - 3, -23, -2)
    e = eps.reshape(-1)
    y = torch.tanh(mean + e * ls.exp())
    if logp_out is not None:
        logp_out.copy_((-0.5 * e * e - ls - 0.9189385332046727 - torch.log(scale * (1 - y * y) + 1e-6)).reshape(logp_out.shape))
    a = scale * torch.tanh(mean) + base if deterministic else scale * y + base
    ap_out.copy_(torch.clamp(a, lo, hi).reshape(ap_out.shape))


def gauss_head_bwd(raw, eps, dap, dlogp, scale, base, lo, hi, draw):
    mean, lsr = raw[:, 0], raw[:, 1] - 3
    ls = torch.clamp(lsr, -23, -2)
    e, sd = eps.reshape(-1), ls.exp()
    y = torch.tanh(mean + e * sd)
    omy = 1 - y * y
    a = scale * y + base
    g_ap = dap.reshape(-1) * scale * omy * ((a >= lo) & (a <= hi))
    gx = g_ap + dlogp * (2 * scale * y * omy) / (scale * omy + 1e-6)
    dls = (gx * e * sd - dlogp) * ((lsr >= -23) & (lsr <= -2))
    draw.copy_(torch.stack([gx, dls], dim=1))
