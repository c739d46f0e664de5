"""Tensor-level wrappers over the C ABI (include/rpo_hip.h).  PyTorch only provides device memory and the stream.

Every wrapper launches on ``torch.cuda.current_stream()`` so that the calls are ordered with the surrounding torch ops
and are captured when that stream is being captured into a hipGraph.  Tensors must already live on the GPU as
contiguous float32 / int32 / int64: there is no CPU path here.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import CONST, RpoHipError, check

NOISE_NONE = CONST["RPO_NOISE_NONE"]
NOISE_EXPLICIT = CONST["RPO_NOISE_EXPLICIT"]
NOISE_PHILOX = CONST["RPO_NOISE_PHILOX"]
NOISE_UNIFORM = CONST["RPO_NOISE_UNIFORM"]
NOISE_CLIP_ONLY = CONST["RPO_NOISE_CLIP_ONLY"]
STREAM_POLICY = CONST["RPO_STREAM_POLICY"]
ADAM_CLOCK = True           # rpo_adam_step / _multi take the update-clock argument (trainer._uctrl)
STATS_LEN = CONST["RPO_STATS_LEN"]
STATS_SUB = CONST["RPO_STATS_SUB"]
CTRL_LEN = CONST["RPO_CTRL_LEN"]
PROJ_WS_WORDS = CONST["RPO_PROJ_WS_WORDS"]
PROJ_WS_GAVE_UP = CONST["RPO_PROJ_WS_GAVE_UP"]
STAT = {k[len("RPO_STAT_"):].lower(): v for k, v in CONST.items()
        if k.startswith("RPO_STAT_") and k not in ("RPO_STATS_LEN", "RPO_STATS_SUB")}


TUNE = {k[len("RPO_TUNE_"):].lower(): v for k, v in CONST.items() if k.startswith("RPO_TUNE_") and k != "RPO_TUNE_COUNT"}


class tuning(object):
    """Kernel-variant switches of the library (include/rpo_hip.h: rpo_tuning, RPO_TUNE_*; the library reads no environment
    variable).  ``tuning(fwd_stream=0)`` sets them at once; used as a context manager the previous values come back on exit:

        with ops.tuning(bwd_onepass=0):
            ...                                  # the two-launch backward
    ``tuning.get(name)`` queries one.  A/B tests and measurements only: the defaults are what ships."""

    def __init__(self, **values):
        lib = _lib.load()
        self._old = {}
        for name, value in values.items():
            if name not in TUNE:
                raise RpoHipError("unknown tuning key %r (have: %s)" % (name, ", ".join(sorted(TUNE))))
            old = lib.rpo_tuning(TUNE[name], int(value))
            if old < 0:
                raise RpoHipError("rpo_tuning(%s) failed" % name)
            self._old[name] = old

    @staticmethod
    def get(name):
        return _lib.load().rpo_tuning(TUNE[name], -1)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        for name, old in self._old.items():
            lib.rpo_tuning(TUNE[name], old)
        return False


def new_stats(stats_cap, device):
    """Statistics buffer [stats_cap, RPO_STATS_SUB, RPO_STATS_LEN] (see include/rpo_hip.h)."""
    return torch.zeros(int(stats_cap), STATS_SUB, STATS_LEN, device=device)


def reduce_stats(rows):
    """[..., RPO_STATS_SUB, RPO_STATS_LEN] -> [..., RPO_STATS_LEN]: sum over the sub-rows, max for the *_max slots."""
    out = rows.sum(dim=-2)
    for key in ("max_ineq_max", "max_eq_max"):
        out[..., STAT[key]] = rows[..., STAT[key]].max(dim=-1).values
    return out


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=torch.float32, allow_none=False, contiguous=True):
    if t is None:
        if allow_none:
            return None
        raise RpoHipError("required tensor is None")
    if not t.is_cuda:
        raise RpoHipError("rpo_amd kernels need GPU tensors (got %s); there is no CPU fallback" % t.device)
    if t.dtype != dtype:
        raise RpoHipError("expected %s, got %s" % (dtype, t.dtype))
    if contiguous and not t.is_contiguous():
        raise RpoHipError("expected a contiguous tensor, got strides %s" % (t.stride(),))
    return ctypes.c_void_p(t.data_ptr())


def _row_view(t, width):
    """(pointer, row stride in floats) of a [n, width] float32 view whose rows are contiguous."""
    if t.dim() != 2 or t.shape[1] != width or t.stride(1) != 1:
        raise RpoHipError("expected a [n,%d] row view with unit column stride, got %s / %s" % (width, tuple(t.shape), t.stride()))
    # the stride of a 1-row view is arbitrary (torch keeps whatever the producer left there)
    return _p(t, contiguous=False), int(t.stride(0)) if t.shape[0] > 1 else width


def _col_view(t):
    """(pointer, element stride) of a 1-column view (a column of a gathered batch, or a contiguous vector)."""
    if t.dim() == 2:
        if t.shape[1] != 1:
            raise RpoHipError("expected [n,1], got %s" % (tuple(t.shape),))
        return _p(t, contiguous=False), int(t.stride(0))
    return _p(t, contiguous=False), int(t.stride(0))


def _ring(rows, ring_floats, allow_none=True):
    """Pointer of a replay ring [cap_steps * n_envs, ring_floats].  The ring stride is a COMPILE-TIME constant of the step,
    rollout, rider and fused-sampling kernels (RPO_CART_RING / RPO_PEND_RING / RPO_EVOPF_ROW): a ring of another width --
    e.g. [., row_floats] rows of ABI <= 3 callers -- would be written and read out of bounds, so it is refused here."""
    if rows is None:
        return _p(rows, allow_none=allow_none)
    if rows.dim() != 2 or rows.shape[-1] != ring_floats:
        raise RpoHipError("replay ring must be [rows, %d] float32 (the kernels' compiled ring stride), got %s"
                          % (ring_floats, tuple(rows.shape)))
    return _p(rows)


def _host_ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p)


# =================================================================================================== shared kernels

def philox_fill(out, seed, id_base, index, stream_tag):
    check(_lib.load().rpo_philox_fill(out.shape[0], _p(out, torch.int32), seed, id_base, index, stream_tag, _stream()),
          "rpo_philox_fill")


def philox_normal(out, seed, id_base, salt, stream_tag, ctrl=None):
    check(_lib.load().rpo_philox_normal(out.numel(), _p(out), seed, id_base, salt, stream_tag,
                                        _p(ctrl, torch.int64, allow_none=True), _stream()), "rpo_philox_normal")


def replay_gather(rows, idx, out):
    check(_lib.load().rpo_replay_gather(_p(rows), rows.shape[-1], out.shape[-1], idx.shape[0], _p(idx, torch.int64), _p(out),
                                        _stream()), "rpo_replay_gather")


def replay_sample_gather(rows, cap_steps, n_envs, out, idx_out, seed, salt, ctrl):
    # rows: the ring [cap_steps * n_envs, ring_floats]; out: the batch [B, row_floats] (ring_floats >= row_floats: CartSafe rings
    # keep one 96-byte transition per 128-byte line)
    check(_lib.load().rpo_replay_sample_gather(_p(rows), rows.shape[-1], out.shape[-1], cap_steps, n_envs, out.shape[0], _p(out),
                                               _p(idx_out, torch.int64, allow_none=True), seed, salt,
                                               _p(ctrl, torch.int64), _stream()), "rpo_replay_sample_gather")


def td_huber(q1, q2, qn1, qn2, logp, alpha, reward, done, gamma, loss_out, grad_q1, grad_q2, target_out=None):
    n = q1.shape[0]
    rp, rs = _col_view(reward)
    dp, ds = _col_view(done)
    check(_lib.load().rpo_td_huber(n, _p(q1), _p(q2, allow_none=True), _p(qn1), _p(qn2, allow_none=True),
                                   _p(logp, allow_none=True), alpha, rp, rs, dp, ds, gamma, _p(loss_out),
                                   _p(grad_q1, allow_none=True), _p(grad_q2, allow_none=True),
                                   _p(target_out, allow_none=True), _stream()), "rpo_td_huber")


def absmax(x, max_out):
    """max |x| into max_out[0], or -- when max_out is a gradmax buffer (RPO_GRADMAX_LEN) -- spread over its slots."""
    if max_out.numel() >= CONST["RPO_GRADMAX_LEN"]:
        check(_lib.load().rpo_absmax_slots(x.numel(), _p(x), _p(max_out), _stream()), "rpo_absmax_slots")
    else:
        check(_lib.load().rpo_absmax(x.numel(), _p(x), _p(max_out), _stream()), "rpo_absmax")


def adam_step(param, grad, exp_avg, exp_avg_sq, step_dev, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              maximize=False, clip_thres=0.0, gradmax=None, reset_gradmax=True, clamp_min0=False, target=None,
              tau=0.0, zero_grad=False, clock=None, prepared=False):
    if step_dev.numel() < CONST["RPO_ADAM_STATE_LEN"]:
        raise RpoHipError("step_dev must be int32[RPO_ADAM_STATE_LEN]: {step, pad, arrival word, cached bias corrections, "
                          "sub-counters} (include/rpo_hip.h)")
    check(_lib.load().rpo_adam_step(param.numel(), _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq),
                                    _p(step_dev, torch.int32), lr, beta1, beta2, eps, weight_decay, int(maximize),
                                    clip_thres, _p(gradmax, allow_none=True), int(reset_gradmax), int(zero_grad), int(clamp_min0),
                                    _p(target, allow_none=True), tau, _p(clock, torch.int64, allow_none=True), int(prepared),
                                    _stream()),
          "rpo_adam_step")


class _AdamSegStruct(ctypes.Structure):
    _fields_ = [("n", ctypes.c_longlong), ("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("step_dev", ctypes.c_void_p), ("lr", ctypes.c_float),
                ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("weight_decay", ctypes.c_float), ("maximize", ctypes.c_int), ("clip_thres", ctypes.c_float),
                ("gradmax", ctypes.c_void_p), ("reset_gradmax", ctypes.c_int), ("zero_grad", ctypes.c_int),
                ("clamp_min0", ctypes.c_int), ("target", ctypes.c_void_p), ("tau", ctypes.c_float),
                ("target2", ctypes.c_void_p), ("n2", ctypes.c_longlong), ("polyak_only", ctypes.c_int),
                ("prepared", ctypes.c_int)]


def adam_step_multi(segs, clock=None, prepared=False):
    """One launch for up to four non-overlapping optimiser slices (rpo_adam_step_multi).  Each entry is a dict with the
    keyword arguments of ``adam_step`` (+ ``target2`` / ``n2``), or ``dict(polyak_only=True, param=, target=, tau=)``."""
    arr = (_AdamSegStruct * len(segs))()
    vp = lambda t, dt=torch.float32: None if t is None else _p(t, dt).value                      # noqa: E731
    for a, g in zip(arr, segs):
        a.n, a.param, a.tau = g["param"].numel(), vp(g["param"]), g.get("tau", 0.0)
        a.target, a.target2, a.n2 = vp(g.get("target")), vp(g.get("target2")), int(g.get("n2", 0))
        a.polyak_only = int(bool(g.get("polyak_only", False)))
        if a.polyak_only:
            continue
        a.grad, a.exp_avg, a.exp_avg_sq = vp(g["grad"]), vp(g["exp_avg"]), vp(g["exp_avg_sq"])
        if g["step_dev"].numel() < CONST["RPO_ADAM_STATE_LEN"]:
            raise RpoHipError("step_dev must be int32[RPO_ADAM_STATE_LEN] (include/rpo_hip.h)")
        a.step_dev = vp(g["step_dev"], torch.int32)
        a.lr, a.beta1, a.beta2, a.eps = g["lr"], g.get("beta1", 0.9), g.get("beta2", 0.999), g.get("eps", 1e-8)
        a.weight_decay, a.maximize, a.clip_thres = g.get("weight_decay", 0.0), int(g.get("maximize", False)), g.get("clip_thres", 0.0)
        a.gradmax, a.reset_gradmax = vp(g.get("gradmax")), int(g.get("reset_gradmax", True))
        a.zero_grad, a.clamp_min0 = int(g.get("zero_grad", False)), int(g.get("clamp_min0", False))
        a.prepared = int(bool(prepared))
    check(_lib.load().rpo_adam_step_multi(len(segs), arr, _p(clock, torch.int64, allow_none=True), _stream()),
          "rpo_adam_step_multi")


def min_q_bwd(q1, q2, scale, dq1, dq2):
    """d(-min(q1, q2))/dq * (-scale): see rpo_min_q_bwd."""
    check(_lib.load().rpo_min_q_bwd(q1.numel(), _p(q1), _p(q2), scale, _p(dq1), _p(dq2), _stream()), "rpo_min_q_bwd")


def polyak(param, target, tau):
    check(_lib.load().rpo_polyak(param.numel(), _p(param), _p(target), tau, _stream()), "rpo_polyak")


# =================================================================================================== env kernel sets

class CartSafeKernels(object):
    """HIP kernels of CartSafe-v0.  ``consts`` is the float32[35] table of include/rpo_hip.h (RPO_CART_CONSTS_LEN)."""

    name = "CartSafe-v0"
    obs_dim, internal_dim, action_dim, eq_num, ineq_num, partial_dim = 6, 6, 2, 1, 6, 1
    row_floats = CONST["RPO_CART_ROW"]
    ring_floats = CONST["RPO_CART_RING"]       # floats between consecutive rows of a replay ring (>= row_floats)
    # column ranges of a transition row
    cols = dict(state=(0, 6), action=(6, 8), next_state=(8, 14), reward=(14, 15), done=(15, 16), eq_viol=(16, 17),
                ineq_viol=(17, 23))

    def __init__(self, consts, partial):
        self.consts = np.ascontiguousarray(consts, dtype=np.float32)     # host table, copied into each launch
        if self.consts.shape != (CONST["RPO_CART_CONSTS_LEN"],):
            raise RpoHipError("bad CartSafe constant table")
        self.partial = int(partial)

    @property
    def _cptr(self):
        return _host_ptr(self.consts)

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        # the observation IS the internal state for this env: `obs` aliases `internal`
        check(_lib.load().rpo_cartsafe_reset(internal.shape[0], _p(internal), _p(ep_len, torch.int32), _p(ep_ret),
                                             _p(ep_count, torch.int32), seed, env_id_base, _stream()),
              "rpo_cartsafe_reset")

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        check(_lib.load().rpo_cartsafe_step(
            internal.shape[0], _p(internal), _p(action), _p(ep_len, torch.int32), _p(ep_ret),
            _p(ep_count, torch.int32), _ring(rows, self.ring_floats), cap_steps, _p(stats, allow_none=True),
            0 if stats is None else stats.shape[0], _p(ctrl, torch.int64, allow_none=True), self._cptr, self.partial,
            max_episode_steps, int(auto_reset), viol_thresh, seed, env_id_base, _stream()), "rpo_cartsafe_step")

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None):
        check(_lib.load().rpo_cartsafe_act_project(
            action.shape[0], _p(ap_raw, allow_none=True), _p(noise, allow_none=True), _p(action),
            _p(iters, torch.int32, allow_none=True), noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
            max_steps, corr_lr, corr_eps, corr_momentum, self._cptr, self.partial, seed, env_id_base,
            _p(ctrl, torch.int64, allow_none=True), _p(stats, allow_none=True),
            0 if stats is None else stats.shape[0], _stream()), "rpo_cartsafe_act_project")

    # ---- fused pipelines (rpo_amd/csrc/fused.hip) ----------------------------------------------------------------
    def rollout(self, actor_desc, gauss, scale, base, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps,
                stats, ctrl, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, max_steps, corr_lr, corr_eps,
                corr_momentum, max_episode_steps, auto_reset, viol_thresh, seed, env_id_base, defer_clock=False):
        net = actor_desc.net_struct()
        check(_lib.load().rpo_cartsafe_rollout(
            ctypes.byref(net), int(gauss), scale, base, internal.shape[0], _p(internal), _p(action),
            _p(ep_len, torch.int32), _p(ep_ret), _p(ep_count, torch.int32), _ring(rows, self.ring_floats), cap_steps,
            _p(stats, allow_none=True), 0 if stats is None else stats.shape[0], _p(ctrl, torch.int64), noise_mode,
            eps_start, eps_end, eps_decay, box_lo, box_hi, max_steps, corr_lr, corr_eps, corr_momentum, self._cptr,
            self.partial, max_episode_steps, int(auto_reset), viol_thresh, seed, env_id_base, int(defer_clock), _stream()),
            "rpo_cartsafe_rollout")

    def ddpg_critic_forward(self, actor_target, critic_target, critic, scale, base, rows, cap_steps, n_envs, batch_out,
                            idx_out, idx_in, seed, salt, ctrl, max_steps, corr_lr, corr_eps, corr_momentum, box_lo, box_hi,
                            q_out, qn_out, x0_save, h1_save):
        at, ct, cr = actor_target.net_struct(), critic_target.net_struct(), critic.net_struct()
        check(_lib.load().rpo_cartsafe_ddpg_critic_forward(
            ctypes.byref(at), ctypes.byref(ct), ctypes.byref(cr), scale, base, _ring(rows, self.ring_floats, False), cap_steps, n_envs,
            batch_out.shape[0], _p(batch_out), _p(idx_out, torch.int64, allow_none=True),
            _p(idx_in, torch.int64, allow_none=True), seed, salt, _p(ctrl, torch.int64), max_steps, corr_lr, corr_eps,
            corr_momentum, box_lo, box_hi, self._cptr, self.partial, _p(q_out), _p(qn_out), _p(x0_save), _p(h1_save),
            _stream()), "rpo_cartsafe_ddpg_critic_forward")

    def sac_critic_forward(self, actor, critic_target1, critic_target2, critic1, critic2, scale, base, rows, cap_steps,
                           n_envs, batch_out, idx_out, idx_in, eps_in, sample_seed, sample_salt, noise_seed, noise_id_base,
                           noise_salt, ctrl, max_steps, corr_lr, corr_eps, corr_momentum, box_lo, box_hi,
                           q1_out, q2_out, qn1_out, qn2_out, logp_out, x0_save1, h1_save1, x0_save2, h1_save2):
        nets = [d.net_struct() for d in (actor, critic_target1, critic_target2, critic1, critic2)]
        check(_lib.load().rpo_cartsafe_sac_critic_forward(
            *[ctypes.byref(n) for n in nets], scale, base, _ring(rows, self.ring_floats, False), cap_steps, n_envs, batch_out.shape[0], _p(batch_out),
            _p(idx_out, torch.int64, allow_none=True), _p(idx_in, torch.int64, allow_none=True),
            _p(eps_in, allow_none=True), sample_seed, sample_salt, noise_seed, noise_id_base, noise_salt,
            _p(ctrl, torch.int64), max_steps, corr_lr, corr_eps, corr_momentum, box_lo, box_hi, self._cptr, self.partial,
            _p(q1_out), _p(q2_out), _p(qn1_out), _p(qn2_out), _p(logp_out), _p(x0_save1), _p(h1_save1),
            _p(x0_save2), _p(h1_save2), _stream()), "rpo_cartsafe_sac_critic_forward")

    def complete_bwd(self, obs, grad_action, grad_ap, action=None):
        check(_lib.load().rpo_cartsafe_complete_bwd(grad_action.shape[0], _p(grad_action), _p(grad_ap), self._cptr,
                                                    self.partial, _stream()), "rpo_cartsafe_complete_bwd")

    def resid(self, obs, action, eq_out, ineq_out):
        check(_lib.load().rpo_cartsafe_resid(action.shape[0], _p(action), _p(eq_out, allow_none=True),
                                             _p(ineq_out, allow_none=True), self._cptr, self.partial, _stream()),
              "rpo_cartsafe_resid")

    def ineq_partial_grad(self, obs, action, step_out):
        check(_lib.load().rpo_cartsafe_ineq_partial_grad(action.shape[0], _p(action), _p(step_out), self._cptr,
                                                         self.partial, _stream()), "rpo_cartsafe_ineq_partial_grad")

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None):
        check(_lib.load().rpo_cartsafe_lagrangian(action.shape[0], _p(action), _p(nu), scale, _p(loss_out),
                                                  _p(grad_action, allow_none=True), _p(grad_nu, allow_none=True),
                                                  self._cptr, self.partial, _stream()), "rpo_cartsafe_lagrangian")


class EvopfKernels(object):
    """HIP kernels of EVOPF-v0 (one wavefront per lane).  ``consts`` = float32[RPO_EVOPF_CONSTS_LEN] built by
    rpo_amd/env/electrical_grid/case14.py; it is uploaded once per device."""

    name = "EVOPF-v0"
    obs_dim = internal_dim = CONST["RPO_EVOPF_STATE"]
    action_dim, partial_dim = CONST["RPO_EVOPF_ACTION"], CONST["RPO_EVOPF_PARTIAL"]
    eq_num, ineq_num = CONST["RPO_EVOPF_EQ"], CONST["RPO_EVOPF_INEQ"]
    row_floats = ring_floats = CONST["RPO_EVOPF_ROW"]
    cols = dict(state=(0, 57), action=(57, 100), next_state=(100, 157), reward=(157, 158), done=(158, 159),
                eq_viol=(159, 187), ineq_viol=(187, 245))
    newton_tol, newton_max_iters = 1e-5, 50          # PFFunction(env, tol=1e-5, bsz=256, max_iters=50), evopf.py:786
    episode_steps = 24                               # the loaders run out of data after one day (demand.py:71)
    partial = 0
    CASE14_ADJ = (19, 31, 14, 350, 59, 7216, 456, 192, 9032, 1792, 1568, 6176, 14368, 12544)   # == kAdjMask, evopf_dev.h

    def __init__(self, consts):
        self.consts = np.ascontiguousarray(consts, dtype=np.float32)
        if self.consts.shape != (CONST["RPO_EVOPF_CONSTS_LEN"],):
            raise RpoHipError("bad EVOPF constant table")
        self.consts = self.consts.copy()
        # The solver eliminates in a static order with case14's branch pattern compiled in (csrc/evopf_dev.h, kAdjMask): a
        # Ybus with entries outside that pattern -- or RPO_EVOPF_PIVOT=dynamic -- selects partial pivoting instead.
        yr, yi = (self.consts[CONST[k]:CONST[k] + 196].reshape(14, 14) for k in ("RPO_EVOPF_C_YR", "RPO_EVOPF_C_YI"))
        inside = np.array([[(m >> k) & 1 for k in range(14)] for m in self.CASE14_ADJ], dtype=bool)
        dynamic = bool((((yr != 0) | (yi != 0)) & ~inside).any()) or os.environ.get("RPO_EVOPF_PIVOT", "") == "dynamic"
        # 1.0 = "validated: static order allowed"; everything else (the 0 of a table built elsewhere) = partial pivoting
        self.consts[CONST["RPO_EVOPF_C_FLAGS"]] = 0.0 if dynamic else 1.0
        self.static_order = not dynamic
        self._dev = {}

    def _c(self, like):
        key = like.device
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(self.consts).to(like.device)
        return _p(self._dev[key])

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        check(_lib.load().rpo_evopf_reset(internal.shape[0], _p(internal), _p(ep_len, torch.int32), _p(ep_ret),
                                          _p(ep_count, torch.int32), self._c(internal), seed, env_id_base, _stream()),
              "rpo_evopf_reset")

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        check(_lib.load().rpo_evopf_step(
            internal.shape[0], _p(internal), _p(action), _p(ep_len, torch.int32), _p(ep_ret), _p(ep_count, torch.int32),
            _ring(rows, self.ring_floats), cap_steps, _p(stats, allow_none=True), 0 if stats is None else stats.shape[0],
            _p(ctrl, torch.int64, allow_none=True), max_episode_steps, int(auto_reset), viol_thresh, self._c(internal),
            seed, env_id_base, _stream()), "rpo_evopf_step")

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None,
                    ap_is_raw=False):
        # box_lo / box_hi are ignored: the box is state dependent (EVOPFEnv.update) and evaluated inside the kernel
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_act_project(
            action.shape[0], sp, ss, _p(ap_raw, allow_none=True), _p(noise, allow_none=True), _p(action),
            _p(iters, torch.int32, allow_none=True), noise_mode, int(ap_is_raw), eps_start, eps_end, eps_decay, max_steps,
            corr_lr,
            corr_eps, corr_momentum, self.newton_tol, self.newton_max_iters, self._c(action), seed, env_id_base,
            _p(ctrl, torch.int64, allow_none=True), _p(stats, allow_none=True), 0 if stats is None else stats.shape[0],
            _stream()), "rpo_evopf_act_project")

    def tanh_box_bwd(self, obs, raw, noise, eps_start, eps_end, eps_decay, ctrl, dap, dout):
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_tanh_box_bwd(dout.numel() // self.partial_dim, sp, ss, _p(raw), _p(noise, allow_none=True), eps_start,
                                                 eps_end, eps_decay, _p(ctrl, torch.int64, allow_none=True), _p(dap),
                                                 _p(dout), self._c(dout), _stream()), "rpo_evopf_tanh_box_bwd")

    def gauss_head(self, obs, raw, eps, deterministic, ap_out, logp_out):
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_gauss_head(ap_out.numel() // self.partial_dim, sp, ss, _p(raw), _p(eps), int(deterministic), _p(ap_out),
                                               _p(logp_out, allow_none=True), self._c(raw), _stream()),
              "rpo_evopf_gauss_head")

    def gauss_head_bwd(self, obs, raw, eps, dap, dlogp, draw):
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_gauss_head_bwd(draw.numel() // (2 * self.partial_dim), sp, ss, _p(raw), _p(eps), _p(dap), dlogp, _p(draw),
                                                   self._c(raw), _stream()), "rpo_evopf_gauss_head_bwd")

    def complete_bwd(self, obs, grad_action, grad_ap, action=None, grad_action2=None, grad_action_b=None):
        """grad_action_b (the second twin critic's term) / grad_action2 (the Lagrangian's): more dL/dy terms, added inside the
        kernel in that order (the caller saves the elementwise launches)."""
        check(_lib.load().rpo_evopf_complete_bwd(grad_action.shape[0], _p(action), _p(grad_action), _p(grad_action_b, allow_none=True),
                                                 _p(grad_action2, allow_none=True), _p(grad_ap), self._c(action), _stream()),
              "rpo_evopf_complete_bwd")

    def resid(self, obs, action, eq_out, ineq_out):
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_resid(action.shape[0], sp, ss, _p(action), _p(eq_out, allow_none=True),
                                          _p(ineq_out, allow_none=True), self._c(action), _stream()), "rpo_evopf_resid")

    def ineq_partial_grad(self, obs, action, step_out):
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_ineq_partial_grad(action.shape[0], sp, ss, _p(action), _p(step_out),
                                                      self._c(action), _stream()), "rpo_evopf_ineq_partial_grad")

    def eq_vjp(self, action, grad_eq, grad_action, autograd_sign=True):
        check(_lib.load().rpo_evopf_eq_vjp(action.shape[0], _p(action), _p(grad_eq), _p(grad_action), int(autograd_sign),
                                           self._c(action), _stream()), "rpo_evopf_eq_vjp")

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None, overwrite=False):
        """overwrite: loss_out is written, not accumulated (the caller saves the fill launch before it)."""
        sp, ss = _row_view(obs, self.obs_dim)
        check(_lib.load().rpo_evopf_lagrangian(action.shape[0], sp, ss, _p(action), _p(nu), scale,
                                               _p(loss_out, allow_none=True), _p(grad_action, allow_none=True),
                                               _p(grad_nu, allow_none=True), self._c(action), int(bool(overwrite)), _stream()),
              "rpo_evopf_lagrangian")
    fused_adds = True      # complete_bwd(grad_action2=...) and lagrangian(overwrite=...) exist: no torch launches between them


class PendulumKernels(object):
    """HIP kernels of SpringPendulum-v0."""

    name = "SpringPendulum-v0"
    obs_dim, internal_dim, action_dim, eq_num, ineq_num, partial_dim = 5, 4, 2, 1, 1, 1
    row_floats = CONST["RPO_PEND_ROW"]
    ring_floats = CONST["RPO_PEND_RING"]
    cols = dict(state=(0, 5), action=(5, 7), next_state=(7, 12), reward=(12, 13), done=(13, 14), eq_viol=(14, 15),
                ineq_viol=(15, 16))
    partial = 0

    def reset(self, internal, obs, ep_len, ep_ret, ep_count, seed, env_id_base):
        check(_lib.load().rpo_pendulum_reset(internal.shape[0], _p(internal), _p(obs, allow_none=True),
                                             _p(ep_len, torch.int32), _p(ep_ret), _p(ep_count, torch.int32), seed,
                                             env_id_base, _stream()), "rpo_pendulum_reset")

    def step(self, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps, stats, ctrl, max_episode_steps,
             auto_reset, viol_thresh, seed, env_id_base):
        check(_lib.load().rpo_pendulum_step(
            internal.shape[0], _p(internal), _p(obs, allow_none=True), _p(action), _p(ep_len, torch.int32), _p(ep_ret),
            _p(ep_count, torch.int32), _ring(rows, self.ring_floats), cap_steps, _p(stats, allow_none=True),
            0 if stats is None else stats.shape[0], _p(ctrl, torch.int64, allow_none=True), max_episode_steps,
            int(auto_reset), viol_thresh, seed, env_id_base, _stream()), "rpo_pendulum_step")

    def act_project(self, obs, ap_raw, noise, action, iters, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
                    max_steps, corr_lr, corr_eps, corr_momentum, seed=0, env_id_base=0, ctrl=None, stats=None):
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_act_project(
            action.shape[0], op, ostride, _p(ap_raw, allow_none=True), _p(noise, allow_none=True), _p(action),
            _p(iters, torch.int32, allow_none=True), noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi,
            max_steps, corr_lr, corr_eps, corr_momentum, seed, env_id_base, _p(ctrl, torch.int64, allow_none=True),
            _p(stats, allow_none=True), 0 if stats is None else stats.shape[0], _stream()),
            "rpo_pendulum_act_project")

    def rollout(self, actor_desc, gauss, scale, base, internal, obs, action, ep_len, ep_ret, ep_count, rows, cap_steps,
                stats, ctrl, noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, max_steps, corr_lr, corr_eps,
                corr_momentum, max_episode_steps, auto_reset, viol_thresh, seed, env_id_base, defer_clock=False):
        net = actor_desc.net_struct()
        check(_lib.load().rpo_pendulum_rollout(
            ctypes.byref(net), int(gauss), scale, base, internal.shape[0], _p(internal), _p(obs, allow_none=True),
            _p(action), _p(ep_len, torch.int32), _p(ep_ret), _p(ep_count, torch.int32), _ring(rows, self.ring_floats),
            cap_steps, _p(stats, allow_none=True), 0 if stats is None else stats.shape[0], _p(ctrl, torch.int64),
            noise_mode, eps_start, eps_end, eps_decay, box_lo, box_hi, max_steps, corr_lr, corr_eps, corr_momentum,
            max_episode_steps, int(auto_reset), viol_thresh, seed, env_id_base, int(defer_clock), _stream()),
            "rpo_pendulum_rollout")

    def ddpg_critic_front(self, actor_target, scale, base, rows, cap_steps, n_envs, batch_out, idx_out, idx_in, sample_seed,
                          sample_salt, ctrl, ap_out):
        net = actor_target.net_struct()
        check(_lib.load().rpo_pendulum_ddpg_critic_front(
            ctypes.byref(net), scale, base, _ring(rows, self.ring_floats, False), cap_steps, n_envs, batch_out.shape[0], _p(batch_out),
            _p(idx_out, torch.int64, allow_none=True), _p(idx_in, torch.int64, allow_none=True), sample_seed, sample_salt,
            _p(ctrl, torch.int64), _p(ap_out), _stream()), "rpo_pendulum_ddpg_critic_front")

    def ddpg_critic_back(self, critic_target, critic, batch_rows, next_actions, q_out, qn_out, x0_save, h1_save):
        ct, cr = critic_target.net_struct(), critic.net_struct()
        check(_lib.load().rpo_pendulum_ddpg_critic_back(
            ctypes.byref(ct), ctypes.byref(cr), batch_rows.shape[0], _p(batch_rows), _p(next_actions), _p(q_out),
            _p(qn_out), _p(x0_save), _p(h1_save), _stream()), "rpo_pendulum_ddpg_critic_back")

    def sac_critic_front(self, actor, scale, base, box_lo, box_hi, rows, cap_steps, n_envs, batch_out, idx_out, idx_in,
                         eps_in, sample_seed, sample_salt, noise_seed, noise_id_base, noise_salt, ctrl, ap_out, logp_out):
        net = actor.net_struct()
        check(_lib.load().rpo_pendulum_sac_critic_front(
            ctypes.byref(net), scale, base, box_lo, box_hi, _ring(rows, self.ring_floats, False), cap_steps, n_envs, batch_out.shape[0], _p(batch_out),
            _p(idx_out, torch.int64, allow_none=True), _p(idx_in, torch.int64, allow_none=True),
            _p(eps_in, allow_none=True), sample_seed, sample_salt, noise_seed, noise_id_base, noise_salt,
            _p(ctrl, torch.int64), _p(ap_out), _p(logp_out), _stream()), "rpo_pendulum_sac_critic_front")

    def sac_critic_back(self, critic_target1, critic_target2, critic1, critic2, batch_rows, next_actions,
                        q1_out, q2_out, qn1_out, qn2_out, x0_save1, h1_save1, x0_save2, h1_save2):
        nets = [d.net_struct() for d in (critic_target1, critic_target2, critic1, critic2)]
        check(_lib.load().rpo_pendulum_sac_critic_back(
            *[ctypes.byref(n) for n in nets], batch_rows.shape[0], _p(batch_rows), _p(next_actions),
            _p(q1_out), _p(q2_out), _p(qn1_out), _p(qn2_out), _p(x0_save1), _p(h1_save1), _p(x0_save2),
            _p(h1_save2), _stream()), "rpo_pendulum_sac_critic_back")

    def project_batchref(self, obs, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum):
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_project_batchref(action.shape[0], op, ostride, _p(ap), _p(action),
                                                        _p(iters_out, torch.int32, allow_none=True), max_steps, corr_lr,
                                                        corr_eps, corr_momentum, _stream()),
              "rpo_pendulum_project_batchref")

    def project_batchref_ws(self, obs, ap, action, iters_out, max_steps, corr_lr, corr_eps, corr_momentum, ws, store_mode=1):
        """project_batchref on one workgroup per 16 rows; ws: zero-initialised int64[PROJ_WS_WORDS] workspace (kept by the caller)."""
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_project_batchref_ws(action.shape[0], op, ostride, _p(ap), _p(action),
                                                           _p(iters_out, torch.int32, allow_none=True), max_steps, corr_lr,
                                                           corr_eps, corr_momentum, _p(ws, torch.int64), store_mode, _stream()),
              "rpo_pendulum_project_batchref_ws")

    def complete_bwd(self, obs, grad_action, grad_ap, action=None):
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_complete_bwd(grad_action.shape[0], op, ostride, _p(grad_action), _p(grad_ap),
                                                    _stream()), "rpo_pendulum_complete_bwd")

    def resid(self, obs, action, eq_out, ineq_out):
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_resid(action.shape[0], op, ostride, _p(action), _p(eq_out, allow_none=True),
                                             _p(ineq_out, allow_none=True), _stream()), "rpo_pendulum_resid")

    def ineq_partial_grad(self, obs, action, step_out):
        op, ostride = _row_view(obs, 5)
        check(_lib.load().rpo_pendulum_ineq_partial_grad(action.shape[0], op, ostride, _p(action), _p(step_out),
                                                         _stream()), "rpo_pendulum_ineq_partial_grad")

    def lagrangian(self, action, nu, scale, loss_out, grad_action, grad_nu, obs=None):
        check(_lib.load().rpo_pendulum_lagrangian(action.shape[0], _p(action), _p(nu), scale, _p(loss_out),
                                                  _p(grad_action, allow_none=True), _p(grad_nu, allow_none=True),
                                                  _stream()), "rpo_pendulum_lagrangian")


# =================================================================================================== MLP kernels

class _MlpStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("Ws", "bs", "Wa", "ba", "W0", "b0", "W1", "b1", "W1b", "b1b")] + \
               [(n, ctypes.c_int) for n in ("S", "A", "E", "H", "n_out", "cat", "head_dim")]


class _MlpGradStruct(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("Ws", "bs", "Wa", "ba", "W0", "b0", "W1", "b1", "W1b", "b1b")] + \
               [("splitk_scratch", ctypes.c_void_p), ("splitk_floats", ctypes.c_longlong)]


class _TdStruct(ctypes.Structure):
    _fields_ = [("q", ctypes.c_void_p), ("qn1", ctypes.c_void_p), ("qn2", ctypes.c_void_p), ("logp", ctypes.c_void_p),
                ("reward", ctypes.c_void_p), ("reward_stride", ctypes.c_int), ("done", ctypes.c_void_p),
                ("done_stride", ctypes.c_int), ("alpha", ctypes.c_float), ("gamma", ctypes.c_float),
                ("dq_out", ctypes.c_void_p), ("loss_partial", ctypes.c_void_p)]


class Td(object):
    """``rpo_td``: the TD target / Huber loss as the prologue of the critic's backward pass.  q / qn1 / qn2 / logp /
    dq_out are [n] tensors (qn2, logp may be None), reward / done column views of the gathered batch, loss_partial
    [ceil(n / 16)]."""

    def __init__(self, q, qn1, qn2, logp, reward, done, alpha, gamma, dq_out, loss_partial):
        self.q, self.qn1, self.qn2, self.logp, self.reward, self.done = q, qn1, qn2, logp, reward, done
        self.alpha, self.gamma, self.dq_out, self.loss_partial = float(alpha), float(gamma), dq_out, loss_partial

    def struct(self):
        (rp, rs), (dp, ds) = _col_view(self.reward), _col_view(self.done)
        return _TdStruct(_p(self.q), _p(self.qn1), _p(self.qn2, allow_none=True), _p(self.logp, allow_none=True), rp, rs,
                         dp, ds, self.alpha, self.gamma, _p(self.dq_out), _p(self.loss_partial))


def _td_ref(td):
    return None if td is None else ctypes.byref(td.struct())


def mlp_supported(E, H, cat=False):
    return bool(_lib.load().rpo_mlp_supported(int(E), int(H), int(bool(cat))))


class MlpDesc(object):
    """Pointers into the (flat) parameter / gradient buffers of one network, in the layout of ``rpo_mlp``.

    ``tensors`` maps the field names Ws, bs, Wa, ba, W0, b0, W1, b1, W1b, b1b to parameter tensors (absent / None for
    networks without an action input or a second head).  Gradient pointers come from ``tensor.grad``.
    """
    FIELDS = ("Ws", "bs", "Wa", "ba", "W0", "b0", "W1", "b1", "W1b", "b1b")

    def __init__(self, tensors, S, A, E, H, n_out, cat, head_dim=1):
        self.tensors = {k: tensors.get(k) for k in self.FIELDS}
        self.S, self.A, self.E, self.H, self.n_out, self.cat = int(S), int(A), int(E), int(H), int(n_out), int(bool(cat))
        self.head_dim = int(head_dim)
        self.outs = self.n_out * max(1, self.head_dim)
        self.ein = self.E * (2 if self.cat else 1)
        if self.tensors["W0"].data_ptr() % 16:
            raise RpoHipError("W0 must be 16-byte aligned (FlatParams aligns every tensor)")

    def _ptr(self, t):
        return None if t is None else t.data_ptr()

    def net_struct(self):
        return _MlpStruct(*[self._ptr(self.tensors[k]) for k in self.FIELDS], self.S, self.A, self.E, self.H, self.n_out,
                          self.cat, self.head_dim)

    #: scratch of the split-K weights pass (large batches, include/rpo_hip.h rpo_mlp_grad): a float32 tensor or None
    splitk = None

    def grad_struct(self):
        st = _MlpGradStruct(*[None if self.tensors[k] is None else self._ptr(self.tensors[k].grad) for k in self.FIELDS])
        if self.splitk is not None:
            st.splitk_scratch, st.splitk_floats = _p(self.splitk).value, int(self.splitk.numel())
        return st


def mlp_forward(desc, s, a, out, x0_save=None, h1_save=None, out_mode=0, scale=1.0, base=0.0):
    sp, ss = _row_view(s, desc.S)
    ap, as_ = (None, 0) if desc.A == 0 else _row_view(a, desc.A)
    net = desc.net_struct()
    check(_lib.load().rpo_mlp_forward(ctypes.byref(net), out.shape[0], sp, ss, ap, as_, _p(out),
                                      _p(x0_save, allow_none=True), _p(h1_save, allow_none=True), out_mode, scale, base,
                                      _stream()), "rpo_mlp_forward")


def mlp_forward_multi(calls):
    """One launch for up to four same-shaped networks: calls = [(desc, s, a, out, x0_save, h1_save), ...]."""
    k = len(calls)
    nets = [c[0].net_struct() for c in calls]
    svs = [_row_view(c[1], c[0].S) for c in calls]
    avs = [(None, 0) if c[0].A == 0 else _row_view(c[2], c[0].A) for c in calls]
    ptrs = lambda vals: (ctypes.c_void_p * k)(*vals)                                   # noqa: E731
    as_int = lambda v: None if v is None else (v.value if isinstance(v, ctypes.c_void_p) else v)   # noqa: E731
    check(_lib.load().rpo_mlp_forward_multi(
        k, ptrs([ctypes.addressof(n) for n in nets]), calls[0][3].shape[0],
        ptrs([as_int(v[0]) for v in svs]), (ctypes.c_int * k)(*[v[1] for v in svs]),
        ptrs([as_int(v[0]) for v in avs]), (ctypes.c_int * k)(*[v[1] for v in avs]),
        ptrs([as_int(_p(c[3])) for c in calls]), ptrs([as_int(_p(c[4], allow_none=True)) for c in calls]),
        ptrs([as_int(_p(c[5], allow_none=True)) for c in calls]), _stream()), "rpo_mlp_forward_multi")


def mlp_split_supported(desc):
    net = desc.net_struct()
    return bool(_lib.load().rpo_mlp_split_supported(ctypes.byref(net)))


_FRONT_OK = {}


def front_launch_ok(batch, twin, extra_planes=0):
    """Whether rpo_split_critic_front may be used for this batch: its workgroups hand data over through ONE XCD's L2, which
    needs every workgroup of a row tile -- in every plane of the launch -- on the same XCD.  That is a property of the
    dispatcher (workgroups go to the XCDs round-robin in block order), checked once per process and shape with a probe
    launch of the same grid (rpo_xcc_probe)."""
    key = (int(batch), bool(twin), int(extra_planes))
    if key not in _FRONT_OK:
        # (+ the policy plane / the first riding plane; extra_planes: the projection plane of the SpringPendulum front)
        T, planes = (batch + 15) // 16, 1 + 3 * (2 if twin else 1) + 1 + extra_planes
        out = torch.full((planes, T, 8), -1, dtype=torch.int32, device="cuda")
        check(_lib.load().rpo_xcc_probe(8, T, planes, 256, _p(out, torch.int32), _stream()), "rpo_xcc_probe")
        ids = out.cpu().view(planes, 8 * T)                       # linear block id inside a plane: x + 8 y
        tile = torch.arange(8 * T) % T                            # ns_block: the row tile is the fastest index
        ok = bool((ids >= 0).all())
        for t in range(T):
            ok = ok and len(torch.unique(ids[:, tile == t])) == 1
        _FRONT_OK[key] = ok
    return _FRONT_OK[key]


def mlp_forward_split(calls):
    """Column-split hidden layers of 1..4 networks in one launch: calls = [(desc, s, a, part [8, n, 2], x0_save,
    h1_save), ...]; the head outputs are completed by ``mlp_split_head`` or by the consumer kernel's prologue."""
    k = len(calls)
    nets = [c[0].net_struct() for c in calls]
    svs = [_row_view(c[1], c[0].S) for c in calls]
    avs = [(None, 0) if c[0].A == 0 else _row_view(c[2], c[0].A) for c in calls]
    ptrs = lambda vals: (ctypes.c_void_p * k)(*vals)                                   # noqa: E731
    as_int = lambda v: None if v is None else (v.value if isinstance(v, ctypes.c_void_p) else v)   # noqa: E731
    n = calls[0][3].shape[1]
    check(_lib.load().rpo_mlp_forward_split(
        k, ptrs([ctypes.addressof(x) for x in nets]), n,
        ptrs([as_int(v[0]) for v in svs]), (ctypes.c_int * k)(*[v[1] for v in svs]),
        ptrs([as_int(v[0]) for v in avs]), (ctypes.c_int * k)(*[v[1] for v in avs]),
        ptrs([as_int(_p(c[3])) for c in calls]), ptrs([as_int(_p(c[4], allow_none=True)) for c in calls]),
        ptrs([as_int(_p(c[5], allow_none=True)) for c in calls]), _stream()), "rpo_mlp_forward_split")


def mlp_split_head(desc, part, out, out_mode=0, scale=1.0, base=0.0):
    net = desc.net_struct()
    check(_lib.load().rpo_mlp_split_head(ctypes.byref(net), part.shape[1], _p(part), _p(out), out_mode, scale, base,
                                         _stream()), "rpo_mlp_split_head")


def mlp_backward(desc, s, a, x0, h1, dout, dh, dx0, da=None, param_grads=True, first_layer_state_only=False,
                 gradmax=None, td=None):
    """td (a ``Td``): dout is not read; the rows pass computes it (TD target + Huber) into td.dq_out."""
    sp, ss = _row_view(s, desc.S)
    ap, as_ = (None, 0) if desc.A == 0 else _row_view(a, desc.A)
    net = desc.net_struct()
    grad = desc.grad_struct() if param_grads else None
    n = dout.shape[0] if td is None else td.dq_out.shape[0]
    tds = None if td is None else td.struct()
    check(_lib.load().rpo_mlp_backward(ctypes.byref(net), None if grad is None else ctypes.byref(grad), n,
                                       sp, ss, ap, as_, _p(x0), _p(h1), _p(dout, allow_none=td is not None), _p(dh), _p(dx0),
                                       _p(da, allow_none=True), int(param_grads), int(first_layer_state_only),
                                       _p(gradmax, allow_none=True), None if tds is None else ctypes.byref(tds), _stream()),
          "rpo_mlp_backward")


def tanh_box_bwd(dap, ap_det, noise, eps_start, eps_end, eps_decay, ctrl, lo, hi, scale, base, dout):
    check(_lib.load().rpo_tanh_box_bwd(dap.numel(), _p(dap), _p(ap_det), _p(noise, allow_none=True), eps_start, eps_end,
                                       eps_decay, _p(ctrl, torch.int64, allow_none=True), lo, hi, scale, base, _p(dout),
                                       _stream()), "rpo_tanh_box_bwd")


def gauss_head(raw, eps, scale, base, lo, hi, deterministic, ap_out, logp_out=None):
    check(_lib.load().rpo_gauss_head(raw.shape[0], _p(raw), _p(eps), scale, base, lo, hi, int(deterministic),
                                     _p(ap_out), _p(logp_out, allow_none=True), _stream()), "rpo_gauss_head")


def gauss_head_bwd(raw, eps, dap, dlogp, scale, base, lo, hi, draw):
    check(_lib.load().rpo_gauss_head_bwd(raw.shape[0], _p(raw), _p(eps), _p(dap), dlogp, scale, base, lo, hi, _p(draw),
                                         _stream()), "rpo_gauss_head_bwd")


def mlp_backward_pair(desc1, desc2, s, a, x0_1, h1_1, dout_1, dh_1, dx0_1, da_1, x0_2, h1_2, dout_2, dh_2, dx0_2, da_2,
                      param_grads=True, first_layer_state_only=False, gradmax=None, td1=None, td2=None):
    sp, ss = _row_view(s, desc1.S)
    ap, as_ = (None, 0) if desc1.A == 0 else _row_view(a, desc1.A)
    n1, n2 = desc1.net_struct(), desc2.net_struct()
    g1 = desc1.grad_struct() if param_grads else None
    g2 = desc2.grad_struct() if param_grads else None
    n = dout_1.shape[0] if td1 is None else td1.dq_out.shape[0]
    t1 = None if td1 is None else td1.struct()
    t2 = None if td2 is None else td2.struct()
    check(_lib.load().rpo_mlp_backward_pair(
        ctypes.byref(n1), None if g1 is None else ctypes.byref(g1), ctypes.byref(n2), None if g2 is None else ctypes.byref(g2),
        n, sp, ss, ap, as_, _p(x0_1), _p(h1_1), _p(dout_1, allow_none=td1 is not None), _p(dh_1), _p(dx0_1),
        _p(da_1, allow_none=True), _p(x0_2), _p(h1_2), _p(dout_2, allow_none=td2 is not None), _p(dh_2), _p(dx0_2),
        _p(da_2, allow_none=True), int(param_grads), int(first_layer_state_only), _p(gradmax, allow_none=True),
        None if t1 is None else ctypes.byref(t1), None if t2 is None else ctypes.byref(t2), _stream()),
        "rpo_mlp_backward_pair")


def sac_actor_forward(env_kernels, actor, critic1, critic2, scale, base, box_lo, box_hi, alpha, batch, noise_in, seed,
                      noise_id_base, noise_salt, ctrl, nu, raw, noise_out, logp, actions, dq1, dq2, g_act, partial_out, bufs):
    """bufs = (actor_x0, actor_h1, critic1_x0, critic1_h1, critic2_x0, critic2_h1)."""
    is_cart = isinstance(env_kernels, CartSafeKernels)
    nets = [d.net_struct() for d in (actor, critic1, critic2)]
    check(_lib.load().rpo_sac_actor_forward(
        0 if is_cart else 1, *[ctypes.byref(n) for n in nets], scale, base, box_lo, box_hi, alpha, _p(batch), batch.shape[0],
        _p(noise_in, allow_none=True), seed, noise_id_base, noise_salt, _p(ctrl, torch.int64), _p(nu),
        env_kernels._cptr if is_cart else None, env_kernels.partial if is_cart else 0, _p(raw), _p(noise_out), _p(logp),
        _p(actions), _p(dq1), _p(dq2), _p(g_act), _p(partial_out), *[_p(b) for b in bufs], _stream()),
        "rpo_sac_actor_forward")


def sac_actor_backward(env_kernels, actor, critic1, critic2, shared_embedding, batch, actions, g_act, raw, noise, logp, dq1,
                       dq2, dlogp, box_lo, box_hi, scale, base, saved, scratch, da1, da2, dout, partial_in, lag_out, nu_grad,
                       gradmax):
    """saved = the six pre-activation buffers of the forward; scratch = (actor_dh, actor_dx0, c1_dh, c1_dx0, c2_dh, c2_dx0)."""
    is_cart = isinstance(env_kernels, CartSafeKernels)
    an, c1, c2, ag = actor.net_struct(), critic1.net_struct(), critic2.net_struct(), actor.grad_struct()
    check(_lib.load().rpo_sac_actor_backward(
        0 if is_cart else 1, ctypes.byref(an), ctypes.byref(ag), ctypes.byref(c1), ctypes.byref(c2), int(shared_embedding),
        _p(batch), batch.shape[0], _p(actions), _p(g_act), _p(raw), _p(noise), _p(logp), _p(dq1), _p(dq2), dlogp, box_lo, box_hi, scale,
        base, env_kernels._cptr if is_cart else None, env_kernels.partial if is_cart else 0, *[_p(b) for b in saved],
        *[_p(b) for b in scratch], _p(da1), _p(da2), _p(dout), _p(partial_in), _p(lag_out), _p(nu_grad),
        _p(gradmax, allow_none=True), _stream()), "rpo_sac_actor_backward")


def ddpg_actor_forward(env_kernels, actor, critic, scale, base, box_lo, box_hi, eps_start, eps_end, eps_decay, batch, noise_in,
                       seed, noise_id_base, noise_salt, ctrl, nu, ap_det, noise_out, actions, q_out, dq_out, g_act,
                       partial_out, actor_x0, actor_h1, critic_x0, critic_h1):
    is_cart = isinstance(env_kernels, CartSafeKernels)
    an, cn = actor.net_struct(), critic.net_struct()
    check(_lib.load().rpo_ddpg_actor_forward(
        0 if is_cart else 1, ctypes.byref(an), ctypes.byref(cn), scale, base, box_lo, box_hi, eps_start, eps_end, eps_decay,
        _p(batch), batch.shape[0], _p(noise_in, allow_none=True), seed, noise_id_base, noise_salt, _p(ctrl, torch.int64), _p(nu),
        env_kernels._cptr if is_cart else None, env_kernels.partial if is_cart else 0, _p(ap_det), _p(noise_out), _p(actions),
        _p(q_out), _p(dq_out), _p(g_act), _p(partial_out), _p(actor_x0), _p(actor_h1), _p(critic_x0), _p(critic_h1), _stream()),
        "rpo_ddpg_actor_forward")


def ddpg_actor_backward(env_kernels, actor, critic, shared_embedding, batch, actions, g_act, ap_det, noise, dq, eps_start,
                        eps_end, eps_decay, box_lo, box_hi, scale, base, ctrl, actor_x0, actor_h1, critic_x0, critic_h1,
                        actor_dh, actor_dx0, critic_dh, critic_dx0, da, dout, partial_in, lag_out, nu_grad, gradmax):
    is_cart = isinstance(env_kernels, CartSafeKernels)
    an, cn, ag = actor.net_struct(), critic.net_struct(), actor.grad_struct()
    check(_lib.load().rpo_ddpg_actor_backward(
        0 if is_cart else 1, ctypes.byref(an), ctypes.byref(ag), ctypes.byref(cn), int(shared_embedding), _p(batch),
        batch.shape[0], _p(actions), _p(g_act), _p(ap_det), _p(noise), _p(dq), eps_start, eps_end, eps_decay, box_lo, box_hi,
        scale, base, _p(ctrl, torch.int64), env_kernels._cptr if is_cart else None, env_kernels.partial if is_cart else 0,
        _p(actor_x0), _p(actor_h1), _p(critic_x0), _p(critic_h1), _p(actor_dh), _p(actor_dx0), _p(critic_dh), _p(critic_dx0),
        _p(da), _p(dout), _p(partial_in), _p(lag_out), _p(nu_grad), _p(gradmax, allow_none=True), _stream()),
        "rpo_ddpg_actor_backward")


# =================================================================================================== column-split update

class _SplitUpdateStruct(ctypes.Structure):
    """rpo_split_update of include/rpo_hip.h (same field order)."""
    _P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    _fields_ = (
        [(n, ctypes.c_void_p) for n in ("actor", "actor_target", "critic1", "critic2", "critic_target1", "critic_target2",
                                        "critic1_grad", "critic2_grad", "actor_grad")] +
        [("env", ctypes.c_int), ("twin", ctypes.c_int), ("batch", ctypes.c_int),
         ("rows", ctypes.c_void_p), ("cap_steps", ctypes.c_longlong), ("n_envs", ctypes.c_int),
         ("batch_out", ctypes.c_void_p), ("idx_out", ctypes.c_void_p), ("idx_in", ctypes.c_void_p),
         ("sample_seed", ctypes.c_ulonglong), ("sample_salt", ctypes.c_uint),
         ("eps_in", ctypes.c_void_p), ("noise_seed", ctypes.c_ulonglong), ("noise_id_base", ctypes.c_uint),
         ("noise_salt", ctypes.c_uint), ("ctrl", ctypes.c_void_p),
         ("scale", ctypes.c_float), ("base", ctypes.c_float), ("box_lo", ctypes.c_float), ("box_hi", ctypes.c_float),
         ("max_steps", ctypes.c_int), ("corr_lr", ctypes.c_float), ("corr_eps", ctypes.c_float),
         ("corr_momentum", ctypes.c_float), ("consts_host", ctypes.c_void_p), ("partial", ctypes.c_int),
         ("alpha", ctypes.c_float), ("gamma", ctypes.c_float),
         ("eps_start", ctypes.c_float), ("eps_end", ctypes.c_float), ("eps_decay", ctypes.c_float)] +
        [(n, ctypes.c_void_p) for n in ("part_pi", "part_q1", "part_q2", "part_qn1", "part_qn2", "x0_1", "h1_1", "x0_2",
                                        "h1_2", "x0_a", "h1_a", "logp", "next_actions", "proj_iters", "dq1", "dq2",
                                        "loss_partial", "dx0_1", "dx0_2", "dx0_a", "gradmax", "nu", "nu_grad", "ap_det",
                                        "noise_out", "raw", "actions", "g_act", "lag_partial", "lag_out", "da_part",
                                        "dout")] +
        [("shared_embedding", ctypes.c_int), ("rollout_ctrl", ctypes.c_void_p), ("rollout_stats", ctypes.c_void_p),
         ("rollout_stats_cap", ctypes.c_int), ("prep_step", ctypes.c_void_p), ("prep_beta1", ctypes.c_float),
         ("prep_beta2", ctypes.c_float), ("clock_out", ctypes.c_void_p), ("gradmax_reset", ctypes.c_void_p),
         ("prep2_step", ctypes.c_void_p * 3), ("prep2_beta1", ctypes.c_float * 3), ("prep2_beta2", ctypes.c_float * 3),
         ("gradmax_reset2", ctypes.c_void_p), ("updates_out", ctypes.c_void_p), ("part_pol", ctypes.c_void_p),
         ("tile_sync", ctypes.c_void_p), ("proj_ws", ctypes.c_void_p), ("proj_store_mode", ctypes.c_int), ("debug", ctypes.c_int)])


class _RolloutRiderStruct(ctypes.Structure):
    """rpo_rollout_rider of include/rpo_hip.h (same field order)."""
    _fields_ = [("n_envs", ctypes.c_int), ("gauss", ctypes.c_int), ("scale", ctypes.c_float), ("base", ctypes.c_float),
                ("state", ctypes.c_void_p), ("obs", ctypes.c_void_p), ("action", ctypes.c_void_p),
                ("ep_len", ctypes.c_void_p), ("ep_ret", ctypes.c_void_p), ("ep_count", ctypes.c_void_p),
                ("rows", ctypes.c_void_p), ("cap_steps", ctypes.c_longlong), ("stats", ctypes.c_void_p),
                ("stats_cap", ctypes.c_int), ("noise_mode", ctypes.c_int), ("ctrl", ctypes.c_void_p),
                ("eps_start", ctypes.c_float), ("eps_end", ctypes.c_float), ("eps_decay", ctypes.c_float),
                ("box_lo", ctypes.c_float), ("box_hi", ctypes.c_float), ("max_steps", ctypes.c_int),
                ("corr_lr", ctypes.c_float), ("corr_eps", ctypes.c_float), ("corr_momentum", ctypes.c_float),
                ("max_episode_steps", ctypes.c_int), ("auto_reset", ctypes.c_int), ("viol_thresh", ctypes.c_float),
                ("env_id_base", ctypes.c_uint), ("seed", ctypes.c_ulonglong), ("part", ctypes.c_void_p),
                ("lane_begin", ctypes.c_int), ("lane_end", ctypes.c_int), ("defer_clock", ctypes.c_int)]


class RolloutRider(object):
    """Arguments of the rollout stages that ride on the critic update's launches (rpo_split_critic_fwd_a_ride /
    _fwd_b_ride: the actor forward of a lane range; _bwd_b_ride: the step): the same values `rollout()` of the env kernels
    takes, plus the partials workspace and the lane range."""

    _DTYPES = dict(ep_len=torch.int32, ep_count=torch.int32, ctrl=torch.int64)

    def __init__(self, ring_floats=None, **fields):
        """``ring_floats``: the env kernel set's ring stride; `rows=` tensors of another width are refused (`_ring`)."""
        self.st, self._held, self.ring_floats = _RolloutRiderStruct(), {}, ring_floats
        self.set(**fields)

    def set(self, **fields):
        for k, v in fields.items():
            if isinstance(v, torch.Tensor):
                if k == "rows" and self.ring_floats is not None:
                    _ring(v, self.ring_floats)
                self._held[k] = v
                setattr(self.st, k, _p(v, self._DTYPES.get(k, torch.float32)).value)
            else:
                self._held.pop(k, None)
                setattr(self.st, k, v)


class SplitUpdate(object):
    """Arguments of the column-split update stages (rpo_split_*), built once per trainer: every pointer refers to a static
    device buffer, so the struct is reused for every launch (and every hipGraph capture)."""

    STAGES = ("critic_fwd_a", "critic_fwd_b", "critic_front", "critic_front_pol", "critic_mid", "critic_mid_pol", "critic_pfront", "critic_pfront_pol", "critic_fwd_b_pol", "pend_head_project", "critic_bwd_a", "critic_bwd_b", "policy_a", "policy_b",
              "policy_c", "policy_d", "policy_e", "policy_front", "policy_front_bc")

    def __init__(self, env_kernels, descs, twin, batch, fields):
        """descs: name -> MlpDesc for actor, actor_target (RPODDPG), critic1, critic2, critic_target1, critic_target2;
        fields: remaining struct fields (tensors become device pointers, Python scalars are copied)."""
        self._keep, self._held, self.ring_floats = [], {}, env_kernels.ring_floats
        st = self.st = _SplitUpdateStruct()
        for name in ("actor", "actor_target", "critic1", "critic2", "critic_target1", "critic_target2"):
            d = descs.get(name)
            if d is not None:
                net = d.net_struct()
                self._keep.append(net)
                setattr(st, name, ctypes.addressof(net))
        for name, key in (("critic1_grad", "critic1"), ("critic2_grad", "critic2"), ("actor_grad", "actor")):
            d = descs.get(key)
            if d is not None and d.tensors["W0"].grad is not None:
                g = d.grad_struct()
                self._keep.append(g)
                setattr(st, name, ctypes.addressof(g))
        is_cart = isinstance(env_kernels, CartSafeKernels)
        st.env, st.twin, st.batch = (0 if is_cart else 1), int(bool(twin)), int(batch)
        if is_cart:
            self._keep.append(env_kernels.consts)
            st.consts_host, st.partial = env_kernels.consts.ctypes.data, env_kernels.partial
        self.set(**fields)

    def set(self, **fields):
        for k, v in fields.items():
            if isinstance(v, torch.Tensor):
                dt = torch.int64 if k in ("idx_out", "idx_in", "ctrl", "rollout_ctrl", "clock_out", "updates_out") else \
                    (torch.int32 if k in ("proj_iters", "prep_step", "tile_sync") else torch.float32)
                if k == "proj_ws":
                    dt = torch.int64
                if k == "rows":                                     # the fused sampling reads rows at the compiled ring stride
                    _ring(v, self.ring_floats)
                self._held[k] = v                                   # keeps the buffer alive while the struct points at it
                setattr(self.st, k, _p(v, dt).value)
            else:
                self._held.pop(k, None)
                setattr(self.st, k, v)

    def set_prep2(self, optims):
        """pol_e's bookkeeping list: up to three (step_dev, beta1, beta2) of the slices of the Adam launch behind it."""
        optims = list(optims)
        self._held["prep2"] = [o[0] for o in optims]
        for j in range(3):
            if j < len(optims):
                self.st.prep2_step[j] = _p(optims[j][0], torch.int32).value
                self.st.prep2_beta1[j], self.st.prep2_beta2[j] = optims[j][1], optims[j][2]
            else:
                self.st.prep2_step[j] = None

    def run(self, stage, rider=None):
        """``rider``: a RolloutRider whose share of the next vector step rides on this launch (critic_fwd_a / critic_fwd_b:
        the actor forward of lanes [lane_begin, lane_end), critic_bwd_b: explore / project / step / scatter)."""
        if rider is None:
            check(getattr(_lib.load(), "rpo_split_" + stage)(ctypes.byref(self.st), _stream()), "rpo_split_" + stage)
        else:
            name = "rpo_split_" + stage + "_ride"
            check(getattr(_lib.load(), name)(ctypes.byref(self.st), ctypes.byref(rider.st), _stream()), name)
