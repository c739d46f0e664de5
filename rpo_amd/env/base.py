"""Common host-side machinery of the hard-constraint envs (the reference's "env is also the constraint oracle"
design): Gym 0.19 API for one env + the batched torch constraint API, both backed by HIP kernels.
"""
import copy

import numpy as np
import torch

from .. import gym_shim, ops as hip_ops
from .vec import VecEnv

gym = gym_shim.install()


def default_device():
    # same rule as the reference (cartpole.py:122, rpo_ddpg.py:22)
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class _CompletePartialFn(torch.autograd.Function):
    """Equation solver (complete_partial) with its analytic backward -- both HIP."""

    @staticmethod
    def forward(ctx, env, obs, action_partial):
        k = env.kernels
        n = action_partial.shape[0]
        action = torch.empty(n, k.action_dim, device=action_partial.device)
        lo, hi = env.partial_box
        k.act_project(obs, action_partial.reshape(-1).contiguous(), None, action, None, hip_ops.NOISE_NONE, 0.0, 0.0, 0.0,
                      lo, hi, 0, 0.0, 1e-5, 0.0)
        ctx.env, ctx.obs, ctx.action = env, obs, action
        return action

    @staticmethod
    def backward(ctx, grad_action):
        k = ctx.env.kernels
        n = grad_action.shape[0]
        gap = torch.empty(n * k.partial_dim, device=grad_action.device)
        k.complete_bwd(ctx.obs, grad_action.contiguous(), gap, action=ctx.action)
        return None, None, gap.view(n, k.partial_dim)


class _ResidFn(torch.autograd.Function):
    """(eq_resid, ineq_resid) of a batch of actions: HIP forward, analytic backward w.r.t. the action."""

    @staticmethod
    def forward(ctx, env, obs, action):
        k = env.kernels
        n = action.shape[0]
        eq = torch.empty(n, k.eq_num, device=action.device)
        ineq = torch.empty(n, k.ineq_num, device=action.device)
        k.resid(obs, action.contiguous(), eq, ineq)
        ctx.env, ctx.obs = env, obs
        ctx.save_for_backward(action)
        return eq, ineq

    @staticmethod
    def backward(ctx, grad_eq, grad_ineq):
        (action,) = ctx.saved_tensors
        return None, None, ctx.env._resid_backward(ctx.obs, action, grad_eq, grad_ineq)


class HardConstraintEnv(gym.Env):
    """Shared implementation; subclasses provide the spaces, ``_make_kernels`` and ``_resid_backward``."""

    volatile = False      # state-independent action box (cartpole.py:147, pendulum.py:67)
    update = None

    def __init__(self, backend=None, device=None):
        self.device = torch.device(device) if device is not None else default_device()
        self._backend = backend if backend is not None else hip_ops
        self._kernels = None
        self._vec = None
        self._np_seed = None
        self.holding_eq = False
        self.holding_ineq = False

    # ------------------------------------------------------------------------------------------ plumbing
    @property
    def kernels(self):
        if self._kernels is None:
            self._kernels = self._make_kernels()
        return self._kernels

    @property
    def partial_box(self):
        lo, hi = self.box_constraint_partial
        return float(lo[0]), float(hi[0])

    def make_vec(self, n_envs, seed=0, env_id_base=0, max_episode_steps=None, device=None, **kw):
        return VecEnv(self.kernels, n_envs, device or self.device, seed=seed, env_id_base=env_id_base,
                      max_episode_steps=max_episode_steps, **kw)

    def __deepcopy__(self, memo):
        # RPODDPG keeps `env_eval = copy.deepcopy(env)` (rpo_ddpg.py:61): device state and ctypes handles are rebuilt
        new = copy.copy(self)
        new._kernels = None
        new._vec = None
        memo[id(self)] = new
        return new

    def _t(self, x):
        return torch.as_tensor(x, dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------------------------------ gym API (1 env)
    def seed(self, seed=None):
        self.np_random, seed = gym.utils.seeding.np_random(seed)
        self._np_seed = seed
        return [seed]

    def _single(self):
        if self._vec is None:
            self._vec = self.make_vec(1, seed=0 if self._np_seed is None else int(self._np_seed), stats_cap=2)
            self._row = torch.zeros(1, self.kernels.ring_floats, device=self.device)
        return self._vec

    def reset(self):
        """cartpole.py:231-239 / pendulum.py:130-136: uniform initial state from the env's own numpy generator."""
        vec = self._single()
        vec.set_internal(self._draw_initial()[None, :])
        self.state = vec.internal[0].cpu().numpy().astype(np.float64)
        self.steps_beyond_done = None
        return vec.obs[0].cpu().numpy().astype(np.float64)

    def step(self, action):
        """One env step through the same fused HIP kernel the vectorised rollout uses (n = 1, no auto-reset).
        Returns (obs, reward, done, {'ineq_viol', 'eq_viol'}) like cartpole.py:229 / pendulum.py:128."""
        vec = self._single()
        a_np = np.asarray(action, dtype=np.float32).reshape(1, -1)
        # cartpole.py:170-174 / pendulum.py:85-89: the CLIPPED action must lie in the action space -- it does unless it is NaN
        assert not np.isnan(a_np).any(), "%r (%s) invalid" % (action, type(action))
        a = self._t(a_np)
        vec.ctrl.zero_()
        vec.ep_len.zero_()            # the TimeLimit lives in the gym wrapper, not here
        self.kernels.step(vec.internal, vec.obs, a, vec.ep_len, vec.ep_ret, vec.ep_count, self._row, 1, None, vec.ctrl,
                          2 ** 31 - 1, False, vec.viol_thresh, vec.seed, 0)
        row = self._row[0].cpu().numpy()
        c = self.kernels.cols
        self.state = vec.internal[0].cpu().numpy().astype(np.float64)
        info = {"ineq_viol": row[c["ineq_viol"][0]:c["ineq_viol"][1]].copy(),
                "eq_viol": row[c["eq_viol"][0]:c["eq_viol"][1]].copy()}
        return (row[c["next_state"][0]:c["next_state"][1]].astype(np.float64), float(row[c["reward"][0]]),
                bool(row[c["done"][0]] > 0.5), info)

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope of the MI355X hot path (SURVEY.md §2 row 17)")

    def close(self):
        self._vec = None

    # ------------------------------------------------------------------------------------------ constraint API
    def complete_partial(self, state, action_partial):
        """Equation solver: basic actions -> full action satisfying the equalities (differentiable)."""
        return _CompletePartialFn.apply(self, self._t(state), self._t(action_partial))

    def _resid(self, state, action):
        return _ResidFn.apply(self, self._t(state), self._t(action))

    def eq_resid(self, state, action):
        return self._resid(state, action)[0]

    def ineq_resid(self, state, action):
        return self._resid(state, action)[1]

    def eq_dist(self, state, action):
        return torch.abs(self.eq_resid(state, action))

    def ineq_dist(self, state, action):
        return torch.clamp(self.ineq_resid(state, action), 0)

    def ineq_partial_grad(self, state, action, eps=0):
        """GRG direction; ``eps`` is accepted and ignored exactly like the reference (cartpole.py:402)."""
        action = self._t(action).contiguous()
        out = torch.empty_like(action)
        self.kernels.ineq_partial_grad(self._t(state), action, out)
        return out

    def project(self, state, action_partial, max_steps, lr, corr_eps=1e-5, momentum=0.0, return_iters=False,
                batch_reference=False, **act_kw):
        """complete_partial + grad_steps in one launch (the fused form the trainers use).

        ``batch_reference=True`` asks for the reference's literal behaviour on a batch (one stop test for the whole
        batch, and for SpringPendulum the sample-coupled step of pendulum.py:337-339) where the env has
        such a kernel; otherwise every row is projected independently, as in the reference's B = 1 rollouts."""
        ap = self._t(action_partial).reshape(-1, self.kernels.partial_dim).contiguous()
        n = ap.shape[0]
        action = torch.empty(n, self.kernels.action_dim, device=self.device)
        iters = torch.empty(n, dtype=torch.int32, device=self.device) if return_iters else None
        if batch_reference and n > 1 and hasattr(self.kernels, "project_batchref"):
            it1 = torch.empty(1, dtype=torch.int32, device=self.device) if return_iters else None
            self.kernels.project_batchref(self._t(state), ap, action, it1, int(max_steps), float(lr), float(corr_eps),
                                          float(momentum))
            if return_iters:
                iters[:] = it1
        else:
            lo, hi = self.partial_box
            self.kernels.act_project(self._t(state), ap, None, action, iters, hip_ops.NOISE_NONE, 0.0, 0.0, 0.0, lo, hi,
                                     int(max_steps), float(lr), float(corr_eps), float(momentum), **act_kw)
        return (action, iters) if return_iters else action

    def hold_eq(self):
        self.holding_eq = True

    def release_eq(self):
        self.holding_eq = False

    def hold_ineq(self):
        self.holding_ineq = True

    def release_ineq(self):
        self.holding_ineq = False

    @property
    def box_constraint(self):
        return self.action_space.low, self.action_space.high

    @property
    def box_constraint_partial(self):
        return self.action_space.low[self.partial_actions], self.action_space.high[self.partial_actions]
