"""CartSafe-v0: cart-pole with friction driven by two angled forces, 1 linear equality and 6 linear inequalities on
the action (reference: rpo/env/classic_control/cartpole.py).  Dynamics, violation bookkeeping and all constraint
arithmetic run in the HIP kernels of rpo_amd/csrc/cartsafe.hip; this class holds the constants and the Python surface.

State (= observation): (x, x_dot, xacc, theta, theta_dot, thetaacc).  Action: magnitudes of two forces at angles
delta = (pi/3, -pi/6).  Equality: a . sin(delta) = 0 (no net vertical force).  Inequalities: |a . cos(delta)| <= 8,
|a_i| <= 10.
"""
import math

import numpy as np
import torch

from ..base import HardConstraintEnv, gym

spaces = gym.spaces


class CartSafeEnv(HardConstraintEnv):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 50}

    def __init__(self, partial_actions=None, backend=None, device=None):
        super().__init__(backend, device)
        self.delta = np.array([np.pi / 3, -np.pi / 6])                       # cartpole.py:85
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        fmax = np.finfo(np.float32).max
        high = np.array([self.x_threshold * 2, fmax, fmax, self.theta_threshold_radians * 2, fmax, fmax], dtype=np.float32)
        self.observation_space = spaces.Box(-high, high, dtype=np.float32)
        self.action_space = spaces.Box(np.array([-10, -10], dtype=np.float32), np.array([10, 10], dtype=np.float32))
        self.state_dim, self.action_dim, self.eq_num, self.ineq_num = 6, 2, 1, 6
        self.seed()
        self.state = None
        self.steps_beyond_done = None
        if partial_actions is None:
            # the reference draws the basic action from the GLOBAL numpy RNG (cartpole.py:117); the same draw is made
            # here so that a script seeded with np.random.seed(123) picks the same index ([1]) and leaves the global
            # stream in the same state
            partial_actions = np.random.choice(self.action_dim, self.action_dim - self.eq_num, replace=False)
        self.partial_actions = np.asarray(partial_actions).reshape(-1)
        self.other_actions = np.setdiff1d(np.arange(self.action_dim), self.partial_actions)
        self._build_constants()

    def _build_constants(self):
        """cartpole.py:124-136 on the host in float32 with torch's own sin/cos, as the reference's CPU path does."""
        p, o = self.partial_actions, self.other_actions
        d32 = torch.tensor(self.delta, dtype=torch.float32)
        self.diff_eq = torch.sin(d32).view(1, 2)
        self.diff_eq_partial = self.diff_eq[:, p]
        self.diff_eq_other_inv = torch.inverse(self.diff_eq[:, o])
        self.diff_eq_bias = torch.zeros(1, 1)
        cosd = torch.cos(d32)
        G = torch.zeros(6, 2)
        G[0], G[1] = cosd, -cosd
        G[2, 0], G[3, 0], G[4, 1], G[5, 1] = 1.0, -1.0, 1.0, -1.0
        self.diff_ineq = G
        self.diff_ineq_bias = torch.tensor([8, 8, 10, 10, 10, 10], dtype=torch.float32)
        G_r = G[:, p] - G[:, o] @ (self.diff_eq_other_inv @ self.diff_eq_partial)                 # :397-398
        d_r = self.diff_ineq_bias - (self.diff_eq_bias @ self.diff_eq_other_inv.T) @ G[:, o].T    # :399-400
        self._table = torch.cat([self.diff_eq.reshape(-1), self.diff_eq_partial.reshape(-1),
                                 self.diff_eq_other_inv.reshape(-1), self.diff_eq_bias.reshape(-1), G.reshape(-1),
                                 self.diff_ineq_bias, G_r.reshape(-1), d_r.reshape(-1)]).numpy().astype(np.float32)
        for name in ("diff_eq", "diff_eq_partial", "diff_eq_other_inv", "diff_eq_bias", "diff_ineq", "diff_ineq_bias"):
            setattr(self, name + "_np", getattr(self, name).numpy())
            setattr(self, name, getattr(self, name).to(self.device))

    def _make_kernels(self):
        return self._backend.CartSafeKernels(self._table, int(self.partial_actions[0]))

    def _draw_initial(self):
        return self.np_random.uniform(low=-0.05, high=0.05, size=(6,))                            # cartpole.py:233

    def _resid_backward(self, obs, action, grad_eq, grad_ineq):
        # eq = b - a C^T ; ineq = a G^T - d
        return grad_ineq @ self.diff_ineq - grad_eq @ self.diff_eq

    def eq_grad(self, state, action):
        action = self._t(action)
        return 2 * (action @ self.diff_eq.T - self.diff_eq_bias) @ self.diff_eq                   # cartpole.py:389-390

    def ineq_grad(self, state, action):
        return 2 * self.ineq_dist(state, action) @ self.diff_ineq                                 # cartpole.py:392-394

    # numpy variants used by the reference's env.step info (cartpole.py:410-422)
    def ineq_resid_np(self, state, action):
        return action @ self.diff_ineq_np.T - self.diff_ineq_bias_np

    def ineq_dist_np(self, state, action):
        return np.clip(self.ineq_resid_np(state, action), 0, None)

    def eq_resid_np(self, state, action):
        return self.diff_eq_bias_np - action @ self.diff_eq_np.T

    def eq_dist_np(self, state, action):
        return np.abs(self.eq_resid_np(state, action))
