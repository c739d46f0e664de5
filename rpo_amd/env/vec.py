"""Device-resident state of N vectorised env instances (the rollout side of the hot path).

One ``VecEnv`` per (rank, purpose): the training rollout owns ``n_envs`` lanes plus the replay ring it scatters into;
evaluation uses a second, small one.  All tensors live in HBM; nothing here synchronises with the host.
"""
import torch

from .. import ops as hip_ops


class VecEnv(object):

    def __init__(self, kernels, n_envs, device, seed=0, env_id_base=0, max_episode_steps=None, viol_thresh=1e-3,
                 stats_cap=4096, ctrl=None):
        self.k = kernels
        self.n = int(n_envs)
        self.device = device
        self.seed = int(seed)
        self.env_id_base = int(env_id_base)
        # no TimeLimit wrapper -> effectively unbounded (int32 max) episode length
        self.max_episode_steps = int(max_episode_steps) if max_episode_steps else 2 ** 31 - 1
        self.viol_thresh = float(viol_thresh)
        self.internal = torch.zeros(self.n, kernels.internal_dim, device=device)
        # CartSafe observes its internal state directly; SpringPendulum has a separate observation
        self.obs = self.internal if kernels.obs_dim == kernels.internal_dim else \
            torch.zeros(self.n, kernels.obs_dim, device=device)
        self.action = torch.zeros(self.n, kernels.action_dim, device=device)
        self.ep_len = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.ep_ret = torch.zeros(self.n, device=device)
        self.ep_count = torch.zeros(self.n, dtype=torch.int32, device=device)
        self.stats = hip_ops.new_stats(stats_cap, device)
        self.ctrl = ctrl if ctrl is not None else torch.zeros(hip_ops.CTRL_LEN, dtype=torch.int64, device=device)
        self.steps_host = 0

    def reset(self):
        self.k.reset(self.internal, self.obs, self.ep_len, self.ep_ret, self.ep_count, self.seed, self.env_id_base)
        return self.obs

    def set_internal(self, internal):
        """Inject explicit initial states (parity tests; SURVEY 8c 'parity tests inject initial states')."""
        self.internal.copy_(torch.as_tensor(internal, dtype=torch.float32, device=self.device))
        self.ep_len.zero_()
        self.ep_ret.zero_()
        if self.obs is not self.internal:
            th = self.internal[:, 0]
            self.obs.copy_(torch.stack([torch.cos(th), torch.sin(th), self.internal[:, 1], self.internal[:, 2],
                                        self.internal[:, 3]], dim=1))

    def step(self, action, rows=None, cap_steps=1, auto_reset=True):
        """One fused vector step; scatters the transitions into ``rows`` (the replay ring) when given."""
        self.k.step(self.internal, self.obs, action, self.ep_len, self.ep_ret, self.ep_count, rows, cap_steps,
                    self.stats, self.ctrl, self.max_episode_steps, auto_reset, self.viol_thresh, self.seed,
                    self.env_id_base)
        self.steps_host += 1
        return self.obs
