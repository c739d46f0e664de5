"""The reference's comparison baselines: plain Lagrangian DDPG / SAC on the FULL action (rpo/algo/ddpg_lag.py,
sac_lag.py) -- no equation solver, no projection; the constraints enter only through the multipliers:

    actor loss = mean(-Q(s, a) [+ alpha log pi] + nu . relu(g(s, a)) + lambda . |h(s, a)|)      ddpg_lag.py:252-266

They share the vectorised envs, the replay ring, the flat parameter buffer and the fused optimiser kernels with the RPO
trainers ("cheap once the envs exist"); the networks run through the torch modules and autograd
(the baselines are not part of the hot path that is tuned and benchmarked); the iterations are hipGraph-captured.  Same constructor
arguments as the reference (ddpg_lag.py:13-21, sac_lag.py:12-20) plus the vectorisation keywords of the RPO trainers.
"""
import torch

from .. import ops as hip_ops
from .agent import PDDDPG_PA, PDSAC_PA
from .model import BoxConstraint
from .trainer import _SALT_ACTOR, _SALT_CRITIC, RPOTrainerBase, _TDHuberFn


class _LagrangianBase(RPOTrainerBase):
    """process_action is the identity (ddpg_lag.py:72-75); both multipliers are stepped (ddpg_lag.py:196-198)."""

    def _setup_la(self, env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend, shape):
        # torch modules + autograd: the baselines are comparison points, not part of the tuned hot path.  Their iterations
        # are hipGraph-captured like the RPO trainers' (windows of RPO_GRAPH_CYCLE iterations) whenever nothing in them
        # depends on host state: not with `shape` (the reward shaping indexes the ring by the host's position) and not with
        # the automatic entropy tuning (alpha is read back to the host for the TD kernel)
        graph_ok = not shape and not getattr(self, "automatic_entropy_tuning", False)
        self._setup(env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend,
                    use_graph=None if graph_ok else False, fused=False)
        self.shape = shape
        A = self.kernels.action_dim
        self._noise_b = torch.zeros(self.batch_size, A, device=device)
        self._noise_n = torch.zeros(self.n_local, A, device=device)

    # ------------------------------------------------------------------------------------------ policy
    def process_action(self, state, action_partial, train=True):
        return action_partial

    def _box(self, state):
        box = self.agent.actor.box_constraint
        if box.volatile:
            return self.base_env.update(state, full=True)
        return box.cmin_torch, box.cmax_torch

    def _explore(self, obs, warm):
        raise NotImplementedError

    def _rollout(self, warm, defer_clock=False):          # (no column-split update here: the step kernel keeps its clock)
        v, buf = self.vec, self.buffer
        with torch.no_grad():
            if warm:                                             # agent.random_action: uniform in the (full) box
                lo, hi = self._box(v.obs)
                a = lo + torch.rand(v.n, self.kernels.action_dim, device=self.device) * (hi - lo)
            else:
                a = self._explore(v.obs)
            v.action.copy_(a)
            base = buf.pointer                                   # host mirror of the ring position (eager loop)
            v.step(v.action, rows=buf.rows, cap_steps=buf.capacity, auto_reset=True)
            if self.shape:                                       # reward - 10 max|eq| - 10 max ineq (ddpg_lag.py:115-116)
                c = self.kernels.cols
                rows = buf.rows[base:base + v.n]
                max_eq = rows[:, c["eq_viol"][0]:c["eq_viol"][1]].abs().max(dim=1).values
                pen = max_eq + rows[:, c["ineq_viol"][0]:c["ineq_viol"][1]].max(dim=1).values
                rows[:, c["reward"][0]] -= 10.0 * pen

    def _eval_action(self, v):
        v.action.copy_(self._deterministic(v.obs))

    def grad_steps(self, state, action, train=True):
        return action if train else (action, 0)

    # ------------------------------------------------------------------------------------------ losses
    def _penalty(self, state, actions):
        """mean_b(nu . relu(g) + lambda . |h|) through the env's differentiable constraint API (ddpg_lag.py:256-263)."""
        ag = self.agent
        ineq = self.base_env.ineq_dist(state, actions)
        eq = self.base_env.eq_resid(state, actions)
        return (ag.nju(ineq) + ag.lamb(torch.abs(eq))).mean()

    def _actor_step(self, actor_out):
        ag = self.agent
        ag.actor_optim.step()
        if not self.fixed:
            ag.lamb_optim.step()                                 # ddpg_lag.py:196-198
            ag.nju_optim.step()
        self._after_actor_step(actor_out)


class DDPG_LA(_LagrangianBase):

    def __init__(self, env, work_dir, name, logger, max_steps=10, embed_dim=256, hidden_dim=256, hidden_layer=1,
                 shared_param=True, value_type="add", ex_action_dim=0, lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4,
                 reg=0, eps=0.1, tau=0.005, gamma=0.95, capacity=10000, warmup=1000, corr_lr=1e-5, corr_mode=0,
                 corr_eps=1e-3, corr_momentum=0.5, batch_size=256, policy_fre=2, eval_fre=500, max_epochs=100000,
                 grad_eps=1e-3, eval_steps=None, init_lamb=0.0, init_nju=0.0, fixed=False, clip_thres="inf", shape=False,
                 device=torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"),
                 num_envs=None, seed=None, backend=None):
        base = getattr(env, "unwrapped", env)
        agent = PDDDPG_PA(
            base.state_dim, base.action_dim, base.eq_num, base.ineq_num, embed_dim=embed_dim, hidden_dim=hidden_dim,
            hidden_layer=hidden_layer, shared_param=shared_param, value_type=value_type, ex_action_dim=ex_action_dim,
            box_constraint=BoxConstraint(*base.box_constraint, device=device, volatile=base.volatile, update=base.update,
                                         full=True),
            lr_actor=lr_actor, lr_critic=lr_critic, lr_dual=lr_dual, reg=reg, eps=eps, tau=tau, gamma=gamma,
            capacity=capacity, init_lamb=init_lamb, init_nju=init_nju, device=device, backend=backend,
            clip_thres=clip_thres, full_action=True)
        hp = dict(max_steps=max_steps, corr_lr=corr_lr, eval_lr=corr_lr, corr_eps=corr_eps, corr_momentum=corr_momentum,
                  corr_mode=corr_mode, grad_eps=grad_eps, clip_thres=clip_thres, eval_steps=eval_steps,
                  batch_size=batch_size, policy_fre=policy_fre, eval_fre=eval_fre, warmup=warmup, max_epochs=max_epochs,
                  fixed=fixed, partial=False, eps=eps, eps_start=eps, eps_epoch=1)      # constant exploration scale
        self._setup_la(env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend, shape)

    def _noisy(self, ap, state, noise):
        return self.agent.actor.box_constraint.clip(ap + self.agent.eps * noise, state)      # agent/ddpg.py take_action

    def _explore(self, obs):
        self.backend.philox_normal(self._noise_n, self.seed, self.vec.env_id_base * self._noise_n.shape[1], 0,
                                   hip_ops.CONST["RPO_STREAM_ACT"], self.vec.ctrl)
        return self._noisy(self.agent.actor(obs), obs, self._noise_n)

    def _deterministic(self, obs):
        return self.agent.actor(obs)

    def critic_loss(self, state, action, next_state, done, reward, ineq_viol=None, eq_viol=None):
        """ddpg_lag.py:269-279."""
        ag = self.agent
        with torch.no_grad():
            next_q = ag.critic_target(next_state, ag.actor_target(next_state))
        return _TDHuberFn.apply(self.backend, ag.gamma, 0.0, reward, done, next_q, None, None, ag.critic(state, action), None)

    def actor_loss(self, state):
        """ddpg_lag.py:252-266."""
        ag = self.agent
        self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * self._noise_b.numel(), _SALT_ACTOR,
                                   hip_ops.STREAM_POLICY, self.vec.ctrl)
        actions = self._noisy(ag.actor(state), state, self._noise_b)
        return (-ag.critic(state, actions)).mean() + self._penalty(state, actions)

    def _critic_step(self, actor_step):
        self.agent.critic_optim.step()

    def _after_actor_step(self, actor_out):
        self.agent.soft_update()                                 # only on policy steps (ddpg_lag.py:200)


class SAC_LA(_LagrangianBase):
    sac = True

    def __init__(self, env, work_dir, name, logger, automatic_entropy_tuning=True, alpha=0.2, max_steps=10,
                 embed_dim=256, hidden_dim=256, hidden_layer=1, shared_param=True, value_type="add", ex_action_dim=0,
                 lr_alpha=1e-4, lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4, reg=0, eps=0.1, tau=0.005, gamma=0.95,
                 capacity=10000, warmup=1000, corr_lr=1e-5, corr_mode=0, corr_eps=1e-3, corr_momentum=0.5,
                 batch_size=256, policy_fre=2, eval_fre=500, max_epochs=100000, grad_eps=1e-3, eval_steps=None,
                 init_lamb=0.0, init_nju=0.0, fixed=False, clip_thres="inf", shape=False,
                 device=torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"),
                 num_envs=None, seed=None, backend=None):
        base = getattr(env, "unwrapped", env)
        agent = PDSAC_PA(
            automatic_entropy_tuning, base.state_dim, base.action_dim, base.eq_num, base.ineq_num, embed_dim=embed_dim,
            hidden_dim=hidden_dim, hidden_layer=hidden_layer, shared_param=shared_param, value_type=value_type,
            ex_action_dim=ex_action_dim,
            box_constraint=BoxConstraint(*base.box_constraint, device=device, volatile=base.volatile, update=base.update,
                                         full=True),
            alpha=alpha, lr_alpha=lr_alpha, lr_actor=lr_actor, lr_critic=lr_critic, lr_dual=lr_dual, reg=reg, eps=eps,
            tau=tau, gamma=gamma, capacity=capacity, init_lamb=init_lamb, init_nju=init_nju, device=device,
            backend=backend, clip_thres=clip_thres, full_action=True)
        self.automatic_entropy_tuning = automatic_entropy_tuning
        hp = dict(max_steps=max_steps, corr_lr=corr_lr, eval_lr=corr_lr, corr_eps=corr_eps, corr_momentum=corr_momentum,
                  corr_mode=corr_mode, grad_eps=grad_eps, clip_thres=clip_thres, eval_steps=eval_steps,
                  batch_size=batch_size, policy_fre=policy_fre, eval_fre=eval_fre, warmup=warmup, max_epochs=max_epochs,
                  fixed=fixed, partial=False, eps=eps, eps_start=eps, eps_epoch=1)
        self._setup_la(env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend, shape)

    def _sample_action(self, state, noise, log_pi=False):
        return self.agent.take_action(state, log_pi=log_pi, eps=noise)                        # agent/sac.py take_action

    def _explore(self, obs):
        self.backend.philox_normal(self._noise_n, self.seed, self.vec.env_id_base * self._noise_n.shape[1], 0,
                                   hip_ops.STREAM_POLICY, self.vec.ctrl)
        return self._sample_action(obs, self._noise_n)

    def _deterministic(self, obs):
        return self.agent.take_action(obs, deterministic=True)

    def critic_loss(self, state, action, next_state, done, reward, ineq_viol=None, eq_viol=None):
        """sac_lag.py:300-311."""
        ag = self.agent
        with torch.no_grad():
            self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * self._noise_b.numel(), _SALT_CRITIC,
                                       hip_ops.STREAM_POLICY, self.vec.ctrl)
            next_actions, logp = self._sample_action(next_state, self._noise_b, log_pi=True)
            nq1, nq2 = ag.critic_target(next_state, next_actions)
        q1, q2 = ag.critic(state, action)
        return _TDHuberFn.apply(self.backend, ag.gamma, float(ag.alpha), reward, done, nq1, nq2, logp, q1, q2)

    def actor_loss(self, state):
        """sac_lag.py:280-297 -> (loss, log_pi)."""
        ag = self.agent
        self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * self._noise_b.numel(), _SALT_ACTOR,
                                   hip_ops.STREAM_POLICY, self.vec.ctrl)
        actions, logp = self._sample_action(state, self._noise_b, log_pi=True)
        q1, q2 = ag.critic(state, actions)
        return (ag.alpha * logp - torch.min(q1, q2)).mean() + self._penalty(state, actions), logp

    def _critic_step(self, actor_step):
        self.agent.critic_optim.step()
        if not actor_step:
            self.agent.soft_update()                             # every step (sac_lag.py:219)

    def _after_actor_step(self, actor_out):
        ag = self.agent
        if self.automatic_entropy_tuning:
            _, logp = actor_out
            # d/d log_alpha of -(log_alpha (log pi + H_target)).mean() (sac_lag.py:208-215), stepped by the fused Adam
            torch.neg(logp.detach().mean() + ag.target_entropy, out=ag.log_alpha.grad.view(()))
            ag.alpha_optim.step()
            ag.alpha = ag.log_alpha.detach().exp()
        ag.soft_update()
