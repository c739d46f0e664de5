"""Primal-dual agents acting on the basic ("partial") actions: networks, targets, optimisers, multipliers, replay.

Mirrors ``PDDDPG_PA`` (rpo/algo/agent/ddpg_pa.py:18-119) and ``PDSAC_PA`` (rpo/algo/agent/sac_pa.py:18-127): same
constructor arguments, same attributes (``actor``, ``critic``, ``critic_target``, ``actor_target``, ``nju``, ``lamb``,
``replay_buffer``, ``eps``, ``tau``, ``gamma``) and methods (``take_action``, ``random_action``, ``add``,
``soft_update``, ``hard_update``, ``eps_decay``, ``save_model``, ``load_model``).  Differences, all MI355X-driven:
parameters live in one flat HBM buffer stepped by fused HIP kernels (flat.py), and the replay buffer is the device
ring of rpo_amd/utils/buffer.py, attached by the trainer once the env kernels and the lane count are known.
"""
import copy
import os

import numpy as np
import torch

from ... import ops as hip_ops
from ...utils.buffer import ReplayBuffer
from ..model import (ActionEmbedding, DoubleValueAdd, DoubleValueCat, Dual, GaussianSharedPolicy, SharedPolicy,
                     SharedValueAdd, SharedValueCat, StateEmbedding)
from .flat import FlatParams, FusedAdam


class Agent(object):
    """rpo/algo/agent/base.py:3-16."""

    def __init__(self, state_dim, action_dim, box_constraint):
        self.state_dim, self.action_dim, self.box_constraint = state_dim, action_dim, box_constraint

    def take_action(self, state, deterministic=False):
        return self.box_constraint(np.random.rand(self.action_dim))


class _PartialActionAgent(Agent):

    def __init__(self, state_dim, action_dim, eq_num, ineq_num, box_constraint, lr_actor, lr_critic, lr_dual, reg, eps,
                 tau, gamma, capacity, partial, partial_idx, init_lamb, init_nju, device, backend, clip_thres):
        super().__init__(state_dim, action_dim, box_constraint)
        self.partial, self.partial_idx = partial, partial_idx
        self.eq_num, self.ineq_num = eq_num, ineq_num
        self.tau, self.eps, self.gamma, self.capacity = tau, eps, gamma, capacity
        self.device = device
        self.backend = backend if backend is not None else hip_ops
        self.lr_actor, self.lr_critic, self.lr_dual, self.reg, self.clip_thres = lr_actor, lr_critic, lr_dual, reg, clip_thres
        self._init_duals = (init_lamb, init_nju)
        self.replay_buffer = None

    # ---- construction helpers ---------------------------------------------------------------------------------
    def _finish(self, has_actor_target, extra_params=()):
        """Called by subclasses after ``self.actor`` / ``self.critic`` exist (built on the CPU in the reference's
        order so that the same torch seed gives the same initial weights)."""
        dev = self.device
        self.actor_target = copy.deepcopy(self.actor) if has_actor_target else None
        self.critic_target = copy.deepcopy(self.critic)
        # multipliers: lambda (equalities, never stepped: rpo_ddpg.py:202) and nu (inequalities, DualAdam ascent)
        self.lamb = Dual(self.eq_num)
        self.nju = Dual(self.ineq_num)
        self.lamb.reset_parameters(self._init_duals[0])
        self.nju.reset_parameters(self._init_duals[1])
        # `extra_params` (RPOSAC's log_alpha) ride behind the multipliers: same flat buffer, same policy-step bucket
        self.flat = FlatParams(self.critic, self.actor, dev, extras=[self.lamb.weight, self.nju.weight] + list(extra_params))
        self.critic_target_flat = self.flat.make_target(self.critic, self.critic_target, self.flat.critic_range)
        self.actor_target_flat = self.flat.make_target(self.actor, self.actor_target, self.flat.actor_range) \
            if has_actor_target else None
        for net in (self.actor, self.critic, self.actor_target, self.critic_target):
            box = getattr(net, "box_constraint", None)
            if box is not None:
                box.to(dev)
        clip = 0.0 if self.clip_thres in ("inf", float("inf"), None) else float(self.clip_thres)
        self.critic_optim = FusedAdam(self.backend, self.flat.param(self.flat.critic_range),
                                      self.flat.gradient(self.flat.critic_range), self.lr_critic, self.reg, clip)
        self.actor_optim = FusedAdam(self.backend, self.flat.param(self.flat.actor_range),
                                     self.flat.gradient(self.flat.actor_range), self.lr_actor, self.reg, clip)
        self.lamb_optim = FusedAdam(self.backend, self.lamb.weight.data.view(-1), self.lamb.weight.grad.view(-1),
                                    self.lr_dual, maximize=True)
        self.nju_optim = FusedAdam(self.backend, self.nju.weight.data.view(-1), self.nju.weight.grad.view(-1),
                                   self.lr_dual, maximize=True, clamp_min0=True)

    def attach_env(self, kernels, n_envs, seed, ctrl):
        """Create the device replay ring (ReplayBuffer(capacity, ...) of agent/ddpg_pa.py:70-71, per-env capacity)."""
        self.replay_buffer = ReplayBuffer(self.capacity, n_envs, kernels, self.device, seed=seed, ctrl=ctrl,
                                          ops=self.backend)
        return self.replay_buffer

    # ---- reference surface ------------------------------------------------------------------------------------
    def add(self, state, action, next_state, reward, done, eq_viol, ineq_viol):
        self.replay_buffer.add(state=state, action=action, next_state=next_state, reward=reward, done=done,
                               eq_viol=eq_viol, ineq_viol=ineq_viol)

    def random_action(self, x):
        return self.box_constraint.sample(x)

    def eps_decay(self, decay_value, lb):
        self.eps = max(lb, self.eps - decay_value)

    def _polyak(self, flat_param, flat_target):
        self.backend.polyak(flat_param, flat_target, self.tau)

    def hard_update(self):
        self.critic_target_flat.copy_(self.flat.param(self.flat.critic_range))
        if self.actor_target_flat is not None:
            self.actor_target_flat.copy_(self.flat.param(self.flat.actor_range))

    def save_model(self, save_dir):
        """Checkpoint done right (the reference saves parameter *generators*, agent/ddpg_pa.py:92-94, which cannot be
        loaded back): state dicts of actor / critic plus optimiser moments and multipliers."""
        torch.save(self.actor.state_dict(), os.path.join(save_dir, "actor.pth"))
        torch.save(self.critic.state_dict(), os.path.join(save_dir, "critic.pth"))
        torch.save(dict(actor_optim=self.actor_optim.state_dict(), critic_optim=self.critic_optim.state_dict(),
                        nju=self.nju.state_dict(), lamb=self.lamb.state_dict(), nju_optim=self.nju_optim.state_dict(),
                        eps=self.eps, critic_target=self.critic_target_flat, actor_target=self.actor_target_flat,
                        log_alpha=None if getattr(self, "log_alpha", None) is None else self.log_alpha.data.clone(),
                        alpha_optim=None if getattr(self, "log_alpha", None) is None else self.alpha_optim.state_dict()),
                   os.path.join(save_dir, "agent_state.pth"))

    def load_model(self, load_dir):
        dev = self.device
        for net, name in ((self.actor, "actor.pth"), (self.critic, "critic.pth")):
            sd = torch.load(os.path.join(load_dir, name), map_location=dev)
            with torch.no_grad():
                for k, p in net.state_dict().items():
                    p.copy_(sd[k])          # in place: the tensors are views of the flat buffer
        extra = os.path.join(load_dir, "agent_state.pth")
        if os.path.exists(extra):
            st = torch.load(extra, map_location=dev, weights_only=False)
            self.actor_optim.load_state_dict(st["actor_optim"])
            self.critic_optim.load_state_dict(st["critic_optim"])
            self.nju_optim.load_state_dict(st["nju_optim"])
            with torch.no_grad():
                self.nju.weight.copy_(st["nju"]["weight"])
                self.lamb.weight.copy_(st["lamb"]["weight"])
                if getattr(self, "log_alpha", None) is not None and st.get("log_alpha") is not None:
                    self.log_alpha.copy_(st["log_alpha"])
                    self.alpha_optim.load_state_dict(st["alpha_optim"])
            self.eps = st["eps"]
        self.hard_update()
        if os.path.exists(extra) and st.get("critic_target") is not None:      # exact resume: the targets as they were
            self.critic_target_flat.copy_(st["critic_target"])
            if self.actor_target_flat is not None and st.get("actor_target") is not None:
                self.actor_target_flat.copy_(st["actor_target"])


class PDDDPG_PA(_PartialActionAgent):

    def __init__(self, state_dim, action_dim, eq_num, ineq_num, embed_dim=128, hidden_dim=128, hidden_layer=1,
                 shared_param=True, value_type="add", box_constraint=None, lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4,
                 reg=0, eps=0.1, tau=0.001, gamma=0.98, capacity=10000, ex_action_dim=0, partial=False,
                 partial_idx=None, init_lamb=0.0, init_nju=0.0, device=torch.device("cpu"), backend=None,
                 clip_thres="inf", full_action=False):
        super().__init__(state_dim, action_dim, eq_num, ineq_num, box_constraint, lr_actor, lr_critic, lr_dual, reg,
                         eps, tau, gamma, capacity, partial, partial_idx, init_lamb, init_nju, device, backend,
                         clip_thres)
        # full_action: the policy emits the whole action (agent/ddpg.py:34, the Lagrangian baselines' PDDDPG)
        reduced = action_dim if full_action else action_dim - eq_num - ex_action_dim
        q_in = reduced if partial else action_dim
        # construction order == agent/ddpg_pa.py:32-49 (it fixes the RNG stream of the initial weights)
        state_embed = StateEmbedding(state_dim, embed_dim, hidden_dim)
        action_embed = ActionEmbedding(q_in, embed_dim, hidden_dim)
        state_value = state_embed if shared_param else StateEmbedding(state_dim, embed_dim, hidden_dim)
        self.actor = SharedPolicy(state_dim, reduced, state_embed, embed_dim, hidden_dim, hidden_layer, box_constraint)
        if value_type == "add":
            value_cls = SharedValueAdd
        elif value_type == "cat":
            value_cls = SharedValueCat
        else:
            raise Exception("Unknown Value Net!")
        self.critic = value_cls(state_dim, q_in, state_value, action_embed, embed_dim, hidden_dim, partial=partial,
                                partial_idx=partial_idx)
        self._finish(has_actor_target=True)

    def soft_update(self):
        """Polyak update of both targets (agent/ddpg_pa.py:77-86)."""
        self._polyak(self.flat.param(self.flat.actor_range), self.actor_target_flat)
        self._polyak(self.flat.param(self.flat.critic_range), self.critic_target_flat)

    def take_action(self, state, deterministic=False, target=False):
        """agent/ddpg_pa.py:101-112 (torch form; the vectorised rollout fuses noise + clip into the projection kernel)."""
        state = torch.as_tensor(state, device=self.device)
        ap = self.actor_target(state) if target else self.actor(state)
        if not deterministic:
            ap = ap + self.eps * torch.randn_like(ap)
            ap = self.actor.box_constraint.clip(ap, state)
        return ap


class PDSAC_PA(_PartialActionAgent):

    def __init__(self, automatic_entropy_tuning, state_dim, action_dim, eq_num, ineq_num, embed_dim=128, hidden_dim=128,
                 hidden_layer=1, shared_param=True, value_type="add", box_constraint=None, alpha=0.2, lr_alpha=1e-4,
                 lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4, reg=0, eps=0.1, tau=0.005, gamma=0.98, capacity=10000,
                 ex_action_dim=0, partial=False, partial_idx=None, init_lamb=0.0, init_nju=0.0,
                 device=torch.device("cpu"), backend=None, clip_thres="inf", full_action=False):
        super().__init__(state_dim, action_dim, eq_num, ineq_num, box_constraint, lr_actor, lr_critic, lr_dual, reg,
                         eps, tau, gamma, capacity, partial, partial_idx, init_lamb, init_nju, device, backend,
                         clip_thres)
        self.automatic_entropy_tuning = automatic_entropy_tuning
        reduced = action_dim if full_action else action_dim - eq_num - ex_action_dim  # agent/sac.py: PDSAC
        q_in = reduced if partial else action_dim
        # construction order == agent/sac_pa.py:32-52
        state_embed = StateEmbedding(state_dim, embed_dim, hidden_dim)
        action_embed1 = ActionEmbedding(q_in, embed_dim, hidden_dim)
        action_embed2 = ActionEmbedding(q_in, embed_dim, hidden_dim)
        if shared_param:
            sv1 = sv2 = state_embed
        else:
            sv1 = StateEmbedding(state_dim, embed_dim, hidden_dim)
            sv2 = StateEmbedding(state_dim, embed_dim, hidden_dim)
        self.actor = GaussianSharedPolicy(state_dim, reduced, state_embed, embed_dim, hidden_dim, hidden_layer,
                                          box_constraint)
        if value_type == "add":
            value_cls = DoubleValueAdd
        elif value_type == "cat":
            value_cls = DoubleValueCat
        else:
            raise Exception("Unknown Value Net!")
        self.critic = value_cls(state_dim, q_in, sv1, sv2, action_embed1, action_embed2, embed_dim, hidden_dim,
                                partial=partial, partial_idx=partial_idx)
        self.alpha = alpha
        self.log_alpha = None
        if automatic_entropy_tuning:
            # the reference's target entropy is read from an uninitialised tensor (agent/sac_pa.py:60);
            # the conventional -|A_partial| is used instead.  No script enables this path.  log_alpha is one more word
            # of the flat buffer, stepped by the fused Adam kernel: capturable in the iteration's hipGraph, and its
            # gradient travels in the policy-step all-reduce of a data-parallel run.
            self.target_entropy = -float(reduced)
            self.log_alpha = torch.nn.Parameter(torch.zeros(1))
        self._finish(has_actor_target=False, extra_params=[] if self.log_alpha is None else [self.log_alpha])
        if automatic_entropy_tuning:
            self.alpha_optim = FusedAdam(self.backend, self.log_alpha.data.view(-1), self.log_alpha.grad.view(-1), lr_alpha)

    def soft_update(self):
        """Critic-only Polyak update (agent/sac_pa.py:87-91)."""
        self._polyak(self.flat.param(self.flat.critic_range), self.critic_target_flat)

    def take_action(self, state, deterministic=False, log_pi=False, eps=None):
        """agent/sac_pa.py:105-115: sample (or mean) of the squashed Gaussian, clipped to the box."""
        state = torch.as_tensor(state, device=self.device)
        ap, log_prob, mean = self.actor(state, eps=eps)
        if deterministic:
            ap = mean
        ap = self.actor.box_constraint.clip(ap, state)
        return (ap, log_prob) if log_pi else ap

    def actor_log_std(self, state):
        return self.actor(torch.as_tensor(state, device=self.device))[1]
