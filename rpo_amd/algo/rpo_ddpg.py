"""RPO-DDPG on MI355X (reference: rpo/algo/rpo_ddpg.py).  Same constructor, attributes and methods as the reference's
``RPODDPG``; the loop is vectorised over ``num_envs`` lanes (extra keyword / ``RPO_NUM_ENVS``, default 1)."""
import torch

from .. import ops as hip_ops
from .agent import PDDDPG_PA
from .model import BoxConstraint
from .agent.flat import FusedAdam
from .trainer import _SALT_ACTOR, RPOTrainerBase, _LagrangianFn, _TDHuberFn


class _LazySum(object):
    """Per-workgroup loss partials, summed only when somebody looks (keeps a reduction kernel out of the iteration)."""

    def __init__(self, parts):
        self.parts = parts

    def __float__(self):
        return float(self.parts.sum())

    def detach(self):
        return self.parts.sum()


class _LazyDiff(object):
    """mean Lagrangian term - mean Q from the two words the actor-backward pipeline leaves on the device."""

    def __init__(self, pair):
        self.pair = pair

    def __float__(self):
        return float(self.pair[0] - self.pair[1])

    def detach(self):
        return self.pair[0] - self.pair[1]


class _LazyMeanDiff(object):
    """mean Lagrangian term (one device word) - mean Q (a [B, 1] buffer): no reduction launch inside the iteration."""

    def __init__(self, lag, q):
        self.lag, self.q = lag, q

    def __float__(self):
        return float(self.detach())

    def detach(self):
        return self.lag[0] - self.q.mean()


class RPODDPG(RPOTrainerBase):

    def __init__(self, env, work_dir, name, logger, max_steps=10, embed_dim=256, hidden_dim=256, hidden_layer=1,
                 shared_param=True, value_type="add", ex_action_dim=0, lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4,
                 reg=0, eps=0.1, eps_start=1.0, eps_epoch=10000, tau=0.005, gamma=0.95, capacity=10000, warmup=1000,
                 corr_lr=1e-5, eval_lr=1e-5, corr_mode=0, corr_eps=1e-5, corr_momentum=0.5, batch_size=256,
                 policy_fre=2, eval_fre=500, max_epochs=100000, grad_eps=1e-3, eval_steps=None, init_lamb=0.0,
                 init_nju=0.0, fixed=False, clip_thres="inf", partial=False, partial_idx=None,
                 device=torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"),
                 num_envs=None, seed=None, backend=None, use_graph=None, updates_per_step=None, schedule=None):
        base = getattr(env, "unwrapped", env)
        agent = PDDDPG_PA(
            base.state_dim, base.action_dim, base.eq_num, base.ineq_num, embed_dim=embed_dim, hidden_dim=hidden_dim,
            hidden_layer=hidden_layer, shared_param=shared_param, value_type=value_type, ex_action_dim=ex_action_dim,
            box_constraint=BoxConstraint(*base.box_constraint_partial, device=device, volatile=base.volatile,
                                         update=base.update),
            lr_actor=lr_actor, lr_critic=lr_critic, lr_dual=lr_dual, reg=reg, eps=eps_start, tau=tau, gamma=gamma,
            capacity=capacity, init_lamb=init_lamb, init_nju=init_nju, partial=partial,
            partial_idx=base.partial_actions if partial_idx is None else partial_idx, device=device, backend=backend,
            clip_thres=clip_thres)
        hp = dict(max_steps=max_steps, corr_lr=corr_lr, eval_lr=eval_lr, corr_eps=corr_eps, corr_momentum=corr_momentum,
                  corr_mode=corr_mode, grad_eps=grad_eps, clip_thres=clip_thres, eval_steps=eval_steps,
                  batch_size=batch_size, policy_fre=policy_fre, eval_fre=eval_fre, warmup=warmup, max_epochs=max_epochs,
                  fixed=fixed, partial=partial, eps=eps, eps_start=eps_start, eps_epoch=eps_epoch)
        self._setup(env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend, use_graph, updates_per_step,
                    schedule=schedule)

    # ---- rollout policy (rpo_ddpg.py:98-106, agent/ddpg_pa.py:101-112) ----------------------------------------
    def _actor_out(self, name, obs, save=False, tag=None):
        """Deterministic basic action [n] of `actor` / `actor_target` through the fused MLP kernel.  `tag`: the output buffer's
        name when it must not be the rollout's (the policy step's prefix runs on a second stream BESIDE other launches: with
        num_envs == batch_size the shape-keyed "actor.out" would be one buffer for both)."""
        f = self.fused
        P = self.kernels.partial_dim
        # P > 1 (EVOPF): raw outputs; the env kernels apply the state-dependent tanh box (ap_is_raw)
        return f.forward(name, obs, None, f.buf((tag or name) + ".out", obs.shape[0], P), save=save,
                         tanh_box=self._box_affine).view(-1)

    def _policy_partial(self, obs, warm):
        if warm:
            return None, hip_ops.NOISE_UNIFORM                      # BoxConstraint.sample, inside the kernel
        if self.fused is not None:
            return self._actor_out("actor", obs), hip_ops.NOISE_PHILOX
        return self.agent.actor(obs).reshape(-1), hip_ops.NOISE_PHILOX   # + eps_t * N(0,1), clip: inside the kernel

    def _eval_partial(self, obs):
        if self.fused is not None:
            return self._actor_out("actor", obs)
        return self.agent.actor(obs).reshape(-1)

    # ---- fused pipelines (one launch per stage group; CartSafe kernels provide them) -----------------------------
    @property
    def _pipelines(self):
        k = self.kernels
        return (self.fused is not None and (hasattr(k, "ddpg_critic_forward") or hasattr(k, "ddpg_critic_front"))
                and "actor_target" in self.fused.descs and "critic" in self.fused.descs
                and self.schedule["fused_critic"] and not self._large_batch)

    def _sample(self):
        if self._pipelines:
            # the critic-forward pipeline draws and gathers the batch itself (into self._batch)
            c = self.buffer.split(self._batch)
            return c["state"], c["action"], c["next_state"], c["reward"], c["done"], c["ineq_viol"], c["eq_viol"]
        return super()._sample()

    def _critic_update_pipeline(self, cols):
        f, ag, B, buf = self.fused, self.agent, self.batch_size, self.buffer
        state, action = cols[0], cols[1]
        d = f.descs["critic"]
        scale, base = self._box_affine
        parts = f.buf("loss_parts", (B + 15) // 16)
        idx_in = self._idx_inject() if self._idx_inject is not None else None
        if not hasattr(self.kernels, "ddpg_critic_forward"):
            # SpringPendulum: the chain is cut at the batch-coupled projection (front | project | back)
            ap = f.buf("crit.ap", B)
            self.kernels.ddpg_critic_front(f.descs["actor_target"], scale, base, buf.rows, buf.capacity, buf.n_envs,
                                           self._batch, None, idx_in, buf.seed, 0, self._uctrl, ap)
            next_actions = self._project_batch(cols[2], ap)
            q, qn = f.buf("q", B, 1), f.buf("qn", B, 1)
            self.kernels.ddpg_critic_back(f.descs["critic_target"], d, self._batch, next_actions, q, qn,
                                          f.buf("critic.x0", B, d.ein), f.buf("critic.h1", B, d.H))
            return self._critic_backward_td(cols, q, qn, parts)
        q, qn = f.buf("q", B, 1), f.buf("qn", B, 1)
        self.kernels.ddpg_critic_forward(
            f.descs["actor_target"], f.descs["critic_target"], d, scale, base, buf.rows, buf.capacity, buf.n_envs,
            self._batch, None, idx_in, buf.seed, 0, self._uctrl, self.max_steps, self.corr_lr, self.corr_eps,
            self.corr_momentum, self._box_lo, self._box_hi, q, qn, f.buf("critic.x0", B, d.ein), f.buf("critic.h1", B, d.H))
        self._critic_backward_td(cols, q, qn, parts)

    def _critic_backward_td(self, cols, q, qn, parts):
        """Backward of the critic with the TD target / Huber loss as its prologue (rpo_td): the forward pipeline's target
        chain and critic ran in separate workgroups and meet here."""
        f, B = self.fused, self.batch_size
        td = self.backend.Td(q.view(-1), qn.view(-1), None, None, cols[3], cols[4], 0.0, self.agent.gamma,
                             f.buf("dq", B, 1).view(-1), parts)
        self._zero_grads()
        gm = self._critic_gradmax()
        f.backward("critic", cols[0], cols[1], None, gradmax=gm, td=td)
        self._gradmax_ready = gm is not None
        self.last_losses["critic"] = _LazySum(parts)

    # ---- the update through the hand-written MLP kernels (same arithmetic as critic_loss / actor_loss below) ------
    def _critic_update(self, cols):
        if self.fused is None:
            return super()._critic_update(cols)
        if self._pipelines:
            su = self._split_state()
            if su is not None:
                return self._critic_update_split(su)
            return self._critic_update_pipeline(cols)
        f, ag, B = self.fused, self.agent, self.batch_size
        state, action, next_state, reward, done = cols[:5]
        next_actions = self._project_batch(next_state, self._actor_out("actor_target", next_state))
        qn, q = f.forward_multi([("critic_target", next_state, next_actions, f.buf("qn", B, 1), False),
                                 ("critic", state, action, f.buf("q", B, 1), True)])
        # TD target / Huber loss: the prologue of the backward pass (rpo_td)
        self._critic_backward_td(cols, q, qn, f.buf("loss_parts", (B + 15) // 16))

    @property
    def _actor_pipeline(self):
        d = self.fused.descs if self.fused is not None else {}
        return (hasattr(self.backend, "ddpg_actor_forward") and "actor" in d and "critic" in d and d["actor"].E == 128
                and d["critic"].E == 128 and not d["critic"].cat and self._box_affine is not None
                and self.kernels.partial_dim == 1 and self.kernels.action_dim == 2
                and self.schedule["fused_actor"] and not self._large_batch)

    def _actor_update_pipeline(self, cols):
        """The policy step in two launches + the actor's weights pass (fused.hip)."""
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        da_, dc = f.descs["actor"], f.descs["critic"]
        scale, base = self._box_affine
        b = f.buf
        parts = b("actor.parts", (B + 15) // 16, 8)
        ap_det, noise, actions = b("act.ap_det", B), b("act.noise", B), b("act_pi", B, k.action_dim)
        q, dq, g_act = b("q_pi", B, 1), b("dq_pi", B, 1), b("g_act", B, k.action_dim)
        noise_in = None
        if self._idx_inject is not None:                       # tests replay the reference's draw
            self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * B * k.partial_dim, _SALT_ACTOR, hip_ops.STREAM_POLICY,
                                       self._uctrl)
            noise_in = self._noise_b.view(-1)
        self.backend.ddpg_actor_forward(k, da_, dc, scale, base, self._box_lo, self._box_hi, self.eps_start, self.eps, self.decay_value,
                             self._batch, noise_in, self.seed, self.dist.rank * B, _SALT_ACTOR, self._uctrl,
                             ag.nju.weight.view(-1), ap_det, noise, actions, q, dq, g_act, parts, b("actor.x0", B, da_.ein),
                             b("actor.h1", B, da_.H), b("critic.x0", B, dc.ein), b("critic.h1", B, dc.H))
        self._zero_grads()
        shared = ag.flat.sizes[1] > 0
        opt = ag.actor_optim
        fuse_max = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf")
        lag = b("actor.lag", 2)
        self.backend.ddpg_actor_backward(k, da_, dc, shared, self._batch, actions, g_act, ap_det, noise, dq, self.eps_start, self.eps,
                              self.decay_value, self._box_lo, self._box_hi, scale, base, self._uctrl,
                              b("actor.x0", B, da_.ein), b("actor.h1", B, da_.H), b("critic.x0", B, dc.ein),
                              b("critic.h1", B, dc.H), b("actor.dh", B, da_.H), b("actor.dx0", B, da_.ein),
                              b("critic.dh", B, dc.H), b("critic.dx0", B, dc.ein), b("da", B, k.action_dim), b("do", B),
                              parts, lag, ag.nju.weight.grad.view(-1), opt.gradmax if fuse_max else None)
        self._actor_gradmax_ready = bool(fuse_max)
        loss = _LazyDiff(lag)
        self.last_losses["actor"] = loss
        return loss

    def _actor_update(self, cols):
        if self.fused is None:
            return super()._actor_update(cols)
        if self._actor_pipeline:
            su = self._split_state()
            if su is not None:
                lag, _, _ = self._actor_update_split(su)
                loss = _LazyDiff(lag)
                self.last_losses["actor"] = loss
                return loss
            return self._actor_update_pipeline(cols)
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        state = cols[0]
        pre, self._actor_pre = getattr(self, "_actor_pre", None), None
        if pre is None:
            pre = self._actor_prefix(cols)
        ap_det, noise, actions, lag, g_act, fa = pre
        # (its own buffer: the lazily reduced actor loss below keeps a reference to it, and the next critic-only update writes
        #  Q(s, a_replay) into "q" -- ADVICE r03)
        q = f.forward("critic", state, actions, f.buf("q_pi", B, 1), save=True)
        dq = self._const_dq(B)                                 # d mean(-Q) / dQ: a constant, filled once
        da = f.buf("da", B, k.action_dim)
        shared = ag.flat.sizes[1] > 0      # shared embedding: the critic path contributes to its gradient
        f.backward("critic", state, actions, dq, da=da, param_grads=shared, first_layer_state_only=True)
        P = k.partial_dim
        dap, do = f.buf("dap", B * P), f.buf("do", B, P)
        if fa:
            k.complete_bwd(state, da, dap, action=actions, grad_action2=g_act)
        else:
            da.add_(g_act)
            k.complete_bwd(state, da, dap, action=actions)
        if self._box_affine is None:       # state-dependent box: the env's kernel knows it
            k.tanh_box_bwd(state, ap_det, noise, self.eps_start, self.eps, self.decay_value, self._uctrl, dap, do.view(-1))
        else:
            scale, base = self._box_affine
            self.backend.tanh_box_bwd(dap, ap_det, noise, self.eps_start, self.eps, self.decay_value, self._uctrl,
                                      self._box_lo, self._box_hi, scale, base, do.view(-1))
        # (no shared embedding: every gradient of the actor's slice is written by this backward -- it leaves the inf-norm for
        #  clip_grad_norm_, no rpo_absmax launch)
        opt = ag.actor_optim
        fuse_max = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf") and not shared
        f.backward("actor", state, None, do, gradmax=opt.gradmax if fuse_max else None)
        self._actor_gradmax_ready = bool(fuse_max)
        loss = _LazyMeanDiff(lag, q)                          # lag[0] - mean(Q), reduced only when somebody looks
        self.last_losses["actor"] = loss
        return loss

    def _actor_prefix(self, cols):
        """What the policy step computes from the ACTOR alone (generic launches): pi(s) with saved activations, the exploration
        draw, Complete, and the Lagrangian term of the completed actions (its d/d action and d/d nu).  None of it reads what
        the critic update writes, so the graph windows of EVOPF-v0 run it on the second captured branch BESIDE the critic
        update of a policy iteration (trainer._overlapped_window; round 5: ~50 us off a 364 us chain) -- same launches, same
        arguments, same bits as inside `_actor_update`."""
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        state = cols[0]
        ap_det = self._actor_out("actor", state, save=True, tag="pi")
        self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * B * k.partial_dim, _SALT_ACTOR, hip_ops.STREAM_POLICY,
                                   self._uctrl)
        noise = self._noise_b.view(-1)
        actions = self._complete_only(state, ap_det, noise)
        lag, g_act = f.buf("loss_lag", 1), f.buf("g_act", B, k.action_dim)
        # (EVOPF kernels write the loss term and add the Lagrangian's d/d action themselves: no torch launch in the step)
        fa = bool(getattr(k, "fused_adds", False))
        if not fa:
            lag.zero_()
        self._zero_grads()                 # parameters AND multipliers (they live in the same flat buffer)
        k.lagrangian(actions, ag.nju.weight.view(-1), 1.0 / B, lag, g_act, ag.nju.weight.grad.view(-1), obs=state,
                     **(dict(overwrite=True) if fa else {}))
        return ap_det, noise, actions, lag, g_act, fa

    def _const_dq(self, B):
        key = ("dq_pi_const", B)
        if key not in self.fused._scratch:
            self.fused._scratch[key] = torch.full((B, 1), -1.0 / B, device=self.device)
        return self.fused._scratch[key]

    # ---- losses ---------------------------------------------------------------------------------------------
    def critic_loss(self, state, action, next_state, done, reward, ineq_viol=None, eq_viol=None):
        """Huber(Q(s,a), r + gamma (1-d) Q_targ(s', Proj(Complete(pi_targ(s'))))) (rpo_ddpg.py:327-337)."""
        ag = self.agent
        with torch.no_grad():
            next_partial = ag.take_action(next_state, deterministic=True, target=True)
            next_actions = self.process_action(next_state, next_partial)
            next_q = ag.critic_target(next_state, next_actions)
        q = ag.critic(state, action)
        return _TDHuberFn.apply(self.backend, ag.gamma, 0.0, reward, done, next_q, None, None, q, None)

    def actor_loss(self, state):
        """mean(-Q(s, Complete(pi(s) + noise)) + nu . relu(g)) (rpo_ddpg.py:307-324); the exploration noise of
        take_action comes from the Philox stream instead of torch's global generator."""
        ag = self.agent
        ap = ag.actor(state)
        self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * self.batch_size * self.kernels.partial_dim, _SALT_ACTOR,
                                   hip_ops.STREAM_POLICY, self._uctrl)
        ap = ag.actor.box_constraint.clip(ap + self._eps_now() * self._noise_b, state)
        actions = self.base_env.complete_partial(state, ap)
        loss = (-ag.critic(state, actions)).mean()
        return loss + _LagrangianFn.apply(self.kernels, actions, ag.nju.weight, state)

    # ---- optimiser steps (rpo_ddpg.py:178-205) --------------------------------------------------------------
    # The Polyak updates of a policy step (rpo_ddpg.py:205) ride in the Adam launches that produce the parameters they
    # average -- same arithmetic, two launches fewer.  With a shared state embedding the critic's target must see the
    # embedding AFTER the actor's step, so its update stays a separate launch at the end.
    def _critic_step(self, actor_step):
        ag = self.agent
        fuse = actor_step and ag.flat.sizes[1] == 0
        prepared, self._critic_prepared = getattr(self, "_critic_prepared", False), False
        ag.critic_optim.step(target=ag.critic_target_flat if fuse else None, tau=ag.tau,
                             gradmax_ready=self._gradmax_ready, clock=self._clock(not actor_step), prepared=prepared)
        self._gradmax_ready = False

    def _actor_step(self, actor_out):
        # actor Adam (+ Polyak of the actor target) | multiplier DualAdam | Polyak of the critic target: one launch.
        # With a shared state embedding the actor's step moves the critic's copy of it too, so the shared part of the
        # critic target follows inside the actor's slice (target2) and only the critic-only part is a slice of its own.
        ag, fl = self.agent, self.agent.flat
        c = fl.actor_range[0]                                  # [critic-only | shared | actor-only], padded offsets
        sh = fl.critic_range[1] - c if fl.sizes[1] > 0 else 0
        segs = [ag.actor_optim.segment(target=ag.actor_target_flat, tau=ag.tau,
                                       gradmax_ready=getattr(self, "_actor_gradmax_ready", False),
                                       target2=ag.critic_target_flat[c:c + sh] if sh > 0 else None, n2=sh)]
        self._actor_gradmax_ready = False
        if not self.fixed:
            segs.append(ag.nju_optim.segment())                    # lambda is never stepped (rpo_ddpg.py:202)
        if sh > 0 and c > 0:
            segs.append(dict(polyak_only=True, param=fl.param((0, c)), target=ag.critic_target_flat[:c], tau=ag.tau))
        prepared, self._actor_prepared = getattr(self, "_actor_prepared", False), False
        FusedAdam.step_many(self.backend, segs, clock=self._clock(True), prepared=prepared)
