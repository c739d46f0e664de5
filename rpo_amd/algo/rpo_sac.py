"""RPO-SAC on MI355X (reference: rpo/algo/rpo_sac.py): twin critics, squashed-Gaussian actor, entropy term, critic
Polyak update on every step."""
import torch

from .. import ops as hip_ops
from .agent import PDSAC_PA
from .model import BoxConstraint
from .agent.flat import FusedAdam
from .rpo_ddpg import _LazySum
from .trainer import _SALT_ACTOR, _SALT_CRITIC, RPOTrainerBase, _LagrangianFn, _TDHuberFn


class _LazySumPair(object):
    """mean Lagrangian term (one word left by the actor-backward pipeline) + mean(alpha log pi - min Q) (per-tile sums in
    column 7 of the pipeline's partials), added up only when somebody looks."""

    def __init__(self, lag, parts, batch):
        self.lag, self.parts, self.batch = lag, parts, batch

    def detach(self):
        return self.lag[0] + self.parts[:, 7].sum() / self.batch

    def __float__(self):
        return float(self.detach())


class _LazySacLoss(object):
    """Lagrangian term (one device word) + mean(alpha log pi - min(Q1, Q2)) over buffers that stay valid until the next policy
    step: no reduction launches inside the iteration."""

    def __init__(self, lag, alpha, logp, q1, q2):
        self.lag, self.alpha, self.logp, self.q1, self.q2 = lag, alpha, logp, q1, q2

    def __float__(self):
        return float(self.detach())

    def detach(self):
        return self.lag[0] + (self.alpha * self.logp.view(-1, 1) - torch.min(self.q1, self.q2)).mean()


class RPOSAC(RPOTrainerBase):
    sac = True

    def __init__(self, env, work_dir, name, logger, automatic_entropy_tuning=True, alpha=0.2, max_steps=10,
                 embed_dim=256, hidden_dim=256, hidden_layer=1, shared_param=True, value_type="add", ex_action_dim=0,
                 lr_alpha=1e-4, lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4, reg=0, eps=0.1, eps_start=1.0,
                 eps_epoch=10000, tau=0.005, gamma=0.95, capacity=10000, warmup=1000, corr_lr=1e-5, eval_lr=1e-5,
                 corr_mode=0, corr_eps=1e-5, corr_momentum=0.5, batch_size=256, policy_fre=2, eval_fre=500,
                 max_epochs=100000, grad_eps=1e-3, eval_steps=None, init_lamb=0.0, init_nju=0.0, fixed=False,
                 clip_thres="inf", partial=False, partial_idx=None,
                 device=torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"),
                 num_envs=None, seed=None, backend=None, use_graph=None, updates_per_step=None, schedule=None):
        base = getattr(env, "unwrapped", env)
        agent = PDSAC_PA(
            automatic_entropy_tuning, base.state_dim, base.action_dim, base.eq_num, base.ineq_num,
            embed_dim=embed_dim, hidden_dim=hidden_dim, hidden_layer=hidden_layer, shared_param=shared_param,
            value_type=value_type, ex_action_dim=ex_action_dim,
            box_constraint=BoxConstraint(*base.box_constraint_partial, device=device, volatile=base.volatile,
                                         update=base.update),
            alpha=alpha, lr_alpha=lr_alpha, lr_actor=lr_actor, lr_critic=lr_critic, lr_dual=lr_dual, reg=reg,
            eps=eps_start, tau=tau, gamma=gamma, capacity=capacity, init_lamb=init_lamb, init_nju=init_nju,
            partial=partial, partial_idx=base.partial_actions if partial_idx is None else partial_idx, device=device,
            backend=backend, clip_thres=clip_thres)
        self.automatic_entropy_tuning = automatic_entropy_tuning
        hp = dict(max_steps=max_steps, corr_lr=corr_lr, eval_lr=eval_lr, corr_eps=corr_eps, corr_momentum=corr_momentum,
                  corr_mode=corr_mode, grad_eps=grad_eps, clip_thres=clip_thres, eval_steps=eval_steps,
                  batch_size=batch_size, policy_fre=policy_fre, eval_fre=eval_fre, warmup=warmup, max_epochs=max_epochs,
                  fixed=fixed, partial=partial, eps=eps, eps_start=eps_start, eps_epoch=eps_epoch)
        self._setup(env, work_dir, name, logger, agent, hp, device, num_envs, seed, backend, use_graph, updates_per_step,
                    schedule=schedule)
        self._act_kw = {}        # the Gaussian head kernels emit finished (boxed, clipped) basic actions, never raw ones

    _gauss_policy = True

    def _draw(self, buf, id_base, salt, rollout=False):
        """N(0,1) draws keyed by the step counter: the rollout's (ctrl[T]) or the update's clock (`_uctrl`)."""
        self.backend.philox_normal(buf, self.seed, id_base, salt, hip_ops.STREAM_POLICY,
                                   self.vec.ctrl if rollout else self._uctrl)
        return buf

    # ---- rollout policy (rpo_sac.py:102-110, agent/sac_pa.py:105-115) -------------------------------------------
    def _gauss(self, obs, eps, tag, save=False, deterministic=False, want_logp=True):
        """Fused squashed-Gaussian policy: MLP kernel -> (mean, log-std heads) -> Gaussian head kernel (the env's own one
        when the box depends on the state).  Returns the clipped basic actions [n * P], log pi [n] and the raw heads."""
        f, P = self.fused, self.kernels.partial_dim
        n = obs.shape[0]
        raw = f.forward("actor", obs, None, f.buf(tag + ".raw", n, 2 * P), save=save)
        ap = f.buf(tag + ".ap", n * P)
        logp = f.buf(tag + ".logp", n) if want_logp else None
        if self._box_affine is None:
            self.kernels.gauss_head(obs, raw, eps.view(-1), deterministic, ap, logp)
        else:
            scale, base = self._box_affine
            self.backend.gauss_head(raw, eps.view(-1), scale, base, self._box_lo, self._box_hi, deterministic, ap, logp)
        return ap, logp, raw

    def _policy_partial(self, obs, warm):
        if warm:
            return None, hip_ops.NOISE_UNIFORM
        eps = self._draw(self._noise_n, self.vec.env_id_base * self.kernels.partial_dim, 0, rollout=True)
        if self.fused is not None:
            return self._gauss(obs, eps, "roll", want_logp=False)[0], hip_ops.NOISE_NONE     # already clipped
        ap, _, _ = self.agent.actor(obs, eps=eps)                   # rsample of the squashed Gaussian
        return ap.reshape(-1), hip_ops.NOISE_CLIP_ONLY              # the box clip happens inside the kernel

    def _eval_partial(self, obs):
        if self.fused is not None:
            zeros = self.fused.buf("eval.eps", obs.shape[0] * self.kernels.partial_dim)
            return self._gauss(obs, zeros, "eval", deterministic=True, want_logp=False)[0]
        return self.agent.actor(obs)[2].reshape(-1)                 # the mean action (deterministic=True)

    # ---- fused critic-forward pipeline (CartSafe kernels provide it) ------------------------------------------------
    @property
    def _pipelines(self):
        k = self.kernels
        return (self.fused is not None and (hasattr(k, "sac_critic_forward") or hasattr(k, "sac_critic_front"))
                and "critic1" in self.fused.descs and "actor" in self.fused.descs and self.schedule["fused_critic"] and not self._large_batch)

    def _sample(self):
        if self._pipelines:
            # the critic-forward pipeline draws and gathers the batch itself (into self._batch)
            c = self.buffer.split(self._batch)
            return c["state"], c["action"], c["next_state"], c["reward"], c["done"], c["ineq_viol"], c["eq_viol"]
        return super()._sample()

    def _critic_update_pipeline(self, cols):
        f, ag, B, buf = self.fused, self.agent, self.batch_size, self.buffer
        state, action = cols[0], cols[1]
        d = f.descs["critic1"]
        scale, base = self._box_affine
        inject = self._idx_inject is not None                  # tests replay the reference's draws
        idx_in = self._idx_inject() if inject else None
        eps_in = self._draw(self._noise_b, self.dist.rank * B * self.kernels.partial_dim, _SALT_CRITIC).view(-1) if inject else None
        q1, q2, qn1, qn2 = f.buf("q1", B, 1), f.buf("q2", B, 1), f.buf("qn1", B, 1), f.buf("qn2", B, 1)
        saves = (f.buf("critic1.x0", B, d.ein), f.buf("critic1.h1", B, d.H), f.buf("critic2.x0", B, d.ein),
                 f.buf("critic2.h1", B, d.H))
        logp = f.buf("crit.logp", B)
        if not hasattr(self.kernels, "sac_critic_forward"):
            # SpringPendulum: the chain is cut at the batch-coupled projection (front | project | back)
            ap = f.buf("crit.ap", B)
            self.kernels.sac_critic_front(f.descs["actor"], scale, base, self._box_lo, self._box_hi, buf.rows, buf.capacity,
                                          buf.n_envs, self._batch, None, idx_in, eps_in, buf.seed, 0, self.seed,
                                          self.dist.rank * B, _SALT_CRITIC, self._uctrl, ap, logp)
            next_actions = self._project_batch(cols[2], ap)
            self.kernels.sac_critic_back(f.descs["critic_target1"], f.descs["critic_target2"], d, f.descs["critic2"],
                                         self._batch, next_actions, q1, q2, qn1, qn2, *saves)
        else:
            self.kernels.sac_critic_forward(
                f.descs["actor"], f.descs["critic_target1"], f.descs["critic_target2"], d, f.descs["critic2"], scale, base,
                buf.rows, buf.capacity, buf.n_envs, self._batch, None, idx_in, eps_in, buf.seed, 0, self.seed,
                self.dist.rank * B, _SALT_CRITIC, self._uctrl, self.max_steps, self.corr_lr, self.corr_eps, self.corr_momentum,
                self._box_lo, self._box_hi, q1, q2, qn1, qn2, logp, *saves)
        self._critic_backward_td(cols, q1, q2, qn1, qn2, logp)

    def _critic_backward_td(self, cols, q1, q2, qn1, qn2, logp):
        """Backward of the twin critics with the TD target / Huber terms as its prologue (rpo_td; one loss-partial row
        per critic): with the pipelines, the four chains of a tile ran in separate workgroups and meet here."""
        f, ag, B = self.fused, self.agent, self.batch_size
        parts = f.buf("loss_parts2", 2, (B + 15) // 16)
        tds = [self.backend.Td(q.view(-1), qn1.view(-1), qn2.view(-1), logp, cols[3], cols[4], float(ag.alpha), ag.gamma,
                               f.buf(name, B, 1).view(-1), parts[i]) for i, (q, name) in enumerate(((q1, "dq1"), (q2, "dq2")))]
        self._zero_grads()
        self._gradmax_ready = f.backward_pair("critic1", "critic2", cols[0], cols[1], None, None,
                                              gradmax=self._critic_gradmax(), td1=tds[0], td2=tds[1])
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
        eps = self._draw(self._noise_b, self.dist.rank * B * self.kernels.partial_dim, _SALT_CRITIC)
        ap_next, logp, _ = self._gauss(next_state, eps, "crit")
        next_actions = self._project_batch(next_state, ap_next)
        qn1, qn2, q1, q2 = f.forward_multi([("critic_target1", next_state, next_actions, f.buf("qn1", B, 1), False),
                                            ("critic_target2", next_state, next_actions, f.buf("qn2", B, 1), False),
                                            ("critic1", state, action, f.buf("q1", B, 1), True),
                                            ("critic2", state, action, f.buf("q2", B, 1), True)])
        self._critic_backward_td(cols, q1, q2, qn1, qn2, logp)

    @property
    def _actor_pipeline(self):
        d = self.fused.descs if self.fused is not None else {}
        return (hasattr(self.backend, "sac_actor_forward") and "actor" in d and "critic1" in d and d["actor"].E == 128
                and d["critic1"].E == 128 and not d["critic1"].cat and self._box_affine is not None
                and self.kernels.partial_dim == 1 and self.kernels.action_dim == 2 and self.schedule["fused_actor"]
                and not self._large_batch)

    def _actor_update_pipeline(self, cols):
        """The policy step in two launches + the actor's weights pass (fused.hip)."""
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        da_, d1, d2 = f.descs["actor"], f.descs["critic1"], f.descs["critic2"]
        scale, base = self._box_affine
        b = f.buf
        parts = b("actor.parts", (B + 15) // 16, 8)
        raw, noise, logp = b("pi.raw", B, 2), b("pi.noise", B), b("pi.logp", B)
        actions, g_act = b("act_pi", B, 2), b("g_act", B, 2)
        dq1, dq2 = b("dq1_pi", B, 1), b("dq2_pi", B, 1)
        saved = (b("actor.x0", B, da_.ein), b("actor.h1", B, da_.H), b("critic1.x0", B, d1.ein), b("critic1.h1", B, d1.H),
                 b("critic2.x0", B, d2.ein), b("critic2.h1", B, d2.H))
        noise_in = self._draw(self._noise_b, self.dist.rank * B * self.kernels.partial_dim, _SALT_ACTOR).view(-1) if self._idx_inject is not None else None
        alpha = float(ag.alpha)
        self.backend.sac_actor_forward(k, da_, d1, d2, scale, base, self._box_lo, self._box_hi, alpha, self._batch, noise_in,
                                       self.seed, self.dist.rank * B, _SALT_ACTOR, self._uctrl, ag.nju.weight.view(-1),
                                       raw, noise, logp, actions, dq1, dq2, g_act, parts, saved)
        self._zero_grads()
        opt = ag.actor_optim
        fuse_max = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf")
        scratch = (b("actor.dh", B, da_.H), b("actor.dx0", B, da_.ein), b("critic1.dh", B, d1.H), b("critic1.dx0", B, d1.ein),
                   b("critic2.dh", B, d2.H), b("critic2.dx0", B, d2.ein))
        lag = b("actor.lag", 2)
        self.backend.sac_actor_backward(k, da_, d1, d2, ag.flat.sizes[1] > 0, self._batch, actions, g_act, raw, noise, logp,
                                        dq1, dq2, alpha / B, self._box_lo, self._box_hi, scale, base, saved, scratch,
                                        b("da1", B, 2), b("da2", B, 2), b("draw", B, 2), parts, lag,
                                        ag.nju.weight.grad.view(-1), opt.gradmax if fuse_max else None)
        self._actor_gradmax_ready = bool(fuse_max)
        loss = _LazySumPair(lag, parts, B)
        self.last_losses["actor"] = loss
        return loss, logp.view(-1, 1)

    @property
    def alpha(self):
        """The tuned temperature exp(log_alpha) (rpo_sac.py:216) -- reported only: like the reference, the losses read
        the agent's fixed `alpha` (rpo_sac.py:331,347)."""
        ag = self.agent
        return ag.log_alpha.detach().exp() if ag.log_alpha is not None else ag.alpha

    def _actor_update(self, cols):
        out = self._actor_update_impl(cols)
        if self.automatic_entropy_tuning:
            # d/d log_alpha of -(log_alpha (log pi + H_target)).mean() (rpo_sac.py:210); written AFTER the backward (which
            # may have zeroed the flat gradient) and before the policy-step all-reduce
            ag = self.agent
            torch.neg(out[1].detach().mean() + ag.target_entropy, out=ag.log_alpha.grad.view(()))
        return out

    def _actor_update_impl(self, cols):
        if self.fused is None:
            return super()._actor_update(cols)
        if self._actor_pipeline:
            su = self._split_state()
            if su is not None:
                lag, parts, logp = self._actor_update_split(su)
                loss = _LazySumPair(lag, parts, self.batch_size)
                self.last_losses["actor"] = loss
                return loss, logp.view(-1, 1)
            return self._actor_update_pipeline(cols)
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        state = cols[0]
        pre, self._actor_pre = getattr(self, "_actor_pre", None), None
        if pre is None:
            pre = self._actor_prefix(cols)
        eps, ap, logp, raw, actions, lag, g_act, fa = pre
        # (their own buffers: the lazily reduced actor loss keeps references to them, and the next critic-only update writes
        #  Q(s, a_replay) into "q1" / "q2" -- ADVICE r03)
        q1 = f.forward("critic1", state, actions, f.buf("q1_pi", B, 1), save=True)
        q2 = f.forward("critic2", state, actions, f.buf("q2_pi", B, 1), save=True)
        # d(-min(q1, q2))/dq: the smaller one takes the gradient, ties are split (torch.min's backward)
        dq1, dq2 = f.buf("dq1", B, 1), f.buf("dq2", B, 1)
        if hasattr(self.backend, "min_q_bwd"):
            self.backend.min_q_bwd(q1, q2, -1.0 / B, dq1, dq2)
        else:
            w1 = (q1 < q2).to(torch.float32) + 0.5 * (q1 == q2).to(torch.float32)
            torch.mul(w1, -1.0 / B, out=dq1)
            torch.mul(1.0 - w1, -1.0 / B, out=dq2)
        da1, da2 = f.buf("da1", B, k.action_dim), f.buf("da2", B, k.action_dim)
        shared = ag.flat.sizes[1] > 0
        f.backward_pair("critic1", "critic2", state, actions, dq1, dq2, da1, da2, param_grads=shared,
                        first_layer_state_only=True)
        P = k.partial_dim
        dap, draw = f.buf("dap", B * P), f.buf("draw", B, 2 * P)
        if fa:
            k.complete_bwd(state, da1, dap, action=actions, grad_action_b=da2, grad_action2=g_act)
        else:
            da1.add_(da2).add_(g_act)
            k.complete_bwd(state, da1, dap, action=actions)
        if self._box_affine is None:
            k.gauss_head_bwd(state, raw, eps.view(-1), dap, float(ag.alpha) / B, draw)
        else:
            scale, base = self._box_affine
            self.backend.gauss_head_bwd(raw, eps.view(-1), dap, float(ag.alpha) / B, scale, base, self._box_lo,
                                        self._box_hi, draw)
        # (no shared embedding: every gradient of the actor's slice is written by this backward -- it leaves the inf-norm for
        #  clip_grad_norm_, no rpo_absmax launch)
        opt = ag.actor_optim
        fuse_max = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf") and not shared
        f.backward("actor", state, None, draw, gradmax=opt.gradmax if fuse_max else None)
        self._actor_gradmax_ready = bool(fuse_max)
        loss = _LazySacLoss(lag, float(ag.alpha), logp, q1, q2)   # reduced only when somebody looks (five launches otherwise)
        self.last_losses["actor"] = loss
        return loss, logp.view(-1, 1)

    def _actor_prefix(self, cols):
        """The part of the policy step that depends on the ACTOR alone (see RPODDPG._actor_prefix): the rsample draw -- into a
        buffer of its own, the critic update draws into `_noise_b` --, the Gaussian head with saved activations, Complete, the
        Lagrangian term."""
        f, ag, B, k = self.fused, self.agent, self.batch_size, self.kernels
        state = cols[0]
        if getattr(self, "_noise_pi", None) is None:
            self._noise_pi = torch.zeros_like(self._noise_b)
        eps = self._draw(self._noise_pi, self.dist.rank * B * self.kernels.partial_dim, _SALT_ACTOR)
        ap, logp, raw = self._gauss(state, eps, "pi", save=True)
        actions = self._complete_only(state, ap)
        lag, g_act = f.buf("loss_lag", 1), f.buf("g_act", B, k.action_dim)
        fa = bool(getattr(k, "fused_adds", False))             # (EVOPF kernels: see RPODDPG._actor_update)
        if not fa:
            lag.zero_()
        self._zero_grads()                 # parameters AND multipliers (they live in the same flat buffer)
        k.lagrangian(actions, ag.nju.weight.view(-1), 1.0 / B, lag, g_act, ag.nju.weight.grad.view(-1), obs=state,
                     **(dict(overwrite=True) if fa else {}))
        return eps, ap, logp, raw, actions, lag, g_act, fa

    # ---- losses ---------------------------------------------------------------------------------------------
    def critic_loss(self, state, action, next_state, done, reward, ineq_viol=None, eq_viol=None):
        """y = r + gamma (1-d) (min Q_targ(s', a') - alpha log pi(a'|s')), a' ~ pi(s') projected; loss = sum of the
        two Huber terms (rpo_sac.py:342-353)."""
        ag = self.agent
        with torch.no_grad():
            eps = self._draw(self._noise_b, self.dist.rank * self.batch_size * self.kernels.partial_dim, _SALT_CRITIC)
            next_partial, logp = ag.take_action(next_state, log_pi=True, eps=eps)
            next_actions = self.process_action(next_state, next_partial)
            nq1, nq2 = ag.critic_target(next_state, next_actions)
        q1, q2 = ag.critic(state, action)
        return _TDHuberFn.apply(self.backend, ag.gamma, float(ag.alpha), reward, done, nq1, nq2, logp, q1, q2)

    def actor_loss(self, state):
        """mean(alpha log pi - min Q(s, Complete(a)) + nu . relu(g)) (rpo_sac.py:321-339) -> (loss, log_pi)."""
        ag = self.agent
        eps = self._draw(self._noise_b, self.dist.rank * self.batch_size * self.kernels.partial_dim, _SALT_ACTOR)
        ap, logp = ag.take_action(state, log_pi=True, eps=eps)
        actions = self.base_env.complete_partial(state, ap)
        q1, q2 = ag.critic(state, actions)
        loss = (ag.alpha * logp - torch.min(q1, q2)).mean()
        return loss + _LagrangianFn.apply(self.kernels, actions, ag.nju.weight, state), logp

    # ---- optimiser steps (rpo_sac.py:181-219) ---------------------------------------------------------------
    def _critic_step(self, actor_step):
        # soft_update of the critics on every step (rpo_sac.py:219) inside the Adam launch that produces them; only a
        # shared state embedding, which the actor's step also moves, forces the separate launch after that step
        ag = self.agent
        self._fused_polyak = ag.flat.sizes[1] == 0
        ready, self._gradmax_ready = self._gradmax_ready, False
        prepared, self._critic_prepared = getattr(self, "_critic_prepared", False), False
        if self._fused_polyak:
            ag.critic_optim.step(target=ag.critic_target_flat, tau=ag.tau, gradmax_ready=ready, clock=self._clock(not actor_step),
                                 prepared=prepared)
            return
        ag.critic_optim.step(gradmax_ready=ready, clock=self._clock(not actor_step), prepared=prepared)
        if not actor_step:
            ag.soft_update()

    def _actor_step(self, actor_out):
        ag = self.agent
        segs = [ag.actor_optim.segment(gradmax_ready=getattr(self, "_actor_gradmax_ready", False))]
        self._actor_gradmax_ready = False
        if not self.fixed:
            segs.append(ag.nju_optim.segment())
        if self.automatic_entropy_tuning:
            segs.append(ag.alpha_optim.segment())               # rpo_sac.py:210-216, gradient left by _actor_update
        prepared, self._actor_prepared = getattr(self, "_actor_prepared", False), False
        FusedAdam.step_many(self.backend, segs, clock=self._clock(True), prepared=prepared)   # actor Adam | DualAdam (| log_alpha): one launch
        if not self._fused_polyak:
            ag.soft_update()
