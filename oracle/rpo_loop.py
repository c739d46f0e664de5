"""Oracle for the trainer loop: a single-env, per-step CPU restatement of ``RPODDPG`` / ``RPOSAC``
(reference: rpo/algo/rpo_ddpg.py:79-205, rpo_sac.py:83-219, agents agent/ddpg_pa.py, agent/sac_pa.py).
TEST INFRASTRUCTURE ONLY: it is the checker of tests/ and the timed "port" of bench.py's ``cpu_baseline`` leg.

One env, one Python iteration per env step, numpy ring buffer, float64 env dynamics (oracle.cartsafe / .pendulum),
torch-CPU float32 MLPs with autograd, ``torch.optim.Adam`` -- the reference's cadence and arithmetic, written against
the oracle's own env functions.  Networks are plain functions of a parameter dict that uses the reference's
``state_dict`` names, so a reference checkpoint drops in.  Every random draw can be injected (``noise_fn``,
``index_fn``) so that fixtures captured from the reference can be replayed exactly.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cartsafe as cs
from . import pendulum as pd
from .replay import ReplayBuffer

LOG_SIG_MAX, LOG_SIG_MIN = -2, -23           # model/policy.py:6-7


# ------------------------------------------------------------------------------------------------ env adapters
class CartAdapter(object):
    name, state_dim, action_dim, eq_num, ineq_num = "CartSafe-v0", 6, 2, 1, 6
    box = (-10.0, 10.0)

    def __init__(self, partial=1, seed=None):
        self.c = cs.Constants(partial)
        self.partial, self.other = self.c.partial, self.c.other
        self.rng = np.random.RandomState(seed)
        self.state = None
        self.G = torch.tensor(self.c.G)
        self.d = torch.tensor(self.c.d)
        self.k = float(-(self.c.C_p * self.c.C_o_inv)[0, 0])       # d a_other / d a_partial

    def reset(self):
        self.state = cs.reset(self.rng, 1)
        self.len = 0
        return self.state[0].copy()

    def step(self, action):
        nxt, rew, term, ineq, eq = cs.step(self.state, action[None, :], self.c)
        self.state = nxt
        self.len += 1
        done = bool(term[0]) or self.len >= cs.MAX_EPISODE_STEPS       # gym TimeLimit(200)
        return nxt[0].copy(), float(rew[0]), done, ineq[0], eq[0]

    def complete_t(self, state, ap):                                  # differentiable, cartpole.py:369-373
        other = ap * self.k
        return torch.cat([ap, other], dim=1) if self.partial == 0 else torch.cat([other, ap], dim=1)

    def ineq_dist_t(self, state, action):
        return torch.clamp(action @ self.G.T - self.d, 0)

    def project(self, state, action, lr, steps, corr_eps, momentum):
        return cs.grad_steps(action, self.c, lr, steps, corr_eps, momentum)


class PendulumAdapter(object):
    name, state_dim, action_dim, eq_num, ineq_num = "SpringPendulum-v0", 5, 2, 1, 1
    box = (-6.0, 6.0)
    partial, other = 0, 1

    def __init__(self, seed=None, reference_batch_semantics=False):
        self.rng = np.random.RandomState(seed)
        self.internal = None
        # True: replay the reference's literal batched projection (batch-global stop test, cross-coupled
        # ineq_partial_grad) -- only used to match fixtures of the reference's B=256 critic_loss
        self.reference_batch_semantics = reference_batch_semantics

    def reset(self):
        self.internal = pd.reset(self.rng, 1)
        self.len = 0
        return pd.get_obs(self.internal)[0]

    def step(self, action):
        nxt, obs, rew, term, ineq, eq = pd.step(self.internal, action[None, :])
        self.internal = nxt
        self.len += 1
        done = bool(term[0]) or self.len >= pd.MAX_EPISODE_STEPS
        return obs[0], float(rew[0]), done, ineq[0], eq[0]

    def complete_t(self, state, ap):                                  # pendulum.py:256-262 with set_eq :264-288
        cos_t, sin_t, thdot, l, ldot = [state[:, i:i + 1] for i in range(5)]
        b = -pd.M_DT * ldot - (l * pd.M * thdot ** 2 - pd.K * (l - pd.L0) - pd.M * pd.G * cos_t)
        return torch.cat([ap, (b - ap * sin_t) * (1.0 / cos_t)], dim=1)

    def ineq_dist_t(self, state, action):
        return torch.clamp((action * action).sum(dim=1, keepdim=True) - pd.MAX_SUMMATION, 0)

    def project(self, state, action, lr, steps, corr_eps, momentum):
        ref = self.reference_batch_semantics
        return pd.grad_steps(np.asarray(state, dtype=np.float32), action, lr, steps, corr_eps, momentum,
                             batch_global_stop=ref, batched_reference=ref)


# ------------------------------------------------------------------------------------------------ networks
def _linear_init(gen_out, gen_in):
    lin = torch.nn.Linear(gen_in, gen_out)        # the reference's initialiser, drawn from torch's global generator
    return lin.weight.detach().clone().requires_grad_(), lin.bias.detach().clone().requires_grad_()


def _embed(p, prefix, x):
    return F.linear(x, p[prefix + ".embeds.0.weight"], p[prefix + ".embeds.0.bias"])      # embedding.py:21-29


def _head(p, prefix, x, n=2):
    for i in range(n):
        x = F.linear(F.relu(x), p["%s.%d.weight" % (prefix, i)], p["%s.%d.bias" % (prefix, i)])
    return x


class Nets(object):
    """Parameter dicts named like the reference's modules; ``shared`` ties the state embedding of actor and critic
    (agent/ddpg_pa.py:34-36)."""

    def __init__(self, env, embed_dim, hidden_dim, shared, sac):
        S, A = env.state_dim, env.action_dim
        self.sac, self.shared = sac, shared
        lo, hi = env.box
        self.scale, self.base = (hi - lo) / 2, lo + (hi - lo) / 2      # BoxConstraint tanh style, model/utils.py:19-21
        self.lo, self.hi = lo, hi
        a, c = {}, {}
        # construction order of agent/ddpg_pa.py:32-49 / agent/sac_pa.py:32-52
        se_w, se_b = _linear_init(embed_dim, S)
        if not sac:
            ae = _linear_init(embed_dim, A)
            sv = (se_w, se_b) if shared else _linear_init(embed_dim, S)
            a["state_embed.embeds.0.weight"], a["state_embed.embeds.0.bias"] = se_w, se_b
            a["affines.0.weight"], a["affines.0.bias"] = _linear_init(hidden_dim, embed_dim)
            a["affines.1.weight"], a["affines.1.bias"] = _linear_init(1, hidden_dim)
            c["state_embed.embeds.0.weight"], c["state_embed.embeds.0.bias"] = sv
            c["action_embed.embeds.0.weight"], c["action_embed.embeds.0.bias"] = ae
            c["affines.0.weight"], c["affines.0.bias"] = _linear_init(hidden_dim, embed_dim)
            c["affines.1.weight"], c["affines.1.bias"] = _linear_init(1, hidden_dim)
        else:
            ae1, ae2 = _linear_init(embed_dim, A), _linear_init(embed_dim, A)
            sv1 = (se_w, se_b) if shared else _linear_init(embed_dim, S)
            sv2 = (se_w, se_b) if shared else _linear_init(embed_dim, S)
            a["state_embed.embeds.0.weight"], a["state_embed.embeds.0.bias"] = se_w, se_b
            a["affines.0.weight"], a["affines.0.bias"] = _linear_init(hidden_dim, embed_dim)
            a["affine_mean.weight"], a["affine_mean.bias"] = _linear_init(1, hidden_dim)
            a["affine_log_std.weight"], a["affine_log_std.bias"] = _linear_init(1, hidden_dim)
            for i, (sv, ae) in enumerate(((sv1, ae1), (sv2, ae2)), 1):
                c["state_embed%d.embeds.0.weight" % i], c["state_embed%d.embeds.0.bias" % i] = sv
                c["action_embed%d.embeds.0.weight" % i], c["action_embed%d.embeds.0.bias" % i] = ae
            w10, w20 = _linear_init(hidden_dim, embed_dim), _linear_init(hidden_dim, embed_dim)
            w11, w21 = _linear_init(1, hidden_dim), _linear_init(1, hidden_dim)
            c["affines1.0.weight"], c["affines1.0.bias"] = w10
            c["affines2.0.weight"], c["affines2.0.bias"] = w20
            c["affines1.1.weight"], c["affines1.1.bias"] = w11
            c["affines2.1.weight"], c["affines2.1.bias"] = w21
        self.actor, self.critic = a, c
        self.actor_target = {k: v.detach().clone() for k, v in a.items()} if not sac else None
        self.critic_target = {k: v.detach().clone() for k, v in c.items()}

    @staticmethod
    def unique(params):
        seen, out = set(), []
        for v in params.values():
            if id(v) not in seen:
                seen.add(id(v))
                out.append(v)
        return out

    def load(self, actor_sd, critic_sd):
        with torch.no_grad():
            for d, sd in ((self.actor, actor_sd), (self.critic, critic_sd)):
                for k in d:
                    d[k].copy_(torch.as_tensor(sd[k]))
            if self.actor_target is not None:
                for k in self.actor:
                    self.actor_target[k].copy_(self.actor[k])
            for k in self.critic:
                self.critic_target[k].copy_(self.critic[k])

    # deterministic actor, model/policy.py:24-33
    def pi(self, s, target=False):
        p = self.actor_target if target else self.actor
        x = _head(p, "affines", _embed(p, "state_embed", s))
        return self.scale * torch.tanh(x) + self.base

    # squashed Gaussian actor, model/policy.py:48-71 -> (action, log_prob, mean_action)
    def pi_gauss(self, s, eps):
        p = self.actor
        x = F.relu(_head(p, "affines", _embed(p, "state_embed", s), n=1))
        mean = F.linear(x, p["affine_mean.weight"], p["affine_mean.bias"])
        log_std = torch.clamp(F.linear(x, p["affine_log_std.weight"], p["affine_log_std.bias"]) - 3, LOG_SIG_MIN, LOG_SIG_MAX)
        std = log_std.exp()
        x = mean + eps * std
        logp = -((x - mean) ** 2) / (2 * std ** 2) - std.log() - math.log(math.sqrt(2 * math.pi))
        y = torch.tanh(x)
        logp = logp - torch.log(self.scale * (1 - y.pow(2)) + 1e-6)
        return self.scale * y + self.base, logp.sum(1, keepdim=True), self.scale * torch.tanh(mean) + self.base

    def q(self, s, a, target=False):
        p = self.critic_target if target else self.critic
        if not self.sac:
            return _head(p, "affines", _embed(p, "state_embed", s) + _embed(p, "action_embed", a))   # value.py:51-59
        return tuple(_head(p, "affines%d" % i, _embed(p, "state_embed%d" % i, s) + _embed(p, "action_embed%d" % i, a))
                     for i in (1, 2))                                                                 # value.py:125-140


# ------------------------------------------------------------------------------------------------ trainer
class OracleRPO(object):

    def __init__(self, env, sac=False, alpha=0.2, max_steps=10, embed_dim=256, hidden_dim=256, shared_param=True,
                 lr_actor=1e-4, lr_critic=3e-4, lr_dual=1e-4, eps=0.1, eps_start=1.0, eps_epoch=10000, tau=0.005,
                 gamma=0.95, capacity=10000, warmup=1000, corr_lr=1e-5, eval_lr=1e-5, corr_eps=1e-5, corr_momentum=0.5,
                 batch_size=256, policy_fre=2, eval_steps=None, init_nju=0.0, clip_thres=float("inf"),
                 noise_fn=None, index_fn=None):
        self.env, self.sac, self.alpha = env, sac, alpha
        self.nets = Nets(env, embed_dim, hidden_dim, shared_param, sac)
        self.actor_optim = torch.optim.Adam(Nets.unique(self.nets.actor), lr=lr_actor)
        self.critic_optim = torch.optim.Adam(Nets.unique(self.nets.critic), lr=lr_critic)
        self.nju = torch.full((1, env.ineq_num), float(init_nju), requires_grad=True)     # Dual, model/dual.py:47-65
        self.nju_optim = torch.optim.Adam([self.nju], lr=lr_dual, maximize=True)          # DualAdam :27-45
        self.buffer = ReplayBuffer(capacity, env.state_dim, env.action_dim, env.eq_num, env.ineq_num)
        self.max_steps, self.corr_lr, self.eval_lr, self.corr_eps, self.corr_momentum = \
            max_steps, corr_lr, eval_lr, corr_eps, corr_momentum
        self.eval_steps = max_steps if eval_steps is None else eval_steps
        self.batch_size, self.policy_fre, self.warmup, self.tau, self.gamma = batch_size, policy_fre, warmup, tau, gamma
        self.clip_thres = clip_thres
        self.eps_now, self.eps_end, self.decay = eps_start, eps, (eps_start - eps) / eps_epoch
        self.noise_fn = noise_fn or (lambda shape, tag: torch.randn(shape))
        self.index_fn = index_fn or (lambda size, num: np.random.randint(0, size, size=num))
        self.t, self.viol_steps, self.returns = 0, 0, []
        self.state = None

    # ---- action selection + projection (rpo_ddpg.py:72-77,98-109) --------------------------------------------
    def _partial(self, s, tag, deterministic=False, target=False):
        if self.sac:
            ap, logp, mean = self.nets.pi_gauss(s, self.noise_fn((s.shape[0], 1), tag))
            ap = mean if deterministic else ap
            return torch.clamp(ap, self.nets.lo, self.nets.hi), logp
        ap = self.nets.pi(s, target=target)
        if not deterministic:
            ap = torch.clamp(ap + self.eps_now * self.noise_fn((s.shape[0], 1), tag), self.nets.lo, self.nets.hi)
        return ap, None

    def process_action(self, s, ap, train=True):
        a0 = self.env.complete_t(s, ap).detach().numpy()
        lr, steps = (self.corr_lr, self.max_steps) if train else (self.eval_lr, self.eval_steps)
        a, it = self.env.project(s.numpy(), a0, lr, steps, self.corr_eps, self.corr_momentum)
        return torch.tensor(a), it

    # ---- losses ---------------------------------------------------------------------------------------------
    def critic_loss(self, b):
        n = self.nets
        with torch.no_grad():
            if self.sac:
                ap, logp = self._partial(b["next_state"], "critic")
                qn1, qn2 = n.q(b["next_state"], self.process_action(b["next_state"], ap)[0], target=True)
                nq = torch.min(qn1, qn2) - self.alpha * logp
            else:
                ap, _ = self._partial(b["next_state"], "critic", deterministic=True, target=True)
                nq = n.q(b["next_state"], self.process_action(b["next_state"], ap)[0], target=True)
            y = b["reward"] + self.gamma * (1 - b["done"]) * nq
        if self.sac:
            q1, q2 = n.q(b["state"], b["action"])
            return F.smooth_l1_loss(q1, y) + F.smooth_l1_loss(q2, y)
        return F.smooth_l1_loss(n.q(b["state"], b["action"]), y)

    def actor_loss(self, b):
        n = self.nets
        ap, logp = self._partial(b["state"], "actor")
        actions = self.env.complete_t(b["state"], ap)
        lag = F.linear(self.env.ineq_dist_t(b["state"], actions), self.nju)
        if self.sac:
            q1, q2 = n.q(b["state"], actions)
            return (self.alpha * logp - torch.min(q1, q2) + lag).mean()
        return (-n.q(b["state"], actions) + lag).mean()

    # ---- one update (rpo_ddpg.py:163-205 / rpo_sac.py:167-219) ------------------------------------------------
    def train(self, t, batch=None):
        n = self.nets
        if batch is None:
            batch = self.buffer.sample(self.batch_size, self.index_fn(self.buffer.size, self.batch_size))
        b = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in batch.items()}
        out = {}
        loss = self.critic_loss(b)
        self.critic_optim.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(Nets.unique(n.critic), self.clip_thres, float("inf"))
        self.critic_optim.step()
        out["critic_loss"] = float(loss.detach())
        if t % self.policy_fre == 0:
            loss = self.actor_loss(b)
            self.actor_optim.zero_grad()
            self.nju_optim.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(Nets.unique(n.actor), self.clip_thres, float("inf"))
            self.actor_optim.step()
            self.nju_optim.step()
            with torch.no_grad():
                self.nju.clamp_(0)
            out["actor_loss"] = float(loss.detach())
            if not self.sac:
                self._polyak(n.actor_target, n.actor)
                self._polyak(n.critic_target, n.critic)
        if self.sac:
            self._polyak(n.critic_target, n.critic)
        return out

    def _polyak(self, target, source):
        with torch.no_grad():
            for k in target:
                target[k].copy_(target[k] * (1.0 - self.tau) + source[k] * self.tau)

    # ---- rollout + update, one env step per iteration (rpo_ddpg.py:91-161) ----------------------------------
    def run(self, steps, train=True):
        if self.state is None:
            self.state = self.env.reset()
            self.ep_ret = 0.0
        for _ in range(int(steps)):
            s = torch.tensor(self.state, dtype=torch.float32).unsqueeze(0)
            with torch.no_grad():
                if self.t < self.warmup:
                    u = 2 * torch.rand(1, 1) - 1
                    ap = self.nets.scale * u + self.nets.base
                else:
                    ap, _ = self._partial(s, "rollout")
                action = self.process_action(s, ap)[0].numpy()[0]
            self.eps_now = max(self.eps_end, self.eps_now - self.decay)
            nxt, reward, done, ineq, eq = self.env.step(action)
            self.buffer.add(state=self.state, action=action, next_state=nxt, reward=reward, done=done, eq_viol=eq,
                            ineq_viol=ineq)
            self.viol_steps += max(float(ineq.max()), float(np.abs(eq).max())) > 1e-3
            self.ep_ret += reward
            self.t += 1
            if done:
                self.returns.append(self.ep_ret)
                self.ep_ret = 0.0
                nxt = self.env.reset()
            self.state = nxt
            if train and self.t >= self.warmup:
                self.train(self.t)
        return self
