"""Flat float32 parameter / gradient buffers and the fused optimiser launches built on them.

The reference updates 6-12 small tensors with one ``torch.optim.Adam`` python loop per optimiser, a
``clip_grad_norm_`` pass and a Polyak loop (rpo_ddpg.py:178-205, agent/ddpg_pa.py:77-86).  Here every optimiser owns a
contiguous slice of ONE buffer, so clip + Adam (+ Polyak) is two kernel launches regardless of the number of tensors,
and a data-parallel all-reduce is one collective on one bucket.

Layout: ``[ critic-only | shared | actor-only ]`` (each segment padded to 4 floats for 16-byte alignment).  With
``shared_param=True`` the state embedding belongs to both optimisers, exactly like the reference where both Adams hold
the same Parameter objects (agent/ddpg_pa.py:34-36,52-53): critic Adam covers ``[0, c+s)``, actor Adam
``[c, c+s+a)``, each with its own moments and step counter.
"""
import torch


def _pad4(n):
    return (n + 3) // 4 * 4


def _unique(params):
    seen, out = set(), []
    for p in params:
        if id(p) not in seen:
            seen.add(id(p))
            out.append(p)
    return out


class FlatParams(object):

    def __init__(self, critic, actor, device, extras=()):
        cp, ap = _unique(critic.parameters()), _unique(actor.parameters())
        a_ids, c_ids = set(id(p) for p in ap), set(id(p) for p in cp)
        seg_c = [p for p in cp if id(p) not in a_ids]
        seg_s = [p for p in cp if id(p) in a_ids]
        seg_a = [p for p in ap if id(p) not in c_ids]
        self.sizes = [sum(p.numel() for p in seg) for seg in (seg_c, seg_s, seg_a)]
        # every tensor starts on a 16-byte boundary (float4 loads in the MLP kernels); padding floats stay zero
        c, s, a = [sum(_pad4(p.numel()) for p in seg) for seg in (seg_c, seg_s, seg_a)]
        x = sum(_pad4(p.numel()) for p in extras)
        self.critic_range = (0, c + s)
        self.actor_range = (c, c + s + a)
        # the multipliers (`extras`) sit right behind the actor slice: on a policy step their gradient travels in the
        # same all-reduce as the actor's (one bucket), while every optimiser still steps only its own slice
        self.policy_bucket = (c, c + s + a + x)
        self.total = c + s + a + x
        self.data = torch.zeros(self.total, device=device)
        self.grad = torch.zeros(self.total, device=device)
        self.offset = {}
        for start, seg in ((0, seg_c), (c, seg_s), (c + s, seg_a), (c + s + a, list(extras))):
            off = start
            for p in seg:
                n = p.numel()
                self.data[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.data[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                self.offset[id(p)] = off
                off += _pad4(n)
        self.unique_numel = sum(self.sizes)

    def make_target(self, module, target_module, rng):
        """Re-home ``target_module`` (a deepcopy of ``module``) into a flat buffer laid out like ``data[rng]``."""
        lo, hi = rng
        flat = torch.zeros(hi - lo, device=self.data.device)
        for tp, p in zip(_unique(target_module.parameters()), _unique(module.parameters())):
            off = self.offset[id(p)] - lo
            n = p.numel()
            flat[off:off + n].copy_(tp.data.reshape(-1))
            tp.data = flat[off:off + n].view(tp.shape)
            tp.requires_grad_(False)
        return flat

    def param(self, rng):
        return self.data[rng[0]:rng[1]]

    def gradient(self, rng):
        return self.grad[rng[0]:rng[1]]


class FusedAdam(object):
    """clip_grad_norm_(inf) + Adam (+ clamp, + Polyak) on one flat slice: two launches (rpo_absmax, rpo_adam_step)."""

    def __init__(self, backend, param, grad, lr, weight_decay=0.0, clip_thres=0.0, maximize=False, clamp_min0=False,
                 betas=(0.9, 0.999), eps=1e-8):
        self.backend, self.param, self.grad = backend, param, grad
        self.lr, self.weight_decay, self.maximize, self.clamp_min0 = lr, weight_decay, maximize, clamp_min0
        self.betas, self.eps = betas, eps
        self.clip_thres = clip_thres
        dev = param.device
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        # {step, pad, arrival word (8 B), cached bias corrections of the next step (2 doubles), ..., 16 sub-counters}
        const = getattr(backend, "CONST", {})
        self.step_dev = torch.zeros(const.get("RPO_ADAM_STATE_LEN", 544), dtype=torch.int32, device=dev)
        # inf-norm of the gradient slice, in RPO_GRADMAX_SLOTS slots on separate cache lines (include/rpo_hip.h)
        self.gradmax = torch.zeros(const.get("RPO_GRADMAX_LEN", 512), device=dev)
        # True: the step leaves a zeroed gradient slice behind (optimizer.zero_grad() folded into the Adam launch); the
        # trainers switch it on when every backward of the iteration goes through the accumulating MLP kernels
        self.zero_grad_after = False

    def step(self, target=None, tau=0.0, gradmax_ready=False, clock=None, prepared=False):
        """``gradmax_ready``: the backward pass already left the inf-norm of this slice in ``self.gradmax``.
        ``clock``: device counter the launch advances when it has finished (the trainers' update clock).
        ``prepared``: the launch before this one advanced the step counter, left the bias corrections and (if due) advanced
        the clock (rpo_split_update.prep_step / clock_out); the next update's first launch zeroes ``gradmax``."""
        clip = self.clip_thres if self.clip_thres and self.clip_thres != float("inf") else 0.0
        if clip > 0 and not gradmax_ready:
            self.backend.absmax(self.grad, self.gradmax)
        kw = {} if clock is None else dict(clock=clock)
        if prepared:
            kw = dict(prepared=True)
        self.backend.adam_step(self.param, self.grad, self.exp_avg, self.exp_avg_sq, self.step_dev, self.lr,
                               self.betas[0], self.betas[1], self.eps, self.weight_decay, self.maximize, clip,
                               self.gradmax, True, self.clamp_min0, target, tau, zero_grad=self.zero_grad_after, **kw)

    def segment(self, target=None, tau=0.0, gradmax_ready=False, target2=None, n2=0):
        """This optimiser's step as one slice of ``step_many`` (the inf-norm launch, when needed, happens here)."""
        clip = self.clip_thres if self.clip_thres and self.clip_thres != float("inf") else 0.0
        if clip > 0 and not gradmax_ready:
            self.backend.absmax(self.grad, self.gradmax)
        return dict(param=self.param, grad=self.grad, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq,
                    step_dev=self.step_dev, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                    weight_decay=self.weight_decay, maximize=self.maximize, clip_thres=clip, gradmax=self.gradmax,
                    reset_gradmax=True, clamp_min0=self.clamp_min0, target=target, tau=tau, zero_grad=self.zero_grad_after,
                    target2=target2, n2=n2)

    @staticmethod
    def step_many(backend, segs, clock=None, prepared=False):
        """Several non-overlapping slices (``segment()`` dicts or ``dict(polyak_only=True, param=, target=, tau=)``) in
        one launch where the backend has rpo_adam_step_multi; one launch each otherwise.  ``clock`` rides on slice 0.
        ``prepared``: see ``step`` (every stepped slice; the clock was advanced by the preparing launch)."""
        if prepared:
            clock = None
        if hasattr(backend, "adam_step_multi") and 1 < len(segs) <= 4:
            kw = {} if clock is None else dict(clock=clock)
            if prepared:
                kw["prepared"] = True
            return backend.adam_step_multi(segs, **kw)
        for i, g in enumerate(segs):
            if g.get("polyak_only"):
                backend.polyak(g["param"], g["target"], g["tau"])
                continue
            g = dict(g)
            param, t2, n2 = g.pop("param"), g.pop("target2", None), g.pop("n2", 0)
            if prepared:
                g["prepared"] = True
            if clock is not None and i == 0:
                g["clock"] = clock
            backend.adam_step(param, g.pop("grad"), g.pop("exp_avg"), g.pop("exp_avg_sq"), g.pop("step_dev"), g.pop("lr"),
                              **g)
            if t2 is not None and n2 > 0:
                backend.polyak(param[:n2], t2, g["tau"])

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, step=self.step_dev, lr=self.lr)

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        step = sd["step"]
        self.step_dev.zero_()                                   # (older checkpoints hold 4 / 8 words: the cache starts empty)
        k = min(8, step.numel())
        self.step_dev[:k].copy_(step[:k])
