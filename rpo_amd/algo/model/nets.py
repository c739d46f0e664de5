"""Actor / critic networks of the RPO agents (PyTorch-ROCm: rocBLAS GEMMs + autograd).

Architectures, parameter names and construction order follow the reference so that (a) a reference ``state_dict``
loads unchanged and (b) the same ``torch.manual_seed`` produces bit-identical initial weights:
embeddings rpo/algo/model/embedding.py:6-54, policies model/policy.py:9-71, value heads model/value.py:5-140,
multipliers model/dual.py:47-65, action box model/utils.py:5-114.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

LOG_SIG_MAX = -2      # model/policy.py:6-7
LOG_SIG_MIN = -23


def _stack(in_dim, out_dim, hidden_dim, layers):
    """`layers` Linear modules in_dim -> ... -> out_dim (the reference's embed_layer == 1 collapses to one Linear)."""
    if layers == 1:
        return nn.ModuleList([nn.Linear(in_dim, out_dim)])
    mods = [nn.Linear(in_dim, hidden_dim)]
    for _ in range(layers - 2):
        mods.append(nn.Linear(hidden_dim, hidden_dim))
    mods.append(nn.Linear(hidden_dim, out_dim))
    return nn.ModuleList(mods)


class _Embedding(nn.Module):
    def __init__(self, in_dim, embed_dim, hidden_dim=256, embed_layer=1):
        super().__init__()
        self.embeds = _stack(in_dim, embed_dim, hidden_dim, embed_layer)

    def forward(self, x):
        x = self.embeds[0](x)
        for layer in self.embeds[1:]:
            x = layer(F.relu(x))
        return x


class StateEmbedding(_Embedding):
    pass


class ActionEmbedding(_Embedding):
    pass


class SharedEmbedding(_Embedding):
    pass


class BoxConstraint(object):
    """Affine map from tanh / sigmoid range onto [cmin, cmax], plus clip / sample (model/utils.py:5-114).
    ``volatile`` boxes are recomputed from the state through ``update(state, full=...)`` (EVOPF)."""

    def __init__(self, cmin, cmax, device, style="tanh", volatile=False, update=None, full=False, verbose=False):
        self.style, self.volatile, self.update, self.full, self.device = style, volatile, update, full, device
        self.cmin, self.cmax = np.asarray(cmin), np.asarray(cmax)
        self.scale, self.base = self._affine(self.cmin, self.cmax)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32).to(device)   # noqa: E731
        self.base_torch, self.scale_torch = f32(self.base), f32(self.scale)
        self.cmin_torch, self.cmax_torch = f32(self.cmin), f32(self.cmax)
        if verbose:
            print("cmax:", cmax, "cmin", cmin, "scale", self.scale, "base", self.base)

    def _affine(self, cmin, cmax):
        if self.style == "sigmoid":
            return cmax - cmin, cmin
        scale = (cmax - cmin) / 2
        return scale, cmin + scale

    def to(self, *args, **kwargs):
        for name in ("base_torch", "scale_torch", "cmin_torch", "cmax_torch"):
            setattr(self, name, getattr(self, name).to(*args, **kwargs))
        return self

    def cuda(self):
        return self.to("cuda")

    def update_box(self, state):
        self.cmin_vol, self.cmax_vol = self.update(state, full=self.full)
        self.scale_vol, self.base_vol = self._affine(self.cmin_vol, self.cmax_vol)
        f32 = lambda a: torch.as_tensor(a, dtype=torch.float32).to(self.device)   # noqa: E731
        self.base_torch_vol, self.scale_torch_vol = f32(self.base_vol), f32(self.scale_vol)
        self.cmin_torch_vol, self.cmax_torch_vol = f32(self.cmin_vol), f32(self.cmax_vol)

    def _pick(self, tensor_like, name):
        suffix = "_vol" if self.volatile else ""
        return getattr(self, name + ("_torch" if tensor_like else "") + suffix)

    def __call__(self, x, state=None):
        if self.volatile:
            self.update_box(state)
        is_t = isinstance(x, torch.Tensor)
        return self._pick(is_t, "scale") * x + self._pick(is_t, "base")

    def clip(self, x, state=None):
        if self.volatile:
            self.update_box(state)
        if isinstance(x, torch.Tensor):
            return torch.clip(x, self._pick(True, "cmin"), self._pick(True, "cmax"))
        return np.clip(x, self._pick(False, "cmin"), self._pick(False, "cmax"))

    def sample(self, state):
        u = torch.rand_like(self.base_torch)
        if self.style == "tanh":
            u = 2 * u - 1
        if self.volatile:
            self.update_box(state)
        return self._pick(True, "scale") * u + self._pick(True, "base")

    def sample_np(self, state):
        u = np.random.rand(*self.base.shape)
        if self.style == "tanh":
            u = 2 * u - 1
        if self.volatile:
            self.update_box(state)
        return self._pick(False, "scale") * u + self._pick(False, "base")

    def action_scale(self, state=None):
        if self.volatile:
            self.update_box(state)
        return self._pick(isinstance(state, torch.Tensor), "scale")


class SharedPolicy(nn.Module):
    """Deterministic actor: embed -> ReLU MLP -> tanh -> box (model/policy.py:9-33)."""

    def __init__(self, state_dim, action_dim, state_embed, embed_dim, hidden_dim=256, hidden_layer=1, box_constraint=None):
        super().__init__()
        self.box_constraint = box_constraint
        self.state_embed = state_embed
        dims = [embed_dim] + [hidden_dim] * hidden_layer + [action_dim]
        self.affines = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

    def forward(self, s):
        x = self.state_embed(s)
        for affine in self.affines:
            x = affine(F.relu(x))
        if self.box_constraint:
            x = self.box_constraint(torch.tanh(x), s)
        return x


class GaussianSharedPolicy(nn.Module):
    """Squashed-Gaussian actor (model/policy.py:35-71).  Returns (action, log_prob, mean_action).
    ``eps`` lets the caller supply the standard-normal draw of ``rsample`` (tests; Philox-driven rollouts)."""

    def __init__(self, state_dim, action_dim, state_embed, embed_dim, hidden_dim=256, hidden_layer=1, box_constraint=None):
        super().__init__()
        self.box_constraint = box_constraint
        self.state_embed = state_embed
        dims = [embed_dim] + [hidden_dim] * hidden_layer
        self.affines = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.affine_mean = nn.Linear(hidden_dim, action_dim)
        self.affine_log_std = nn.Linear(hidden_dim, action_dim)

    def forward(self, s, eps=None):
        x = self.state_embed(s)
        for affine in self.affines:
            x = affine(F.relu(x))
        x = F.relu(x)
        mean = self.affine_mean(x)
        log_std = torch.clamp(self.affine_log_std(x) - 3, min=LOG_SIG_MIN, max=LOG_SIG_MAX)
        std = log_std.exp()
        if eps is None:
            eps = torch.randn_like(mean)
        x = mean + eps * std                                              # Normal(mean, std).rsample()
        # Normal.log_prob(x) = -((x - mean)^2) / (2 var) - log_std - log(sqrt(2 pi))
        log_prob = -((x - mean) ** 2) / (2 * std * std) - log_std - 0.9189385332046727
        if self.box_constraint:
            y = torch.tanh(x)
            x = self.box_constraint(y, s)
            mean = self.box_constraint(torch.tanh(mean), s)
            log_prob = log_prob - torch.log(self.box_constraint.action_scale(s) * (1 - y.pow(2)) + 1e-6)
        else:
            log_prob = log_prob - torch.log(1 - x.pow(2) + 1e-6)
        return x, log_prob.sum(1, keepdim=True), mean


class _ValueBase(nn.Module):
    def __init__(self, partial, partial_idx):
        super().__init__()
        self.partial, self.partial_idx = partial, partial_idx

    def _a(self, a):
        return a[:, self.partial_idx] if self.partial else a

    @staticmethod
    def _head(in_dim, hidden_dim, hidden_layer):
        dims = [in_dim] + [hidden_dim] * hidden_layer + [1]
        return [nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]

    @staticmethod
    def _run(affines, x):
        for affine in affines:
            x = affine(F.relu(x))
        return x


class SharedValueAdd(_ValueBase):
    """Q(s,a) = MLP(relu(E_s s + E_a a)) (model/value.py:33-59)."""
    joint = staticmethod(lambda s, a: s + a)
    width = 1

    def __init__(self, state_dim, action_dim, state_embed, action_embed, embed_dim, hidden_dim=256, hidden_layer=1,
                 partial=False, partial_idx=None):
        super().__init__(partial, partial_idx)
        self.state_embed, self.action_embed = state_embed, action_embed
        self.affines = nn.ModuleList(self._head(embed_dim * self.width, hidden_dim, hidden_layer))

    def forward(self, s, a):
        return self._run(self.affines, self.joint(self.state_embed(s), self.action_embed(self._a(a))))


class SharedValueCat(SharedValueAdd):
    """Concatenating variant (model/value.py:5-31)."""
    joint = staticmethod(lambda s, a: torch.cat([s, a], dim=1))
    width = 2


class DoubleValueAdd(_ValueBase):
    """Twin critics (model/value.py:102-140); heads are created interleaved like the reference (init order)."""
    joint = staticmethod(lambda s, a: s + a)
    width = 1

    def __init__(self, state_dim, action_dim, state_embed1, state_embed2, action_embed1, action_embed2, embed_dim,
                 hidden_dim=256, hidden_layer=1, partial=False, partial_idx=None):
        super().__init__(partial, partial_idx)
        self.state_embed1, self.action_embed1 = state_embed1, action_embed1
        self.state_embed2, self.action_embed2 = state_embed2, action_embed2
        dims = [embed_dim * self.width] + [hidden_dim] * hidden_layer + [1]
        self.affines1, self.affines2 = nn.ModuleList(), nn.ModuleList()
        for a, b in zip(dims[:-1], dims[1:]):
            self.affines1.append(nn.Linear(a, b))
            self.affines2.append(nn.Linear(a, b))

    def forward(self, s, a):
        a = self._a(a)
        q1 = self._run(self.affines1, self.joint(self.state_embed1(s), self.action_embed1(a)))
        q2 = self._run(self.affines2, self.joint(self.state_embed2(s), self.action_embed2(a)))
        return q1, q2


class DoubleValueCat(DoubleValueAdd):
    joint = staticmethod(lambda s, a: torch.cat([s, a], dim=1))
    width = 2


class Dual(nn.Module):
    """Lagrange multipliers as a bias-free 1 x dim linear map: Dual(x) = x . weight^T (model/dual.py:47-65)."""

    def __init__(self, dim, device=None, dtype=None):
        super().__init__()
        self.dim = dim
        self.weight = nn.Parameter(torch.zeros((1, dim), device=device, dtype=dtype))

    def reset_parameters(self, value=0.0):
        nn.init.constant_(self.weight, value)

    def __repr__(self):
        return "Dual(weight: %s)" % (self.weight,)

    def forward(self, x):
        return F.linear(x, self.weight)


class DualAdam(torch.optim.Adam):
    """Adam ascent followed by projection onto the non-negative orthant (model/dual.py:27-45).  The trainers step
    the multipliers with the fused HIP kernel (rpo_adam_step, maximize + clamp); this class is the API-compatible
    torch form."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = super().step(closure=closure)
        for group in self.param_groups:
            for p in group["params"]:
                p.clamp_(0)
        return loss
