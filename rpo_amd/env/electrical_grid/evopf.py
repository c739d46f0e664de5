"""EVOPF-v0: AC optimal power flow on the IEEE 14-bus network with a battery at every generator bus (reference:
rpo/env/electrical_grid/evopf.py).  One episode = 24 hourly steps of a random load / price day.

Observation [57] = per-unit active / reactive demand of the 14 buses, state of charge of the 5 batteries, 24-hour price
look-ahead.  Action [43] = (pg[5], qg[5], vm[14], va[14], pe[5]).  The actor sets the 14 basic actions (pg at the 4 PV
generators, vm at the 5 generator buses, pe); ``complete_partial`` solves the 28 power-balance equations for the rest
by Newton's method, ``ineq_partial_grad`` is the reduced gradient used by the GRG projection.  All of it runs in the
HIP kernels of rpo_amd/csrc/evopf.hip (one wavefront per env lane); this class holds the tables and the Python surface.

Differences to the reference, all stated in DESIGN.md: the loaders' random day (np.random.dirichlet / rand / randn,
data/demand.py:53-62, data/price.py:41-43) comes from the build's Philox streams keyed by (seed, lane, episode);
``render`` (igraph) and ``opt_solve`` (pypower's interior-point OPF) are not part of the hot path.
"""
import numpy as np
import torch

from ... import ops as hip_ops
from ..base import HardConstraintEnv, gym
from . import case14

spaces = gym.spaces


class EVOPFEnv(HardConstraintEnv):
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 50}
    volatile = True                                                      # evopf.py:346

    def __init__(self, backend=None, device=None):
        super().__init__(backend, device)
        bus, gen = case14.BUS, case14.GEN
        self.nbus, self.ng = bus.shape[0], gen.shape[0]
        self.ne = self.ng
        self.nahead = case14.NAHEAD
        self.baseMVA = case14.BASE_MVA
        self.genbase = gen[:, case14.GEN_MBASE]
        # bus classes and index sets, evopf.py:225-239,278-311
        self.slack = np.where(bus[:, case14.BUS_TYPE] == 3)[0]
        self.pv = np.where(bus[:, case14.BUS_TYPE] == 2)[0]
        self.spv = np.sort(np.concatenate([self.slack, self.pv]))
        self.pq = np.setdiff1d(np.arange(self.nbus), self.spv)
        self.slack_ = np.array([np.where(x == self.spv)[0][0] for x in self.slack])
        self.pv_ = np.array([np.where(x == self.spv)[0][0] for x in self.pv])
        self.spv_ = np.arange(self.ng)
        self.nslack, self.npv = len(self.slack), len(self.pv)
        if list(self.spv) != [0, 1, 2, 5, 7]:
            raise ValueError("the EVOPF kernels are compiled for the IEEE-14 bus classification")
        self.pg_start_yidx, self.qg_start_yidx = 0, self.ng
        self.vm_start_yidx, self.va_start_yidx = 2 * self.ng, 2 * self.ng + self.nbus
        self.pe_start_yidx = 2 * self.ng + 2 * self.nbus
        self._xdim, self._ydim = 2 * self.nbus, 2 * self.ng + 2 * self.nbus + self.ne
        self._partial_vars = np.concatenate([self.pg_start_yidx + self.pv_, self.vm_start_yidx + self.spv,
                                             self.va_start_yidx + self.slack, self.pe_start_yidx + self.spv_])
        self._other_vars = np.setdiff1d(np.arange(self._ydim), self._partial_vars)
        self._partial_actions = np.concatenate([self.pg_start_yidx + self.pv_, self.vm_start_yidx + self.spv,
                                                self.pe_start_yidx + self.spv_])
        self._other_actions = self._other_vars.copy()
        self.we, self.wg = 5.0, 1.0                                                       # evopf.py:244-245
        self.pmax, self.pmin = gen[:, case14.GEN_PMAX] / self.genbase, gen[:, case14.GEN_PMIN] / self.genbase
        self.qmax, self.qmin = gen[:, case14.GEN_QMAX] / self.genbase, gen[:, case14.GEN_QMIN] / self.genbase
        self.vmax, self.vmin = bus[:, case14.VMAX].copy(), bus[:, case14.VMIN].copy()
        # Battery(p_data, num=ne, genbase=baseMVA, init_strategy="empty"), evopf.py:243 with the defaults of :25-26
        self.evs_low, self.evs_high, self.evs_p_min, self.evs_p_max = 0.1, 0.8, -0.2, 0.2
        self.eta_in = self.eta_out = 0.9
        pd_max = qd_max = 10.0                                                            # evopf.py:319-331
        high = np.array([pd_max] * self.nbus + [qd_max] * self.nbus + [self.evs_high] * self.ne + [240.0] * self.nahead)
        low = np.array([-pd_max] * self.nbus + [-qd_max] * self.nbus + [self.evs_low] * self.ne + [0.0] * self.nahead)
        a_high = np.concatenate([self.pmax, self.qmax, self.vmax, [np.pi] * self.nbus, [self.evs_p_max] * self.ne])
        a_low = np.concatenate([self.pmin, self.qmin, self.vmin, [-np.pi] * self.nbus, [self.evs_p_min] * self.ne])
        self.action_space = spaces.Box(low=a_low.astype(np.float32), high=a_high.astype(np.float32), dtype=np.float32)
        self.observation_space = spaces.Box(low=low.astype(np.float32), high=high.astype(np.float32), dtype=np.float32)
        self.state_dim, self.action_dim = self.observation_space.shape[0], self.action_space.shape[0]
        self.eq_num, self.ineq_num = 28, 58                                               # evopf.py:336-337
        self._table = case14.kernel_constants(hip_ops.CONST, hip_ops.CONST["RPO_EVOPF_CONSTS_LEN"])
        self._box_cache = {}
        self.seed()
        self.state = None
        self._episodes = 0

    # ------------------------------------------------------------------------------------------ properties
    partial_actions = property(lambda self: self._partial_actions)
    other_actions = property(lambda self: self._other_actions)
    partial_vars = property(lambda self: self._partial_vars)
    other_vars = property(lambda self: self._other_vars)
    xdim = property(lambda self: self._xdim)
    ydim = property(lambda self: self._ydim)
    neq = property(lambda self: 2 * self.nbus)
    nineq = property(lambda self: 4 * self.ng + 2 * self.nbus)

    @property
    def box_constraint(self):
        return self.action_space.low, self.action_space.high

    @property
    def box_constraint_partial(self):
        return self.action_space.low[self.partial_actions], self.action_space.high[self.partial_actions]

    def _make_kernels(self):
        return self._backend.EvopfKernels(self._table)

    # ------------------------------------------------------------------------------------------ state-dependent box
    def battery_bounds(self, soc):
        """Battery.update_bound (evopf.py:127-143): charge-rate limits shrunk by the state of charge; torch or numpy."""
        if isinstance(soc, torch.Tensor):
            p_max = torch.clamp(self.evs_high - soc, max=self.evs_p_max) / self.eta_in
            p_min = self.eta_out * torch.clamp(self.evs_low - soc, min=self.evs_p_min)
        else:
            p_max = np.minimum(self.evs_p_max, self.evs_high - soc) / self.eta_in
            p_min = self.eta_out * np.maximum(self.evs_p_min, self.evs_low - soc)
        return p_max, p_min

    def update(self, state, full=False):
        """(low, high) of the action box for every row of ``state`` (evopf.py:769-783).  Tensors in, tensors out (on
        the state's device, no host round trip -- the reference goes through numpy); arrays in, arrays out."""
        low, high = self.box_constraint if full else self.box_constraint_partial
        if isinstance(state, torch.Tensor):
            if state.dim() == 1:
                state = state.view(1, -1)
            key = (state.device, bool(full))
            if key not in self._box_cache:
                self._box_cache[key] = (torch.as_tensor(low, device=state.device), torch.as_tensor(high, device=state.device))
            lo, hi = self._box_cache[key]
            n = state.shape[0]
            soc = state[:, -self.ne - self.nahead:-self.nahead]
            p_max, p_min = self.battery_bounds(soc)
            return (torch.cat([lo[:-self.ne].expand(n, -1), p_min], dim=1),
                    torch.cat([hi[:-self.ne].expand(n, -1), p_max], dim=1))
        state = np.asarray(state)
        if state.ndim == 1:
            state = state[None]
        n = state.shape[0]
        p_max, p_min = self.battery_bounds(state[:, -self.ne - self.nahead:-self.nahead])
        low, high = low[None].repeat(n, axis=0), high[None].repeat(n, axis=0)
        low[:, -self.ne:], high[:, -self.ne:] = p_min, p_max
        return low, high

    # ------------------------------------------------------------------------------------------ gym API (1 env)
    def reset(self):
        """evopf.py:369-380: hour 0 of a fresh random day (episode counter of this env instance)."""
        vec = self._single()
        vec.ep_count.fill_(self._episodes)
        self._episodes += 1
        vec.reset()
        self.state = vec.obs[0].cpu().numpy().astype(np.float64)
        return self.state.copy()

    def step(self, action):
        """evopf.py:348-366 through the vectorised kernel (n = 1, no auto-reset): (obs, reward, done, info)."""
        vec = self._single()
        a = self._t(np.asarray(action, dtype=np.float32).reshape(1, -1))
        vec.ctrl.zero_()
        self.kernels.step(vec.internal, vec.obs, a, vec.ep_len, vec.ep_ret, vec.ep_count, self._row, 1, None, vec.ctrl,
                          2 ** 31 - 1, False, vec.viol_thresh, vec.seed, 0)
        row = self._row[0].cpu().numpy()
        c = self.kernels.cols
        self.state = vec.obs[0].cpu().numpy().astype(np.float64)
        info = {"ineq_viol": row[c["ineq_viol"][0]:c["ineq_viol"][1]][None].copy(),
                "eq_viol": row[c["eq_viol"][0]:c["eq_viol"][1]][None].copy()}
        return self.state.copy(), float(row[c["reward"][0]]), bool(row[c["done"][0]] > 0.5), info

    def decompose(self, state):
        return state[:2 * self.nbus], state[2 * self.nbus:]

    def get_action_vars(self, action):
        ng, nb = self.ng, self.nbus
        return (action[:, :ng], action[:, ng:2 * ng], action[:, 2 * ng:2 * ng + nb], action[:, -self.ne - nb:-self.ne],
                action[:, -self.ne:])

    def obj_fn(self, action):
        """Generation cost / mean(genbase)^2 (evopf.py:509-518)."""
        action = self._t(action)
        if action.dim() == 1:
            action = action.view(1, -1)
        pg_mw = action[:, :self.ng] * self._t(self.genbase)
        quad, lin = self._t(case14.GENCOST[:, 4]), self._t(case14.GENCOST[:, 5])
        cost = (quad * pg_mw ** 2).sum(dim=1) + (lin * pg_mw).sum(dim=1) + float(case14.GENCOST[:, 6].sum())
        return cost / float(self.genbase.mean() ** 2)

    # ------------------------------------------------------------------------------------------ constraint API
    def complete_partial(self, state, action_partial):
        state, ap = self._t(state), self._t(action_partial)
        if ap.dim() == 1:
            ap = ap.view(1, -1)
        return super().complete_partial(state if state.dim() == 2 else state.view(1, -1), ap)

    def _resid_backward(self, obs, action, grad_eq, grad_ineq):
        """d/d action of the residuals: inequality rows are constant +-identity blocks (evopf.py:663-707); the equality
        rows go through rpo_evopf_eq_vjp (torch autograd through eq_resid, used by the Lagrangian baselines)."""
        ng, nb = self.ng, self.nbus
        g = torch.zeros_like(action)
        if grad_eq is not None:                                 # J_eq^T grad_eq, what autograd derives from eq_resid
            self.kernels.eq_vjp(action.contiguous(), grad_eq.contiguous(), g, autograd_sign=True)
        if grad_ineq is not None:
            g[:, :ng] += grad_ineq[:, :ng] - grad_ineq[:, ng:2 * ng]
            g[:, ng:2 * ng] += grad_ineq[:, 2 * ng:3 * ng] - grad_ineq[:, 3 * ng:4 * ng]
            g[:, 2 * ng:2 * ng + nb] += grad_ineq[:, 4 * ng:4 * ng + nb] - grad_ineq[:, 4 * ng + nb:4 * ng + 2 * nb]
            g[:, -self.ne:] += grad_ineq[:, -2 * self.ne:-self.ne] - grad_ineq[:, -self.ne:]
        return g

    # -- the reference's remaining constraint functions (evopf.py:564-594,614-707), thin host methods over the kernels:
    # Jacobian entries come from rpo_evopf_eq_vjp with unit cotangents (autograd_sign = 0: eq_jac's own battery sign, DESIGN E1)
    def eq_jac(self, action):
        """d eq_resid / d action as eq_jac builds it (evopf.py:614-661) -> [n, 28, 43]."""
        a = self._t(action)
        a = (a if a.dim() == 2 else a.view(1, -1)).contiguous()
        n, m, y = a.shape[0], self.eq_num, self._ydim
        rows = a.repeat_interleave(m, dim=0).contiguous()                    # row (b, r) = action b with cotangent e_r
        cot = torch.eye(m, device=a.device, dtype=a.dtype).repeat(n, 1).contiguous()
        out = torch.empty(n * m, y, device=a.device, dtype=a.dtype)
        self.kernels.eq_vjp(rows, cot, out, autograd_sign=False)
        return out.view(n, m, y)

    def ineq_jac(self, state, action):
        """Constant +-identity blocks (evopf.py:663-707), expanded over the batch -> [n, 58, 43]."""
        n = self._t(action).reshape(-1, self._ydim).shape[0]
        key = ("ineq_jac", self.device)
        if key not in self._box_cache:
            ng, nb, ne, y = self.ng, self.nbus, self.ne, self._ydim
            j = torch.zeros(self.ineq_num, y)
            r = 0
            for start, size in ((self.pg_start_yidx, ng), (self.qg_start_yidx, ng), (self.vm_start_yidx, nb),
                                (self.pe_start_yidx, ne)):
                j[r:r + size, start:start + size] = torch.eye(size)
                j[r + size:r + 2 * size, start:start + size] = -torch.eye(size)
                r += 2 * size
            self._box_cache[key] = j.to(self.device)
        return self._box_cache[key].unsqueeze(0).expand(n, -1, -1)

    def eq_grad(self, state, action):
        """2 J_eq^T eq_resid (evopf.py:574-577) -> [n, 43]."""
        state, a = self._t(state), self._t(action)
        a = (a if a.dim() == 2 else a.view(1, -1)).contiguous()
        resid = self.eq_resid(state if state.dim() == 2 else state.view(1, -1), a).detach().contiguous()
        out = torch.empty_like(a)
        self.kernels.eq_vjp(a, resid, out, autograd_sign=False)
        return 2 * out

    def _ineq_jac_t(self, weights):
        """ineq_jac^T weights for [n, 58] weights: every inequality bounds one action component from above or below."""
        ng, nb, ne = self.ng, self.nbus, self.ne
        g = torch.zeros(weights.shape[0], self._ydim, device=weights.device, dtype=weights.dtype)
        g[:, :ng] = weights[:, :ng] - weights[:, ng:2 * ng]
        g[:, ng:2 * ng] = weights[:, 2 * ng:3 * ng] - weights[:, 3 * ng:4 * ng]
        g[:, 2 * ng:2 * ng + nb] = weights[:, 4 * ng:4 * ng + nb] - weights[:, 4 * ng + nb:4 * ng + 2 * nb]
        g[:, -ne:] = weights[:, -2 * ne:-ne] - weights[:, -ne:]
        return g

    def _dist2(self, state, action):
        state, a = self._t(state), self._t(action)
        return self.ineq_dist(state if state.dim() == 2 else state.view(1, -1), a if a.dim() == 2 else a.view(1, -1)).detach()

    def ineq_grad(self, state, action, eps=0.0):
        """2 J_ineq^T (dist + eps 1[dist > 0]) (evopf.py:579-583)."""
        d = self._dist2(state, action)
        return 2 * self._ineq_jac_t(d + (d > 0) * eps)

    def ineq_grad_new(self, state, action, eps=0.0):
        """J_ineq^T 1[dist > 0]: +-1 per violated bound (evopf.py:585-589); `eps` is unused there too."""
        return self._ineq_jac_t((self._dist2(state, action) > 0).to(torch.float32))

    def ineq_dist_np(self, state, action):
        """evopf.py:564-567: one (state, action) pair in, [1, 58] array out."""
        a = torch.as_tensor(np.asarray(action), dtype=torch.float32, device=self.device).view(1, -1)
        s = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.device).view(1, -1)
        return self.ineq_dist(s, a).detach().cpu().numpy()

    def eq_resid_np(self, state, action):
        """evopf.py:569-572."""
        a = torch.as_tensor(np.asarray(action), dtype=torch.float32, device=self.device).view(1, -1)
        s = torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.device).view(1, -1)
        return self.eq_resid(s, a).detach().cpu().numpy()

    def opt_solve(self, *a, **k):
        raise NotImplementedError("pypower's interior-point OPF baseline (evopf.py:729-759) is outside the RPO hot path")
