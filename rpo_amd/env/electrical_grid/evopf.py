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

    def opt_solve(self, *a, **k):
        raise NotImplementedError("pypower's interior-point OPF baseline (evopf.py:729-759) is outside the RPO hot path")
