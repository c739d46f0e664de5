"""Vectorised RPO trainer core shared by ``RPODDPG`` and ``RPOSAC`` (reference: rpo/algo/rpo_ddpg.py, rpo_sac.py).

One loop iteration = one *vector* step: N env instances act, are projected onto their constraints, stepped and
scattered into the replay ring by three launches (actor MLP -> ``*_act_project`` -> ``*_step``), followed -- at the
reference's cadence -- by one constrained policy update on a batch of ``batch_size`` sampled transitions
(rpo_ddpg.py:91-161).  Nothing in an iteration synchronises with the host: the step counter, ring position, RNG
counters and Adam steps live in device memory, so a whole iteration is captured once into a hipGraph and replayed.
With ``num_envs == 1`` the loop is the reference's algorithm step for step.

Data parallelism (``torch.distributed`` initialised, backend nccl == RCCL): rank r owns env ids
``[r * n_local, (r + 1) * n_local)`` and its own replay shard; the flat gradient slice of each optimiser is
all-reduced (mean) once per update, so replicas stay bit-identical without parameter broadcasts.
"""
import copy
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import ops as hip_ops
from .agent.fused import FusedNets
from .model import BoxConstraint



class NonFiniteError(FloatingPointError):
    """An env lane was stepped with a NaN action or reached a non-finite state (ctrl[RPO_CTRL_NONFINITE] of the step
    kernels; the reference's `assert self.action_space.contains(action_fixed)`, cartpole.py:170-174, pendulum.py:85-89)."""


_SALT_CRITIC = 1 << 20      # Philox index offsets that separate the update step's draws from the rollout's
_SALT_ACTOR = 2 << 20


# ------------------------------------------------------------------------------------------------ autograd bridges
class _TDHuberFn(torch.autograd.Function):
    """TD target + Huber loss, forward and backward in one HIP launch (rpo_td_huber)."""

    @staticmethod
    def forward(ctx, backend, gamma, alpha, reward, done, qn1, qn2, logp, q1, q2):
        n = q1.shape[0]
        loss = torch.zeros(1, device=q1.device)
        g1 = torch.empty(n, device=q1.device)
        g2 = torch.empty(n, device=q1.device) if q2 is not None else None
        flat = lambda x: None if x is None else x.reshape(-1).contiguous()   # noqa: E731
        backend.td_huber(flat(q1), flat(q2), flat(qn1), flat(qn2), flat(logp), alpha, reward, done, gamma, loss, g1, g2)
        ctx.save_for_backward(g1, g2)
        ctx.shape = q1.shape
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        g1, g2 = ctx.saved_tensors
        return (None,) * 8 + (grad_out * g1.view(ctx.shape), None if g2 is None else grad_out * g2.view(ctx.shape))


class _LagrangianFn(torch.autograd.Function):
    """mean_b nu . relu(g(a_b)): loss, d/da and d/dnu from one HIP launch (rpo_*_lagrangian)."""

    @staticmethod
    def forward(ctx, kernels, action, nu, state=None):
        n = action.shape[0]
        loss = torch.zeros(1, device=action.device)
        g_a = torch.empty_like(action)
        g_nu = torch.zeros(nu.numel(), device=action.device)
        kernels.lagrangian(action.contiguous(), nu.reshape(-1).contiguous(), 1.0 / n, loss, g_a, g_nu, obs=state)
        ctx.save_for_backward(g_a, g_nu)
        ctx.nu_shape = nu.shape
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        g_a, g_nu = ctx.saved_tensors
        return None, grad_out * g_a, grad_out * g_nu.view(ctx.nu_shape), None


# ------------------------------------------------------------------------------------------------ helpers
class _Dist(object):
    def __init__(self, force=False):
        init = dist.is_available() and dist.is_initialized()
        # force (trainer kwarg `force_dist`; RPO_SCHEDULE=force_dist=1): take the data-parallel code path with a single rank
        # too (tests, bench.py --force-dist: the RCCL collective inside the iteration's hipGraph on a one-GPU box)
        self.on = init and (dist.get_world_size() > 1 or bool(force))
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        # RCCL collectives are stream-ordered device work: they are captured INSIDE the iteration's hipGraph (and inside
        # the multi-iteration windows) like any kernel.  Host-driven backends (gloo: CPU tests, two ranks sharing one
        # GPU) cannot be captured: there the iteration is cut into graph segments with eager collectives in between.
        self.in_graph = self.on and dist.get_backend() == "nccl"

    def mean_(self, tensors):
        """In-place all-reduce(mean) of flat gradient buckets: ONE collective per bucket, one bucket per update (critic
        slice; on policy steps also actor slice + multipliers)."""
        if not self.on:
            return
        avg = dist.get_backend() == "nccl"          # RCCL averages in the collective; gloo needs the explicit scale
        for t in tensors:
            if avg:
                dist.all_reduce(t, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t.mul_(1.0 / self.world)

    def sync_initial_state(self, seed, agent, device):
        """Broadcast rank 0's parameters (flat buffer incl. the multipliers), target networks and Philox seed."""
        bufs = [agent.flat.data, agent.critic_target_flat]
        if agent.actor_target_flat is not None:
            bufs.append(agent.actor_target_flat)
        for b in bufs:
            dist.broadcast(b, src=0)
        s = torch.tensor([int(seed)], dtype=torch.int64, device=device)
        dist.broadcast(s, src=0)
        return int(s.item())

    def sum_(self, t):
        if self.on:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_ok(self, ok, device):
        """True iff `ok` holds on EVERY rank (one MIN all-reduce of a flag; host-synchronising -- for rare decisions only)."""
        if not self.on:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def replaying_everywhere(self, graphs, device):
        """For the bench line: did every rank replay captured graphs (with the collectives inside them)?"""
        mine = bool(graphs.enabled and not graphs.capture_failed and any(e["graph"] is not None for e in graphs.entries.values()))
        return self.all_ok(mine, device)


_NODE_KINDS = ("kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record", "sem_signal", "sem_wait",
               "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "batch_mem_op")


def _graph_node_kinds(g):
    """{kind: count} of the nodes of a captured ``torch.cuda.CUDAGraph(keep_graph=True)`` (hipGraphGetNodes /
    hipGraphNodeGetType of the HIP runtime the process is bound to).  The windows are meant to hold KERNEL nodes only (plus the
    collective backend's own in data-parallel runs): a hipMemsetAsync captured into them was replayed with a stale fill
    pattern, so anything that is not a kernel is worth a look -- ``RPO_GRAPH_AUDIT=1``."""
    import ctypes
    from .. import _lib
    _lib._bind_to_torch_hip_runtime()
    bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    hip = ctypes.CDLL(bundled if os.path.exists(bundled) else "libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
    graph = ctypes.c_void_p(int(g.raw_cuda_graph()))
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(1, n.value))()
    if n.value and hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    kinds = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        if hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        name = _NODE_KINDS[t.value] if 0 <= t.value < len(_NODE_KINDS) else "type_%d" % t.value
        kinds[name] = kinds.get(name, 0) + 1
    return kinds


class _GraphCache(object):
    """Eager for the first ``warm`` calls of a key (on a side stream, as hipGraph capture of autograd wants), then
    captured once and replayed.  Every call performs the work exactly once."""

    def __init__(self, enabled, warm=3, owner=None):
        self.enabled, self.warm = enabled, warm
        self.capture_failed = False     # a capture was attempted and dropped (on this rank or, data-parallel, on any rank)
        self.entries = {}
        self.side = None
        self.audit = os.environ.get("RPO_GRAPH_AUDIT", "0") == "1"   # keep the captured hipGraph_t and count its node kinds
        # host state `fn` changes while it is being captured (hand-over flags, launch arguments): an aborted capture has
        # run the Python but no launch -- the eager re-run must start from the state the capture started from
        # (owner._host_state / _set_host_state).  A WEAK reference: a trainer <-> cache cycle would leave the trainer's
        # hipGraphs to the cyclic collector, which may run -- and destroy them -- in the middle of a later capture.
        import weakref
        self._owner = weakref.ref(owner) if owner is not None else None

    def run(self, key, fn):
        if not self.enabled:
            return fn()
        e = self.entries.setdefault(key, {"count": 0, "graph": None})
        if e["graph"] is not None:
            e["graph"].replay()
            return
        if e["count"] < self.warm:
            e["count"] += 1
            if self.side is None:
                self.side = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                fn()
            cur.wait_stream(self.side)
            return
        g = torch.cuda.CUDAGraph(keep_graph=True) if self.audit else torch.cuda.CUDAGraph()
        owner = self._owner() if self._owner is not None else None
        state = owner._host_state() if owner is not None else None
        import gc
        gc_was_on = gc.isenabled()
        # A collection inside the capture could destroy an older trainer's hipGraph ("operation not permitted when stream is
        # capturing", fatal) -- and one right BEHIND it, between the instantiation and the first launch of the new graph, was a
        # segmentation fault inside hipGraphLaunch in the full test session (round 5: trainers of earlier tests were still
        # waiting for the cyclic collector).  So: collect what is collectable NOW, outside the capture, and keep the collector
        # off until the new graph has been launched once.
        gc.collect()
        gc.disable()
        failure = None
        try:
            try:
                # thread-local capture mode: the collective backend's watchdog thread may touch the HIP runtime meanwhile
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    fn()
            except Exception as exc:                            # noqa: BLE001  (capture is an optimisation, not a need)
                failure = exc
            # Data-parallel ranks decide TOGETHER: a rank that replays a window with captured collectives cannot pair with a
            # rank that issues them eagerly one iteration at a time in another order of host work -- if any rank's capture
            # failed, every rank drops its graph and continues eagerly (one flag all-reduce per capture, never per iteration).
            ok_everywhere = failure is None
            if owner is not None and owner.dist.on and owner.dist.world > 1:
                ok_everywhere = owner.dist.all_ok(failure is None, owner.device)
            if not ok_everywhere:
                import warnings
                why = "%s: %s" % (type(failure).__name__, failure) if failure is not None else "another rank's capture failed"
                warnings.warn("hipGraph capture failed (%s); continuing with eager launches" % why)
                self.enabled = False
                self.capture_failed = True
                del g
                torch.cuda.synchronize()
                if owner is not None:
                    owner._set_host_state(state)
                fn()                                            # nothing ran during the aborted capture
                return
            e["graph"] = g
            if self.audit:
                e["node_kinds"] = _graph_node_kinds(g)
            g.replay()
        finally:                                                # (KeyboardInterrupt / SystemExit inside fn() too: ADVICE r03)
            if gc_was_on:
                gc.enable()


def _env_int(name, default):
    v = os.environ.get(name)
    return default if v in (None, "") else int(v)


#: The launch schedule's optional parts, all ON by default.  Every part off gives the same results (bit for bit
#: unless a test states a tolerance): they exist for the A/B tests that pin the fused / riding launches against the plain
#: ones, and as the operational fallback the hand-over error messages name (`front=0`).  One variable sets them from outside
#: -- RPO_SCHEDULE="front=0,ride=0" -- or the trainers' `schedule=dict(front=0)` argument; read ONCE, at construction.
#:   fused_mlp      hand-written f32-MFMA MLP kernels (0: the torch modules + autograd)
#:   fused_rollout  actor -> head -> Complete -> GRG -> env step -> ring scatter as one launch (0: three launches)
#:   fused_critic   critic-update pipelines (0: generic MLP launches + rpo_*_act_project / rpo_td)
#:   fused_actor    policy-step pipelines (0: generic launches)
#:   split          column-split update stages rpo_split_* (0: the row-tile pipelines of fused.hip)
#:   ride           the next vector step rides on the critic update's launches where nothing is shared (0: serial windows)
#:   front          in-launch hand-overs: fused critic / policy fronts, the SpringPendulum front around the projection and the
#:                  projection on one workgroup per row tile (0: one launch per stage, one-workgroup projection)
#:   branch         second captured branch of the generic-launch windows (EVOPF-v0): rollout t+1 beside update t, and the
#:                  actor-only prefix of the policy step beside the critic update (0: serial windows)
#:   force_dist     (default 0) data-parallel code path over a one-rank process group
SCHEDULE_DEFAULTS = dict(fused_mlp=1, fused_rollout=1, fused_critic=1, fused_actor=1, split=1, ride=1, front=1, branch=1,
                         force_dist=0)


def parse_schedule(overrides=None):
    out = dict(SCHEDULE_DEFAULTS)
    text = os.environ.get("RPO_SCHEDULE", "")
    for item in filter(None, (x.strip() for x in text.split(","))):
        k, _, v = item.partition("=")
        if k.strip() not in out:
            raise ValueError("RPO_SCHEDULE: unknown part %r (have: %s)" % (k, ", ".join(sorted(out))))
        out[k.strip()] = int(v) if v.strip() else 1
    for k, v in (overrides or {}).items():
        if k not in out:
            raise ValueError("schedule: unknown part %r (have: %s)" % (k, ", ".join(sorted(out))))
        out[k] = int(v)
    return out


# ------------------------------------------------------------------------------------------------ trainer core
class RPOTrainerBase(object):
    """Everything except agent construction and the two losses."""

    sac = False

    def _setup(self, env, work_dir, name, logger, agent, hp, device, num_envs=None, seed=None, backend=None,
               use_graph=None, updates_per_step=None, fused=None, schedule=None):
        self.env, self.agent, self.device = env, agent, device
        self.schedule = parse_schedule(schedule)
        if fused is not None:
            self.schedule["fused_mlp"] = int(bool(fused))
        self.work_dir, self.name, self.logger = work_dir, name, logger
        for k, v in hp.items():
            setattr(self, k, v)
        if self.eval_steps is None:
            self.eval_steps = self.max_steps
        if self.corr_mode != 0:
            # rpo_ddpg.py:276-278 / rpo_sac.py:290-292: `nju(ineq_grad) + lamb(eq_grad)` applies Dual.forward (F.linear with a
            # [1, ineq_num] / [1, eq_num] weight, model/dual.py:63-65) to gradients of width action_dim: a shape error in the
            # reference for every env it ships (cart 2 vs 6 / 1, pendulum 2 vs 1 / 1, EVOPF 43 vs 58 / 28; verified on the
            # unmodified reference: "mat1 and mat2 shapes cannot be multiplied (1x2 and 6x1)").  There is no behaviour to
            # reproduce, so the argument is refused here instead of at the first projection.
            raise ValueError("corr_mode=%r: only corr_mode=0 (reduced-gradient projection) is defined; the reference's "
                             "corr_mode=1 branch (rpo_ddpg.py:276-278) raises a shape error for every env" % (self.corr_mode,))
        self.backend = backend if backend is not None else hip_ops
        self.base_env = getattr(env, "unwrapped", env)
        self.kernels = self.base_env.kernels
        self.env_eval = copy.deepcopy(self.env)                          # rpo_ddpg.py:61
        self.box_constraint = BoxConstraint(*self.base_env.box_constraint, device=device)
        self.decay_value = (hp["eps_start"] - hp["eps"]) / hp["eps_epoch"]   # rpo_ddpg.py:70
        self.dist = _Dist(force=self.schedule["force_dist"])
        n_total = int(num_envs) if num_envs is not None else _env_int("RPO_NUM_ENVS", 1)
        if n_total % self.dist.world:
            raise ValueError("num_envs (%d) must be divisible by the world size (%d)" % (n_total, self.dist.world))
        self.num_envs, self.n_local = n_total, n_total // self.dist.world
        if seed is None:
            seed = _env_int("RPO_SEED", None)
        if seed is None:
            # drawn from torch's global generator AFTER the networks were initialised, so that the initial weights
            # match the reference for the same torch.manual_seed and the Philox streams still depend on that seed
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        if self.dist.on:
            # data-parallel replicas must start identical and draw from ONE Philox seed; only gradients travel
            # afterwards.  Rank 0's initial state is made everybody's, once, here (a rank that skipped
            # torch.manual_seed, or built its networks in another order, would otherwise average the gradients of a
            # different model without any error).
            seed = self.dist.sync_initial_state(seed, agent, device)
        self.seed = int(seed)
        self.max_episode_steps = getattr(env, "_max_episode_steps", None)
        self.vec = self.base_env.make_vec(self.n_local, seed=self.seed, env_id_base=self.dist.rank * self.n_local,
                                          max_episode_steps=self.max_episode_steps, device=device)
        self.buffer = agent.attach_env(self.kernels, self.n_local, self.seed + 7919 * (self.dist.rank + 1), self.vec.ctrl)
        self._batch = torch.zeros(self.batch_size, self.kernels.row_floats, device=device)
        P = self.kernels.partial_dim                                    # basic actions per env (1, 1, 14)
        self._noise_b = torch.zeros(self.batch_size, P, device=device)
        self._noise_n = torch.zeros(self.n_local, P, device=device)
        self._box_lo, self._box_hi = self.base_env.partial_box
        # RPO_GRAPH_CYCLE: iterations per hipGraph window (default 16, rounded to a multiple of policy_fre); 1: one graph per
        # iteration; 0: no hipGraphs at all (eager launches)
        cycle = _env_int("RPO_GRAPH_CYCLE", 16)
        if use_graph is None:
            use_graph = cycle > 0 and device.type == "cuda"
        self._graphs = _GraphCache(use_graph, owner=self)
        self._tail = None           # data-parallel runs: deferred last segment of the previous iteration
        self._last_cols = self._last_actor_out = None
        self._cycle = max(cycle, 1) // max(1, self.policy_fre) * max(1, self.policy_fre)
        # projection of training batches: the reference's literal batched semantics (default) or row-wise
        # (RPO_ROWWISE_PROJECTION=1); rollouts are always per lane == the reference's B = 1 calls
        rowwise = _env_int("RPO_ROWWISE_PROJECTION", None)
        self.batch_reference = not bool(rowwise)
        if self.batch_size > 1024 and self.batch_reference and hasattr(self.kernels, "project_batchref"):
            # the reference's batched projection couples every sample of a SpringPendulum batch with every other (n^2 terms
            # per GRG iteration, pendulum.py:337-339) and stops on the batch maximum (rpo_ddpg.py:271-272): the
            # kernel is defined up to 1024 rows.  Larger batches are projected row by row -- OTHER semantics than the
            # reference's, so it has to be asked for (RPO_ROWWISE_PROJECTION=1), not fallen into.  Envs without
            # a batch-coupled kernel (CartSafe: both forms coincide, cartpole.py:403; EVOPF: hazard E2) are not affected.
            raise ValueError(
                "batch_size=%d > 1024 on %s: the reference's sample-coupled batched projection is defined up to 1024 rows "
                "here; set RPO_ROWWISE_PROJECTION=1 to project training batches row by row (per-sample stop test, "
                "no coupling between samples)" % (self.batch_size, self.kernels.name))
        #: recorded in checkpoints / bench lines: how training batches are projected
        self.projection_mode = ("batch-reference" if self.batch_reference and hasattr(self.kernels, "project_batchref")
                                else "row-wise")
        # hand-written f32-MFMA MLP kernels for actor / critics (schedule fused_mlp=0: the torch modules + autograd)
        self.fused = FusedNets.build(agent, self.backend, device) if self.schedule["fused_mlp"] else None
        if self.fused is not None and self.batch_size >= hip_ops.CONST.get("RPO_SPLITK_FROM", 1 << 30) and device.type == "cuda":
            # one LARGE batch per update (batch 256 * N): the parameter-gradient reductions of the backward
            # kernels split the batch over the chip (rpo_mlp_grad.splitk_scratch)
            self.fused.enable_splitk(self.batch_size)
        # updates per vector step: 1 = the reference's loop cadence (rpo_ddpg.py:160-161); num_envs = "UTD-matched":
        # as many batch-`batch_size` updates per env step as the reference performs (metric iii)
        self.updates_per_step = max(1, int(updates_per_step) if updates_per_step is not None
                                    else _env_int("RPO_UPDATES_PER_STEP", 1))
        self._updates = 0           # updates done so far (drives the policy_fre cadence of the extra updates)
        box = agent.actor.box_constraint
        self._box_affine = (float(np.asarray(box.scale).reshape(-1)[0]), float(np.asarray(box.base).reshape(-1)[0])) \
            if box is not None and not box.volatile else None
        # hand-written backward kernels accumulate into the flat gradient buffer; their optimiser steps then consume the
        # gradients (zero them), which removes the fill launches from the iteration (data-parallel: the slices are
        # averaged in place between backward and step, then consumed the same way).  The torch/autograd path keeps its
        # explicit zeroing.
        self._self_cleaning = self.fused is not None
        self._gradmax_ready = False
        for opt in (agent.critic_optim, agent.actor_optim, agent.nju_optim):
            opt.zero_grad_after = self._self_cleaning
        # fused multi-output actor: raw outputs, the env's projection kernel applies the state-dependent tanh box
        self._act_kw = dict(ap_is_raw=True) if (self.fused is not None and self._box_affine is None) else {}
        # Update clock: the update kernels of the column-split path read their step index from `_uctrl[0]` instead of
        # ctrl[RPO_CTRL_T], and the iteration's last optimiser launch advances it -- so the NEXT rollout (which advances
        # ctrl[RPO_CTRL_T]) may run on another stream of the same hipGraph while the update is still going.  Without the
        # split path it is an alias of the rollout's ctrl and nothing changes.
        self._uctrl, self._uclock_ok, self._ovl_stream = self.vec.ctrl, True, None
        self._ride, self._rider_cache, self._ride_cut = None, None, 0
        self._clock_pending, self._iter_actor_step, self._critic_prepared, self._gradmax_stale = False, None, False, False
        self._actor_prepared, self._actor_gradmax_stale = False, False
        self._bump_updates_now, self._updates_out, self._pol_a_done = False, None, False
        self._bump = self.updates_per_step == 1
        if self.fused is not None and device.type == "cuda" and getattr(self.backend, "ADAM_CLOCK", False):
            self._uctrl, self._uclock_ok = torch.zeros_like(self.vec.ctrl), False
        self.buffer.sample_ctrl = self._uctrl
        self._split_state()
        self._t = 0                 # loop iterations (== vector steps) done
        self._harvested = 0         # vector steps whose statistics were already pulled off the device
        self._pending = []          # per-step rows waiting for the return of the episodes they belong to
        self._vec_eval = None
        self.last_losses = {}
        self._idx_inject = None     # tests: callable returning the replay indices of the next sampled batch
        self._eval_init_inject = None   # tests: [10, internal_dim] initial states of the evaluation episodes
        self.viol_steps, self.env_steps, self.viol_rate, self.proj_iters_mean = 0.0, 0.0, 0.0, 0.0

    # ------------------------------------------------------------------------------------------ projection API
    def process_action(self, state, action_partial, train=True):
        """Equation solver + GRG projection (rpo_ddpg.py:72-77), one fused launch."""
        ref = self.batch_reference
        if train:
            return self.base_env.project(state, action_partial, self.max_steps, self.corr_lr, self.corr_eps,
                                         self.corr_momentum, batch_reference=ref)
        action, iters = self.base_env.project(state, action_partial, self.eval_steps, self.eval_lr, self.corr_eps,
                                              self.corr_momentum, return_iters=True, batch_reference=ref)
        return action, int(iters.max())

    def grad_steps(self, state, action, train=True):
        """GRG loop on an action that already satisfies the equalities (rpo_ddpg.py:266-305, corr_mode 0).  The fused
        kernel restarts from the action's basic components, which reproduces the reference whenever the input lies on
        the equality manifold -- the only way the reference ever calls it (rpo_ddpg.py:74-75)."""
        idx = torch.as_tensor(np.asarray(self.base_env.partial_actions), device=self.device)
        return self.process_action(state, torch.as_tensor(action, device=self.device)[:, idx], train=train)

    # ------------------------------------------------------------------------------------------ rollout
    def _policy_partial(self, obs, warm):
        """Basic action proposed for every lane + the noise mode the projection kernel must apply."""
        raise NotImplementedError

    #: RPOSAC samples the squashed Gaussian inside the rollout pipeline; RPODDPG adds exploration noise to the actor
    _gauss_policy = False

    @property
    def _large_batch(self):
        """One LARGE batch per update (batch_size >= RPO_SPLITK_FROM, e.g. 256 * num_envs: as many sampled transitions per
        env step as the reference consumes): the update runs through the generic MLP kernels -- 64-row tiles
        in the forward (0.48 of the f32 MFMA peak at 2^20 rows), split-K weights pass in the backward -- instead of the
        batch-256 pipelines, whose 16-row tiles each stream the whole weight matrix."""
        return self.batch_size >= hip_ops.CONST.get("RPO_SPLITK_FROM", 1 << 30)

    @property
    def _rollout_pipeline(self):
        return (self.fused is not None and hasattr(self.kernels, "rollout") and "actor" in self.fused.descs
                and self.schedule["fused_rollout"])

    @property
    def _defer_ok(self):
        """A rollout that is followed by the column-split critic update may leave its step counter to that update's first
        launch (rpo_*_rollout(defer_clock=1), rpo_split_update.rollout_ctrl): saves the arrival counting behind the
        rollout's last workgroup."""
        return bool(getattr(self, "_pipelines", False) and self._split_state() is not None)

    def _rollout(self, warm, defer_clock=False):
        """``defer_clock``: the caller runs the column-split critic update next, whose first launch advances the step
        counter for this one (`_critic_update_split`)."""
        v = self.vec
        defer_clock = bool(defer_clock and not warm and self._rollout_pipeline and self._defer_ok)
        if not warm and self._rollout_pipeline:
            # actor -> head -> equation solver -> projection -> env step -> replay scatter in ONE launch (fused.hip)
            scale, base = self._box_affine
            self.kernels.rollout(self.fused.descs["actor"], self._gauss_policy, scale, base, v.internal,
                                 None if v.obs is v.internal else v.obs, v.action, v.ep_len, v.ep_ret, v.ep_count,
                                 self.buffer.rows, self.buffer.capacity, v.stats, v.ctrl,
                                 hip_ops.NOISE_NONE if self._gauss_policy else hip_ops.NOISE_PHILOX, self.eps_start,
                                 self.eps, self.decay_value, self._box_lo, self._box_hi, self.max_steps, self.corr_lr,
                                 self.corr_eps, self.corr_momentum, v.max_episode_steps, True, v.viol_thresh, self.seed,
                                 v.env_id_base, **(dict(defer_clock=True) if defer_clock else {}))
            self._clock_pending = defer_clock
            v.steps_host += 1
            return
        with torch.no_grad():
            ap, mode = self._policy_partial(v.obs, warm)
            self.kernels.act_project(v.obs, ap, None, v.action, None, mode, self.eps_start, self.eps, self.decay_value,
                                     self._box_lo, self._box_hi, self.max_steps, self.corr_lr, self.corr_eps,
                                     self.corr_momentum, self.seed, v.env_id_base, v.ctrl, v.stats,
                                     **({} if warm else self._act_kw))
            v.step(v.action, rows=self.buffer.rows, cap_steps=self.buffer.capacity, auto_reset=True)

    # ------------------------------------------------------------------------------------------ update
    def _sample(self):
        self.buffer.sample_rows(self.batch_size, out=self._batch)
        c = self.buffer.split(self._batch)
        return c["state"], c["action"], c["next_state"], c["reward"], c["done"], c["ineq_viol"], c["eq_viol"]

    def _eps_now(self):
        """Exploration scale inside the update (take_action in actor_loss, rpo_ddpg.py:309).  Constant in every script
        (eps == eps_start); computed from the device step counter when it decays so that graph replays stay exact."""
        if self.decay_value == 0:
            return self.eps_start
        t = self._uctrl[0].to(torch.float32)
        return torch.clamp(self.eps_start - self.decay_value * t, min=self.eps)

    def _critic_gradmax(self):
        """Where the critic's backward kernels may leave the inf-norm of its gradient slice (clip_grad_norm_,
        rpo_ddpg.py:180): only when that slice was zero before (self-cleaning steps) and is not averaged over ranks
        afterwards.  `_critic_step` then skips the rpo_absmax launch."""
        opt = self.agent.critic_optim
        ok = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf")
        self._gradmax_ready = False
        return opt.gradmax if ok else None

    def _zero_grads(self):
        """Before a backward of the fused path: nothing to do when the optimiser steps leave zeroed slices behind
        (only the multipliers of a `fixed` run are never stepped)."""
        if not self._self_cleaning:
            self.agent.flat.grad.zero_()
        elif self.fixed:
            self.agent.nju.weight.grad.zero_()

    def _critic_update(self, cols):
        state, action, next_state, reward, done, ineq_viol, eq_viol = cols
        loss = self.critic_loss(state, action, next_state, done, reward, ineq_viol, eq_viol)
        self.agent.flat.grad.zero_()
        loss.backward()
        self.last_losses["critic"] = loss.detach()

    def _actor_update(self, cols):
        out = self.actor_loss(cols[0])
        loss = out[0] if isinstance(out, tuple) else out
        self.agent.flat.grad.zero_()       # parameters and multipliers share the flat gradient buffer
        loss.backward()
        self.last_losses["actor"] = loss.detach()
        # detached: the result outlives the iteration on the trainer (segments / graph replays), and must not keep the
        # autograd graph alive across iterations
        return tuple(x.detach() for x in out) if isinstance(out, tuple) else loss.detach()

    def _segments(self, warm, do_train, actor_step, rollout=True):
        """The iteration as a list of (device work, gradient buffers to all-reduce afterwards).  ``rollout=False``: the
        update alone (the iteration of an evaluation is split: rollout | eval() | update, rpo_ddpg.py:140-161)."""
        ag, fl = self.agent, self.agent.flat
        segs = []
        if not do_train:
            return [(lambda: self._rollout(warm), [])]
        # the sampled columns / the actor's outputs are views of static buffers: they are kept on the trainer (not in a
        # per-iteration closure) because a segment may be replayed from its graph while a later one still runs eagerly

        def s1():
            if rollout:
                self._rollout(warm, defer_clock=True)
            self._last_cols = self._sample()
            self._iter_actor_step = actor_step
            self._bump_updates_now = self._updates_inkernel      # (updates_per_step > 1: for the first extra update)
            self._critic_update(self._last_cols)
        segs.append((s1, [fl.gradient(fl.critic_range)]))
        if actor_step:
            def s2():
                self._critic_step(actor_step)
                self._last_actor_out = self._actor_update(self._last_cols)
            segs.append((s2, [fl.gradient(fl.policy_bucket)]))

            def s3():
                self._actor_step(self._last_actor_out)
            segs.append((s3, []))
        else:
            segs.append((lambda: self._critic_step(actor_step), []))
        return segs

    def _critic_step(self, actor_step):
        raise NotImplementedError

    def _clock(self, last):
        """`clock=` argument of the optimiser launch that ends the iteration's update (``last``), see `_uctrl`."""
        return self._uctrl if (last and self._bump and self._uctrl is not self.vec.ctrl) else None

    def _sync_uclock(self, rollout_pending):
        """Set the update clock to the number of vector steps the next update must see (eagerly, outside any capture):
        needed whenever the previous iteration did not end with a clock-advancing optimiser launch."""
        if self._uctrl is self.vec.ctrl or self._uclock_ok:
            return
        if rollout_pending:
            torch.add(self.vec.ctrl[0:1], 1, out=self._uctrl[0:1])
        else:
            self._uctrl[0:1].copy_(self.vec.ctrl[0:1])

    def _actor_step(self, actor_out):
        raise NotImplementedError

    def _policy_optims(self):
        """The optimisers `_actor_step` steps, in the order of its slices (the prepared launch's bookkeeping list)."""
        ag = self.agent
        out = [ag.actor_optim]
        if not self.fixed:
            out.append(ag.nju_optim)
        if getattr(self, "automatic_entropy_tuning", False):
            out.append(ag.alpha_optim)
        return out

    def _iteration(self, warm, do_train, actor_step, rollout=True):
        segs = self._segments(warm, do_train, actor_step, rollout)
        if do_train:
            self._sync_uclock(rollout_pending=rollout)
        self._uclock_ok = do_train and self._bump
        if not self.dist.on or self.dist.in_graph:
            self._graphs.run((warm, do_train, actor_step, rollout), lambda: self._run_segments(segs))
            return
        # Collectives stay eager between captured segments.  The last segment of an iteration (the optimiser step that
        # follows the last all-reduce) has no collective behind it: it is deferred and captured together with the first
        # segment of the NEXT iteration, so that an iteration costs as many graph launches as it has all-reduces.
        tail, self._tail = self._tail, None
        for i, (fn, reduce_after) in enumerate(segs):
            key = (warm, do_train, actor_step, rollout, i)
            if i == 0 and tail is not None:
                prev_fn, first = tail[1], fn
                key, fn = ("after",) + tail[0] + key, (lambda: (prev_fn(), first()))
            if i == len(segs) - 1 and not reduce_after and i > 0:
                self._tail = (key, fn)                             # flushed by the next iteration or _flush_tail()
                return
            self._graphs.run(key, fn)
            self.dist.mean_(reduce_after)

    def _run_segments(self, segs):
        """Segments back to back on the current stream, each followed by the all-reduce of its gradient bucket
        (data-parallel runs over RCCL: the collective is captured with the kernels around it)."""
        for fn, reduce_after in segs:
            fn()
            self.dist.mean_(reduce_after)

    def _flush_tail(self):
        """Run the deferred last segment of the previous iteration now (before anything reads the parameters)."""
        tail, self._tail = self._tail, None
        if tail is not None:
            self._graphs.run(tail[0], tail[1])

    @property
    def _updates_inkernel(self):
        """Several updates per vector step: ctrl[RPO_CTRL_UPDATES] is advanced by the update's own last stage (column-split
        critic AND policy stages) instead of a torch launch between two updates."""
        return bool(self.updates_per_step > 1 and getattr(self, "_pipelines", False) and getattr(self, "_actor_pipeline", False)
                    and self._split_state() is not None)

    def _extra_body(self, actor_step):
        if self._updates_inkernel:
            self._bump_updates_now = True                       # (the previous update advanced the counter for this one)
        else:
            self._uctrl[hip_ops.CONST["RPO_CTRL_UPDATES"]] += 1
        cols = self._sample()
        self._iter_actor_step = actor_step
        self._critic_update(cols)
        fl = self.agent.flat
        self.dist.mean_([fl.gradient(fl.critic_range)])
        self._critic_step(actor_step)
        if actor_step:
            out = self._actor_update(cols)
            self.dist.mean_([fl.gradient(fl.policy_bucket)])
            self._actor_step(out)

    def _extra_updates(self):
        """The 2nd .. updates_per_step-th update of the current vector step: same kernels as the first one, with
        ctrl[RPO_CTRL_UPDATES] = k separating their Philox draws; windows of RPO_GRAPH_CYCLE updates share a hipGraph."""
        F, k, U = self.policy_fre, 1, self.updates_per_step
        while k < U:
            L = self._cycle
            if L > 1 and U - k >= L and self._updates % F == 0 and self._graphs.enabled and \
                    (not self.dist.on or self.dist.in_graph):
                base = self._updates
                self._graphs.run(("extra", L), lambda: [self._extra_body((base + j + 1) % F == 0) for j in range(L)])
            else:
                L = 1
                actor_step = (self._updates + 1) % F == 0
                if self.dist.on and not self.dist.in_graph:
                    self._extra_body(actor_step)                # host-driven collectives inside: eager
                else:
                    self._graphs.run(("extra", actor_step), lambda: self._extra_body(actor_step))
            self._updates += L
            k += L

    def train(self, t):
        """One constrained policy update at loop index ``t`` (rpo_ddpg.py:163-205), eagerly, without a rollout."""
        self._uclock_ok = False
        self._sync_uclock(rollout_pending=False)                # the update sees the vector steps taken so far
        bump, self._bump = self._bump, False                    # ... and leaves the clock alone
        try:
            self._train_body(t)
        finally:
            self._bump = bump
            self._uclock_ok = False

    def _train_body(self, t):
        cols = self._sample()
        actor_step = self._iter_actor_step = t % self.policy_fre == 0
        self._critic_update(cols)
        fl = self.agent.flat
        self.dist.mean_([fl.gradient(fl.critic_range)])
        self._critic_step(actor_step)
        if actor_step:
            out = self._actor_update(cols)
            self.dist.mean_([fl.gradient(fl.policy_bucket)])
            self._actor_step(out)

    # ------------------------------------------------------------------------------------------ column-split update
    def _split_state(self):
        """The column-split update stages (rpo_split_*, rpo_amd/csrc/nsplit.hip) when this configuration supports them:
        CartSafe / SpringPendulum kernels, every network 128 -> 256 with scalar heads, the reference's batched projection
        semantics on SpringPendulum, batch <= 1024.  Schedule ``split=0`` keeps the row-tile pipelines."""
        if getattr(self, "_split_cache", False) is not False:
            return self._split_cache
        self._split_cache = None
        f, k, be = self.fused, self.kernels, self.backend
        if f is None or not hasattr(be, "SplitUpdate") or not self.schedule["split"] or self._box_affine is None \
                or self.device.type != "cuda":
            return None
        if not isinstance(k, (be.CartSafeKernels, be.PendulumKernels)) or self.batch_size > 1024:
            return None
        if isinstance(k, be.PendulumKernels) and not self.batch_reference:
            return None
        d = f.descs
        names = ("actor", "critic1", "critic2", "critic_target1", "critic_target2") if self.sac else \
            ("actor", "actor_target", "critic", "critic_target")
        if any(n not in d or not be.mlp_split_supported(d[n]) for n in names):
            return None
        B, ag, buf = self.batch_size, self.agent, self.buffer
        T = (B + 15) // 16
        b = f.buf
        descs = dict(actor=d["actor"])
        if self.sac:
            descs.update(critic1=d["critic1"], critic2=d["critic2"], critic_target1=d["critic_target1"],
                         critic_target2=d["critic_target2"])
        else:
            descs.update(actor_target=d["actor_target"], critic1=d["critic"], critic_target1=d["critic_target"])
        scale, base = self._box_affine
        c1 = "critic1" if self.sac else "critic"
        fields = dict(
            rows=buf.rows, cap_steps=buf.capacity, n_envs=buf.n_envs, batch_out=self._batch, sample_seed=buf.seed,
            sample_salt=0, noise_seed=self.seed, noise_id_base=self.dist.rank * B, noise_salt=_SALT_CRITIC, ctrl=self._uctrl,
            scale=scale, base=base, box_lo=self._box_lo, box_hi=self._box_hi, max_steps=self.max_steps,
            corr_lr=self.corr_lr, corr_eps=self.corr_eps, corr_momentum=self.corr_momentum,
            alpha=float(getattr(ag, "alpha", 0.0)), gamma=ag.gamma, eps_start=self.eps_start, eps_end=self.eps,
            eps_decay=self.decay_value,
            part_pi=b("split.part_pi", 8, B, 2), part_pol=b("split.part_pol", 8, B, 2),
            part_q1=b("split.part_q1", 8, B, 2), part_qn1=b("split.part_qn1", 8, B, 2),
            x0_1=b(c1 + ".x0", B, d[c1].ein), h1_1=b(c1 + ".h1", B, d[c1].H), dq1=b("dq1" if self.sac else "dq", B, 1),
            dx0_1=b(c1 + ".dx0", B, d[c1].ein), loss_partial=b("split.loss_parts", 2, T),
            next_actions=b("split.next_actions", B, 2), logp=b("crit.logp", B),
            tile_sync=torch.zeros((3 * T + 1) * 32, dtype=torch.int32, device=self.device))
        if self.sac:
            fields.update(part_q2=b("split.part_q2", 8, B, 2), part_qn2=b("split.part_qn2", 8, B, 2),
                          x0_2=b("critic2.x0", B, d["critic2"].ein), h1_2=b("critic2.h1", B, d["critic2"].H),
                          dq2=b("dq2", B, 1), dx0_2=b("critic2.dx0", B, d["critic2"].ein))
        # policy step
        da_ = d["actor"]
        fields.update(nu=ag.nju.weight.view(-1), nu_grad=ag.nju.weight.grad.view(-1), noise_out=b("act.noise", B),
                      actions=b("act_pi", B, 2), g_act=b("g_act", B, 2), lag_partial=b("actor.parts", T, 8),
                      lag_out=b("actor.lag", 2), da_part=b("split.da_part", 2, 8, B, 2), dout=b("split.dout", B, 2),
                      x0_a=b("actor.x0", B, da_.ein), h1_a=b("actor.h1", B, da_.H), dx0_a=b("actor.dx0", B, da_.ein),
                      shared_embedding=int(ag.flat.sizes[1] > 0))
        if self.sac:
            fields.update(raw=b("pi.raw", B, 2))
        else:
            fields.update(ap_det=b("act.ap_det", B))
        if isinstance(k, be.PendulumKernels) and self.schedule["front"] and B <= 256 and self.max_steps <= 30:
            # the batch-coupled projection on one workgroup per row tile (rpo_split_pend_head_project); workspace of
            # RPO_PROJ_WS_WORDS 64-bit words; store mode 1: plain stores when the workgroups share an XCD (checked inside every
            # launch), agent-scope granule stores otherwise
            fields.update(proj_ws=torch.zeros(hip_ops.PROJ_WS_WORDS, dtype=torch.int64, device=self.device), proj_store_mode=1)
        self._split_cache = be.SplitUpdate(k, descs, self.sac, B, fields)
        self._front_cache = bool(self.schedule["front"]) and hasattr(be, "front_launch_ok") and be.front_launch_ok(B, self.sac)
        # SpringPendulum: fwd_a + projection + fwd_b + bwd_a as one launch (rpo_split_critic_pfront)
        self._pfront = bool(self._front_cache and "proj_ws" in fields and be.front_launch_ok(B, self.sac, 1))
        self._split_loss = fields["loss_partial"]
        self._split_logp = (fields["logp"], b("pi.logp", B))          # log pi(a'|s') of the critic update | log pi(a|s)
        return self._split_cache

    def _actor_update_split(self, su):
        """Policy step through the column-split stages: pol_a .. pol_e (rpo_amd/csrc/nsplit.hip)."""
        B, ag, k = self.batch_size, self.agent, self.kernels
        noise_in = None
        if self._idx_inject is not None:                          # tests replay the reference's draw
            self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * B * k.partial_dim, _SALT_ACTOR,
                                       hip_ops.STREAM_POLICY, self._uctrl)
            noise_in = self._noise_b.view(-1)
        crit_logp, pi_logp = self._split_logp
        su.set(noise_salt=_SALT_ACTOR, eps_in=noise_in, logp=pi_logp)
        early, self._pol_a_done = self._pol_a_done, False
        front = self._front_ok()
        if front:                                                 # pol_a (unless done early), pol_b, pol_c and pol_d as one launch
            su.run("policy_front_bc" if early else "policy_front")
        else:
            if not early:                                         # (else: done inside fwd_b's launch of this iteration)
                su.run("policy_a")
            su.run("policy_b")
            su.run("policy_c")
        self._zero_grads()
        opt = ag.actor_optim
        fuse_max = self._self_cleaning and not self.dist.on and opt.clip_thres and opt.clip_thres != float("inf")
        su.set(gradmax=opt.gradmax if fuse_max else None)
        if not front:
            su.run("policy_d")
        # "prepared" optimiser launch behind the policy step (as for the critic's, `_critic_update_split`): pol_e advances the
        # step counters of the slices `_actor_step` will step and the update clock; the next fwd_a zeroes the actor's gradmax
        # (only when the critic update runs through the split stages too: its fwd_a is what zeroes the gradmax afterwards)
        prep = bool(fuse_max) and bool(getattr(self, "_pipelines", False))
        optims = [o for o in self._policy_optims() if o is not None] if prep else []
        su.set_prep2([(o.step_dev, o.betas[0], o.betas[1]) for o in optims])
        su.set(clock_out=self._clock(True) if prep else None, updates_out=self._updates_out)
        su.run("policy_e")
        su.set(clock_out=None, updates_out=None)
        self._updates_out = None
        self._actor_prepared = prep
        self._actor_gradmax_stale = prep
        su.set(noise_salt=_SALT_CRITIC, eps_in=None, logp=crit_logp)
        self._actor_gradmax_ready = bool(fuse_max)
        f = self.fused
        return f.buf("actor.lag", 2), f.buf("actor.parts", (B + 15) // 16, 8), pi_logp

    def _set_hand_overs(self, su, actor_step):
        """What this critic update does on behalf of other launches, and what it leaves to later ones (every
        hand-over relies on stream order only -- a later launch starts after every workgroup of an earlier one has finished):

          fwd_a   advances the step counter of the rollout launched right before (`_clock_pending`: defer_clock) and zeroes
                  the gradmax buffers the previous update's prepared Adam launches consumed (`_gradmax_stale`)
          bwd_b   prepares the critic's Adam launch (step counter, bias corrections), advances the update clock when no
                  policy step follows, and the update sub-index when several updates share a vector step

        ``actor_step``: whether a policy step follows this critic update (None: the caller did not say -> no hand-overs,
        the optimiser launches keep their own bookkeeping)."""
        opt = self.agent.critic_optim
        # (data-parallel runs too: the inf-norm then comes from rpo_absmax_slots behind the all-reduce, into the same slots)
        prep = actor_step is not None
        self._critic_prepared = bool(prep)
        bump_updates, self._bump_updates_now = self._bump_updates_now, False
        self._updates_out = self._uctrl if bump_updates else None          # (pol_e does it behind a policy step)
        pending, self._clock_pending = self._clock_pending, False
        su.set(prep_step=opt.step_dev if prep else None, prep_beta1=opt.betas[0], prep_beta2=opt.betas[1],
               clock_out=self._clock(not actor_step) if prep else None,
               updates_out=self._updates_out if not actor_step else None,
               # Prepared optimiser launches no longer zero their gradmax slots: fwd_a does, for BOTH optimisers and on
               # every prepared update -- not only when a host flag says a prepared launch ran before.  The pointers are
               # frozen into the hipGraph at capture, and a replay does not re-run this Python: with the flags, a graph
               # captured at a position of the policy_fre period that follows a critic-only iteration (policy_fre 3 or 5)
               # never cleared the actor's slots, and clip_grad_norm_ saw a running maximum.  Both buffers are filled only
               # after fwd_a of the same update, so the extra zeroing is harmless.
               gradmax_reset=opt.gradmax if (prep or self._gradmax_stale) else None,
               gradmax_reset2=self.agent.actor_optim.gradmax if (prep or self._actor_gradmax_stale) else None,
               rollout_ctrl=self.vec.ctrl if pending else None, rollout_stats=self.vec.stats if pending else None,
               rollout_stats_cap=self.vec.stats.shape[0] if pending else 0)
        self._gradmax_stale, self._actor_gradmax_stale = bool(prep), False

    def _critic_update_split(self, su):
        """Critic update through the column-split stages: fwd_a | (pend: head + batch projection) | fwd_b | bwd_a | bwd_b."""
        inject = self._idx_inject is not None                   # tests replay the reference's draws
        idx_in = self._idx_inject() if inject else None
        eps_in = None
        if inject and self.sac:
            B = self.batch_size
            self.backend.philox_normal(self._noise_b, self.seed, self.dist.rank * B * self.kernels.partial_dim, _SALT_CRITIC,
                                       hip_ops.STREAM_POLICY, self._uctrl)
            eps_in = self._noise_b.view(-1)
        buf = self.buffer
        su.set(idx_in=idx_in, eps_in=eps_in, rows=buf.rows, cap_steps=buf.capacity, n_envs=buf.n_envs)   # (tests swap the ring)
        actor_step, self._iter_actor_step = self._iter_actor_step, None
        self._set_hand_overs(su, actor_step)
        ride = self._ride                                       # ridden windows: the next vector step rides along
        if ride is not None:
            n, cut = self.vec.internal.shape[0], self._ride_cut
            ride.set(lane_begin=0, lane_end=cut)
        # CartSafe: fwd_a, fwd_b and bwd_a are one launch (the later stages' workgroups wait inside it for the
        # workgroups of their own row tile, rpo_split_critic_front) -- same values, two launch boundaries less
        early = bool(actor_step) and ride is None and self.agent.flat.sizes[1] == 0 and getattr(self, "_actor_pipeline", False)
        if su.st.env == 0 and self._front_ok():
            self._pol_a_done = early
            if ride is not None:
                ride.set(lane_begin=0, lane_end=n)              # the whole actor forward of the next step rides along
            su.run("critic_front_pol" if early else "critic_front", rider=ride)   # (pol_a as one more plane, see below)
            self._critic_update_split_back(su, ride, bwd_a=False)
            return
        if su.st.env == 1 and self._pfront:
            self._pol_a_done = early                             # SpringPendulum: the same, around the projection's workgroups
            if ride is not None:
                ride.set(lane_begin=0, lane_end=n)
            su.run("critic_pfront_pol" if early else "critic_pfront", rider=ride)
            self._critic_update_split_back(su, ride, bwd_a=False)
            return
        su.run("critic_fwd_a", rider=ride)                      # + actor forward of lanes [0, cut)
        if su.st.env == 1:
            su.run("pend_head_project")
        if ride is not None:
            ride.set(lane_begin=cut, lane_end=n)
        # policy iteration without a shared embedding: the policy slabs on the batch states (pol_a) need nothing the critic
        # update produces -- they are an extra plane of fwd_b's launch instead of a launch behind the critic step
        self._pol_a_done = early
        if su.st.env == 1 and self._front_ok():                 # SpringPendulum: fwd_b and bwd_a are one launch
            su.run("critic_mid_pol" if early else "critic_mid", rider=ride)
            self._critic_update_split_back(su, ride, bwd_a=False)
            return
        su.run("critic_fwd_b_pol" if early else "critic_fwd_b", rider=ride)   # (+ actor forward of lanes [cut, n))
        self._critic_update_split_back(su, ride)

    _HOST_STATE = ("_actor_pre", "_clock_pending", "_iter_actor_step", "_critic_prepared", "_gradmax_stale", "_actor_prepared",
                   "_actor_gradmax_stale", "_pol_a_done", "_gradmax_ready", "_actor_gradmax_ready", "_bump_updates_now",
                   "_updates_out", "_last_cols", "_last_actor_out")

    def _host_state(self):
        """What an iteration's Python changes on the HOST besides launching (hand-over flags, the argument struct of the
        column-split stages, the host mirror of the step counter): `_GraphCache` restores it when a capture is aborted."""
        import ctypes
        st = {k: getattr(self, k) for k in self._HOST_STATE if hasattr(self, k)}
        st["steps_host"] = self.vec.steps_host
        su = getattr(self, "_split_cache", None)
        if su:
            st["su"] = (ctypes.string_at(ctypes.addressof(su.st), ctypes.sizeof(su.st)), dict(su._held))
        rc = getattr(self, "_rider_cache", None)
        if rc is not None and hasattr(rc, "st"):
            st["rider"] = ctypes.string_at(ctypes.addressof(rc.st), ctypes.sizeof(rc.st))
        return st

    def _set_host_state(self, st):
        import ctypes
        st = dict(st)
        self.vec.steps_host = st.pop("steps_host")
        su_state, rider = st.pop("su", None), st.pop("rider", None)
        if su_state is not None:
            ctypes.memmove(ctypes.addressof(self._split_cache.st), su_state[0], len(su_state[0]))
            self._split_cache._held = dict(su_state[1])
        if rider is not None:
            ctypes.memmove(ctypes.addressof(self._rider_cache.st), rider, len(rider))
        for k, v in st.items():
            setattr(self, k, v)

    def _handover_flags(self):
        """Device words the in-launch hand-overs raise when a wait was given up: (tile_sync's last word, proj_ws's)."""
        su = getattr(self, "_split_cache", None)
        sync = su._held.get("tile_sync") if su else None
        ws = su._held.get("proj_ws") if su else None
        return (sync[-32:-31] if sync is not None and getattr(self, "_front_cache", False) else None,
                ws[hip_ops.PROJ_WS_GAVE_UP:hip_ops.PROJ_WS_GAVE_UP + 1] if ws is not None else None)

    def _device_flags(self):
        """[(kind, one-word device tensor)]: the hand-over words above and the step kernels' sticky failure word
        ctrl[RPO_CTRL_NONFINITE] (include/rpo_hip.h; SURVEY 5 "failure detection": the reference's NaN -> assert,
        cartpole.py:170-174 / pendulum.py:85-89, kept as a device-side flag)."""
        f, g = self._handover_flags()
        nf = hip_ops.CONST["RPO_CTRL_NONFINITE"]
        return [(k, x) for k, x in (("front", f), ("proj", g), ("nonfinite", self.vec.ctrl[nf:nf + 1])) if x is not None]

    def _raise_flags(self, vals):
        if vals.get("front"):
            raise RuntimeError("rpo_split_*_front: a workgroup gave up waiting for its row tile (tile_sync flag set); the "
                               "values of that launch are undefined -- rerun with RPO_SCHEDULE=front=0")
        if vals.get("proj"):
            raise RuntimeError("rpo_split_pend_head_project / rpo_split_critic_pfront: a workgroup gave up waiting for another "
                               "one's granules (workspace flag set); the values of that launch are undefined -- rerun with "
                               "RPO_SCHEDULE=front=0")
        if vals.get("nonfinite"):
            raise NonFiniteError("vector step %d (0-based): an env lane was stepped with a NaN action or reached a non-finite "
                                 "next state / reward (ctrl[RPO_CTRL_NONFINITE]).  The reference stops at this point too "
                                 "(`assert self.action_space.contains(action_fixed)`, cartpole.py:170-174); the transitions "
                                 "of that step are in the replay ring, so this trainer must not be trained on or "
                                 "checkpointed -- restart from the last checkpoint" % (int(vals["nonfinite"]) - 1))

    def _check_flags(self):
        """The fused front launches raise a flag word when a workgroup gave up waiting for its producers (nsplit.hip,
        kNsSpinMax / kPmSpinMax): the values of that launch are then undefined; the step kernels raise
        ctrl[RPO_CTRL_NONFINITE] when a lane goes non-finite -- fail loudly instead of training on.
        (Synchronising read: harvest, save(), the end of run().)"""
        vals = {k: int(x[0]) for k, x in self._device_flags()}
        if self.dist.on:                                          # every rank raises together (harvest / save are collective
            # points of the loop: no rank is left waiting in the next all-reduce).  A FIXED-size word: which hand-over
            # workspaces exist is a per-rank fact (the placement probe of 4.3 may answer differently on ranks that share a GPU)
            kinds = ("front", "proj", "nonfinite")
            word = torch.tensor([vals.get(k, 0) for k in kinds], dtype=torch.int64, device=self.device)
            dist.all_reduce(word, op=dist.ReduceOp.MAX)
            vals = dict(zip(kinds, word.tolist()))
        self._raise_flags(vals)

    def _poll_flags(self, every=4):
        """The same without waiting for the device: after every `every`-th call (graph windows: every FOURTH; eager
        iterations: every 16th) the flag words are copied to pinned host memory asynchronously (plain device-to-host
        copies, no kernel; every window cost 1.4 us per iteration in the kernel trace); a copy that has landed is
        inspected before a later window is launched, so a lost producer or a NaN actor stops the run within a few windows
        instead of at the next statistics harvest."""
        if self.device.type != "cuda":
            return self._raise_flags({k: int(x[0]) for k, x in self._device_flags()})
        st = getattr(self, "_flag_poll", None)
        if st is None:
            st = self._flag_poll = dict(event=None, calls=0, host={}, pending=[])
        if st["event"] is not None:
            if not st["event"].query():
                return                                            # the previous copy is still in flight: look again later
            st["event"] = None
            self._raise_flags({k: int(st["host"][k][0]) for k in st["pending"]})
        st["calls"] += 1
        if every > 1 and st["calls"] % every != 1:
            return
        flags = self._device_flags()                              # (the hand-over workspaces appear with the first update)
        st["pending"] = [k for k, _ in flags]
        for k, x in flags:
            h = st["host"].get(k)
            if h is None or h.dtype != x.dtype:
                h = st["host"][k] = torch.zeros(1, dtype=x.dtype).pin_memory()
            h.copy_(x, non_blocking=True)
        st["event"] = torch.cuda.Event()
        st["event"].record()

    def _front_ok(self):
        """rpo_split_critic_front usable here (schedule ``front=0``: never): see ops.front_launch_ok."""
        return self._front_cache                                 # (probed in _split_state, outside any graph capture)

    def _critic_update_split_back(self, su, ride, bwd_a=True):
        """bwd_a | bwd_b of the column-split critic update (``bwd_a=False``: it was part of the front launch)."""
        self._zero_grads()
        gm = self._critic_gradmax()
        su.set(gradmax=gm)
        if bwd_a:
            su.run("critic_bwd_a")
        if ride is not None:
            ride.set(defer_clock=int(self._defer_ok))           # ... whose step counter the next update's fwd_a advances
        su.run("critic_bwd_b", rider=ride)                      # + explore / project / step / scatter of every lane
        if ride is not None:
            self._clock_pending = bool(self._defer_ok)
            self.vec.steps_host += 1
        self._gradmax_ready = gm is not None
        from .rpo_ddpg import _LazySum
        self.last_losses["critic"] = _LazySum(self._split_loss if self.sac else self._split_loss[0])

    # ------------------------------------------------------------------------------------------ fused-MLP helpers
    def _project_batch(self, state, ap_flat):
        """Training-batch projection of `ap_flat` [B] -> actions [B, A] (same semantics as process_action)."""
        return self.base_env.project(state, ap_flat, self.max_steps, self.corr_lr, self.corr_eps, self.corr_momentum,
                                     batch_reference=self.batch_reference, **self._act_kw)

    def _complete_only(self, state, ap_flat, noise=None):
        """clip(ap + eps_t * noise) -> equation solver, no GRG steps (actor loss, rpo_ddpg.py:309-310)."""
        f = self.fused
        act = f.buf("act_pi", state.shape[0], self.kernels.action_dim)
        mode = hip_ops.NOISE_NONE if noise is None else hip_ops.NOISE_EXPLICIT
        self.kernels.act_project(state, ap_flat, noise, act, None, mode, self.eps_start, self.eps, self.decay_value,
                                 self._box_lo, self._box_hi, 0, 0.0, self.corr_eps, 0.0, self.seed, 0, self._uctrl, None,
                                 **self._act_kw)
        return act

    # ------------------------------------------------------------------------------------------ main loop
    def run(self, logger=None, eval=True):
        """rpo_ddpg.py:79-161 over vector steps.  ``max_epochs`` counts loop iterations exactly like the reference;
        each one advances all ``num_envs`` lanes."""
        if logger is not None:
            self.logger = logger
        if self._t == 0:
            self.vec.reset()
        self.run_steps(self.max_epochs - self._t, eval=eval)
        self._harvest(final=True)

    def run_steps(self, n, eval=False, train=True):
        """``n`` loop iterations; ``train=False`` collects rollouts only (no sampling, no update)."""
        left = int(n)
        while left > 0:
            t = self._t
            warm = t < self.warmup
            do_train = train and (t + 1) >= self.warmup
            actor_step = do_train and (t + 1) % self.policy_fre == 0
            L = self._cycle_len(t, left, warm, do_train, eval)
            eval_now = eval and (t + 1) % self.eval_fre == 0 and (t + 1) > self.warmup
            if eval_now and do_train:
                # the reference evaluates between the env step and train(t) of this iteration (rpo_ddpg.py:140-161):
                # rollout | eval() | update, so that eval() and the printed multipliers see the same parameters
                self._iteration(warm, False, False)
                self._advance_host(t + 1)
                self._flush_tail()
                self._harvest()
                self._print_eval(t + 1, self.eval())
                self._iteration(warm, True, actor_step, rollout=False)
                self._updates += 1
                if self.updates_per_step > 1:
                    self._flush_tail()
                    self._extra_updates()
                    self._uctrl[hip_ops.CONST["RPO_CTRL_UPDATES"]] = 0
                left -= 1
                continue
            if L > 1:
                # one hipGraph for L consecutive iterations (policy_fre-periodic launch pattern): the same launches
                # in the same order as L single-iteration replays, minus L - 1 graph-to-graph gaps (8.5 us each)
                if do_train:
                    self._sync_uclock(rollout_pending=True)
                self._uclock_ok = do_train and self._bump
                if self._ride_ok(do_train):
                    self._graphs.run(("cycle", L, True, "ride"), lambda: self._ridden_window(t, L))
                elif self._overlap_ok(do_train):
                    self._graphs.run(("cycle", L, True, "overlap"), lambda: self._overlapped_window(t, L))
                else:
                    self._graphs.run(("cycle", L, do_train), lambda: [self._run_segments(self._segments(
                        False, do_train, do_train and (t + i + 1) % self.policy_fre == 0)) for i in range(L)])
                self._updates += L if do_train else 0
                self._poll_flags()
            else:
                self._iteration(warm, do_train, actor_step)
                self._poll_flags(every=16)
                if do_train:
                    self._updates += 1
                    if self.updates_per_step > 1:
                        self._flush_tail()
                        self._extra_updates()
                        self._uctrl[hip_ops.CONST["RPO_CTRL_UPDATES"]] = 0
            for _ in range(L):
                t = t + 1
                self._advance_host(t)
            left -= L
            if t - self._harvested >= self.vec.stats.shape[0] // 2:
                self._harvest()
            if eval_now:                                                   # (rollout-only runs: nothing to order against)
                self._flush_tail()
                self._harvest()
                self._print_eval(t, self.eval())
        self._flush_tail()

    def _advance_host(self, t):
        """Host mirrors of the device-side counters after vector step ``t``."""
        self._t = t
        self.buffer.note_step()
        self.agent.eps_decay(self.decay_value, self.eps)
        self.vec.steps_host = t

    def _ride_ok(self, do_train):
        """The next vector step may ride on the critic update's launches (rpo_split_critic_fwd_a_ride / _fwd_b_ride /
        _bwd_b_ride, rpo_amd/csrc/nsplit.hip): column-split update, one-launch rollout available, no shared state embedding, the update
        on its own clock.  Results are identical to the serial order.  Schedule ``ride=0`` keeps the serial windows."""
        if not (do_train and self._uctrl is not self.vec.ctrl and self.agent.flat.sizes[1] == 0 and self._bump):
            return False
        if not (getattr(self, "_pipelines", False) and self._split_state() is not None and self._rollout_pipeline):
            return False
        return bool(self.schedule["ride"])

    def _rider(self):
        """Arguments of the riding rollout halves: what `_rollout` hands to the one-launch rollout."""
        v, buf = self.vec, self.buffer
        if self._rider_cache is None:
            scale, base = self._box_affine
            self._rider_cache = self.backend.RolloutRider(
                ring_floats=self.kernels.ring_floats, n_envs=v.internal.shape[0], gauss=int(self._gauss_policy), scale=scale, base=base, state=v.internal,
                obs=None if v.obs is v.internal else v.obs, action=v.action, ep_len=v.ep_len, ep_ret=v.ep_ret,
                ep_count=v.ep_count, stats=v.stats, stats_cap=v.stats.shape[0], ctrl=v.ctrl,
                noise_mode=hip_ops.NOISE_NONE if self._gauss_policy else hip_ops.NOISE_PHILOX, eps_start=self.eps_start,
                eps_end=self.eps, eps_decay=self.decay_value, box_lo=self._box_lo, box_hi=self._box_hi,
                max_steps=self.max_steps, corr_lr=self.corr_lr, corr_eps=self.corr_eps, corr_momentum=self.corr_momentum,
                max_episode_steps=v.max_episode_steps, auto_reset=1, viol_thresh=v.viol_thresh, seed=self.seed,
                env_id_base=v.env_id_base, part=self.fused.buf("ride.part", 8, v.internal.shape[0], 2))
        self._rider_cache.set(rows=buf.rows, cap_steps=buf.capacity)
        # separate launches (no fused front): half of the lanes' actor forward rides on fwd_a, the rest on fwd_b
        n = v.internal.shape[0]
        self._ride_cut = min(n, (n // 2 + 15) // 16 * 16)
        return self._rider_cache

    def _ridden_window(self, t, L):
        """L iterations (t is a policy_fre boundary): the rollout of iteration i+1 rides on the critic update of iteration
        i -- its actor forward in the launches of fwd_a and fwd_b (a lane range each), its explore / project / step / scatter
        in bwd_b's, behind fwd_a, which gathers the batch out of the ring first -- unless iteration i ends with a policy step,
        whose new actor the next rollout has to wait for."""
        F, fl = self.policy_fre, self.agent.flat
        self._rollout(False, defer_clock=True)
        for i in range(L):
            actor_step = (t + i + 1) % F == 0
            more = i + 1 < L
            ride = more and not actor_step
            self._ride = self._rider() if ride else None
            self._iter_actor_step = actor_step
            try:
                cols = self._last_cols = self._sample()
                self._critic_update(cols)
            finally:
                self._ride = None
            self.dist.mean_([fl.gradient(fl.critic_range)])
            self._critic_step(actor_step)
            if actor_step:
                self._last_actor_out = self._actor_update(cols)
                self.dist.mean_([fl.gradient(fl.policy_bucket)])
                self._actor_step(self._last_actor_out)
            if more and not ride:
                self._rollout(False, defer_clock=True)

    def _overlap_ok(self, do_train):
        """Rollout t+1 may run beside the update of t (on a second stream of the window's hipGraph) when the update does
        not touch what the rollout reads -- no shared state embedding, and not on policy steps -- and the update reads
        its own clock (`_uctrl`).  Results are identical either way.  OFF for the classic-control envs -- on one MI355X the
        fork / join of the second graph branch costs more than the 16 us rollout it hides (cart-SAC 75.7 vs 71.0 us per
        iteration, measured in round 2; the switch went with round 5) -- and ON for EVOPF-v0, whose rollout is a chain of
        latency-bound launches (one wavefront per lane) that runs well beside the update's.  (`_overlap_enabled = False`
        on a trainer keeps the serial order: the A/B test.)"""
        if not (do_train and self._uctrl is not self.vec.ctrl and self.agent.flat.sizes[1] == 0 and self._bump):
            return False
        if getattr(self, "_pipelines", False):                            # short launches: the branch costs more than it hides
            return False
        return bool(self.schedule["branch"] and getattr(self, "_overlap_enabled", True))

    def _policy_prefix_ok(self):
        """The actor-only prefix of the policy step may run beside the critic update: generic fused launches (not the policy
        pipelines, whose front is one launch), multipliers stepped (`fixed` runs would zero nu's gradient inside the prefix,
        concurrently with the critic update, for nothing: serial there).  `_policy_prefix_enabled = False` on a
        trainer keeps the serial order (A/B: tests/test_evopf_gpu.py)."""
        return bool(self.fused is not None and hasattr(self, "_actor_prefix") and not getattr(self, "_actor_pipeline", False)
                    and self.agent.flat.sizes[1] == 0 and not self.fixed and getattr(self, "_policy_prefix_enabled", True))

    def _overlapped_window(self, t, L):
        """L iterations (t is a policy_fre boundary) with rollout i+1 forked off right after the sampling launch of update i
        -- the gather must see the ring before the next rollout overwrites its oldest slot -- and joined before update
        i+1 samples.  After a policy step the next rollout waits for the new actor (serial).

        Capture order matters (round 5, tools/probe/branch_probe.py): of the two successors of the fork point, hipGraph keeps
        the one captured FIRST in the queue of the fork point (its next launch starts 2.8 us later) and hands the other to a
        second queue (9.5 us).  So the fork is an EVENT recorded behind the sampling launch, the update's launches -- the
        critical chain -- are captured first, and the second branch is captured behind them waiting for that event (it has the
        slack); the join then costs 5 us instead of 8.5.  Same dependencies, same bits."""
        F, fl = self.policy_fre, self.agent.flat
        main = torch.cuda.current_stream()
        if self._ovl_stream is None:
            self._ovl_stream = torch.cuda.Stream()
        side = self._ovl_stream
        self._rollout(False)
        for i in range(L):
            actor_step = (t + i + 1) % F == 0
            more = i + 1 < L
            overlap = more and not actor_step

            self._iter_actor_step = actor_step
            cols = self._last_cols = self._sample()
            # policy iteration: what the policy step computes from the actor alone (pi(s), noise, Complete, Lagrangian) runs on
            # the second branch beside the critic update (`_actor_prefix`; no shared embedding here: `_overlap_ok`)
            prefix = actor_step and self._policy_prefix_ok()
            if overlap or prefix:
                forked = torch.cuda.Event()
                forked.record(main)                                         # (_sample() launched the gather)
            self._critic_update(cols)
            self.dist.mean_([fl.gradient(fl.critic_range)])
            self._critic_step(actor_step)
            if overlap or prefix:                                           # the second branch, captured behind the critical chain
                side.wait_event(forked)
                with torch.cuda.stream(side):
                    if prefix:
                        self._actor_pre = self._actor_prefix(cols)
                    else:
                        self._rollout(False)
            if actor_step:
                if prefix:
                    main.wait_stream(side)
                self._last_actor_out = self._actor_update(cols)
                self.dist.mean_([fl.gradient(fl.policy_bucket)])
                self._actor_step(self._last_actor_out)
            if overlap:
                main.wait_stream(side)
            elif more:
                self._rollout(False)

    def _cycle_len(self, t, left, warm, do_train, eval):
        """Iterations the next launch may cover: RPO_GRAPH_CYCLE (default 16, rounded to a multiple of policy_fre) in the
        steady state of a graph run (single rank, or data-parallel over RCCL whose collectives are captured too; training:
        the window starts on a policy_fre boundary; or rollouts only), when neither an evaluation nor the statistics harvest falls inside it; otherwise 1."""
        L = self._cycle
        if L <= 1 or warm or (self.dist.on and not self.dist.in_graph) or not self._graphs.enabled:
            return 1
        if left < L:
            # the tail of a run: one shorter window of whole policy_fre periods (a hipGraph of its own per length once a caller
            # has repeated the same short run three times -- bench.py --steps 20 --, eager launches on the side stream before)
            L = left // self.policy_fre * self.policy_fre if do_train else left
            if L < 2:
                return 1
        if do_train and (t < self.warmup or t % self.policy_fre or self.updates_per_step > 1):
            return 1
        if eval and (t // self.eval_fre + 1) * self.eval_fre <= t + L:      # the evaluation splits its own iteration
            return 1
        if t + L - self._harvested >= self.vec.stats.shape[0]:
            return 1
        return L

    # ------------------------------------------------------------------------------------------ statistics
    def _harvest(self, final=False):
        """Pull the per-vector-step statistics rows written by the step kernel and turn them into Logger rows:
        ``epoch`` = vector step, ``max_ineq`` / ``max_eq`` = mean over lanes of the per-lane maxima
        (rpo_ddpg.py:120-123), ``reward`` = mean return of the episodes the step belongs to, back-filled when they
        finish (rpo_ddpg.py:135-137; identical to the reference for one lane)."""
        lo, hi = self._harvested, self._t
        if hi <= lo:
            return
        cap = self.vec.stats.shape[0]
        idx = torch.arange(lo, hi, device=self.device) % cap
        rows = hip_ops.reduce_stats(self.vec.stats[idx])
        if self.dist.on:                                                # sums add up over ranks, maxima take the max
            mx = rows.clone()
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            for k in ("max_ineq_max", "max_eq_max"):
                rows[:, hip_ops.STAT[k]] = mx[:, hip_ops.STAT[k]]
        rows = rows.cpu().numpy().astype(np.float64)
        self._check_flags()                                       # (the copy above already waited for the device)
        S = hip_ops.STAT
        n = float(self.num_envs)
        # whole-array bookkeeping (a Python loop over the rows kept the GPU idle for ~12 us per iteration at 4096 lanes)
        cur = np.stack([np.arange(lo, hi, dtype=np.float64), rows[:, S["max_ineq_sum"]] / n, rows[:, S["max_eq_sum"]] / n], 1)
        ends = np.nonzero(rows[:, S["episodes"]] > 0)[0]
        if len(ends) == 0:
            self._pending.extend(cur.tolist())
        else:
            P = len(self._pending)
            allr = np.concatenate([np.asarray(self._pending, dtype=np.float64).reshape(P, 3), cur], 0)
            last = int(ends[-1]) + P
            ret = rows[ends, S["return_sum"]] / rows[ends, S["episodes"]]
            which = np.searchsorted(ends + P, np.arange(last + 1), side="left")       # the episode end a step waits for
            if self.logger is not None:
                k = min(last + 1, max(0, self.logger.capacity - self.logger.pointer))  # like add() until StopIteration
                if k > 0:
                    self.logger.add_rows(epoch=allr[:k, 0], reward=ret[which[:k]], max_ineq=allr[:k, 1], max_eq=allr[:k, 2])
            if self.num_envs == 1 and self.dist.rank == 0 and _env_int("RPO_VERBOSE", 1):
                start = 0
                for j, e in enumerate(ends):
                    seg = allr[start:int(e) + P + 1]                                    # rpo_ddpg.py:134
                    length = rows[e, S["length_sum"]] / rows[e, S["episodes"]]
                    print("episode %d ends. reward: %s, step: %d, ineq_viol: %s, eq_viol: %s"
                          % (lo + int(e) + 1, ret[j], int(length), seg[:, 1].max(), seg[:, 2].max()))
                    start = int(e) + P + 1
            self._pending = allr[last + 1:].tolist()
        # constraint-violation rate: fraction of env steps with max(max_ineq, max_eq) > 1e-3
        self.viol_steps += rows[:, S["viol_count"]].sum()
        self.env_steps += n * (hi - lo)
        self.viol_rate = self.viol_steps / self.env_steps
        self.proj_iters_mean = rows[:, S["proj_iters"]].sum() / (n * (hi - lo))
        self._harvested = hi

    # ------------------------------------------------------------------------------------------ evaluation
    def _eval_partial(self, obs):
        raise NotImplementedError

    def _eval_action(self, v):
        """Deterministic policy + eval_steps projection iterations into v.action (rpo_ddpg.py:224-226)."""
        ap = self._eval_partial(v.obs)
        self.kernels.act_project(v.obs, ap, None, v.action, None, hip_ops.NOISE_NONE, 0.0, 0.0, 0.0, self._box_lo,
                                 self._box_hi, self.eval_steps, self.eval_lr, self.corr_eps, self.corr_momentum,
                                 **self._act_kw)

    def eval(self, rendering=False):
        """10 evaluation episodes of at most 500 steps with the deterministic policy and ``eval_steps`` projection
        iterations (rpo_ddpg.py:207-264), run as 10 parallel lanes.  Returns the reference's 10-tuple."""
        lanes, horizon = 10, 500
        if self._vec_eval is None:
            self._vec_eval = self.base_env.make_vec(lanes, seed=self.seed ^ 0x5EED5EED, env_id_base=0,
                                                    max_episode_steps=self.max_episode_steps, device=self.device,
                                                    stats_cap=2)
            self._eval_rows = torch.zeros(lanes, self.kernels.ring_floats, device=self.device)
        v, c = self._vec_eval, self.kernels.cols
        v.ep_count += 1                                                 # fresh initial states at every evaluation
        v.reset()
        if self._eval_init_inject is not None:
            v.set_internal(self._eval_init_inject)
        alive = torch.ones(lanes, device=self.device)
        total = torch.zeros(lanes, device=self.device)
        mean_ineq, mean_eq, max_ineq, max_eq = [torch.zeros(lanes, device=self.device) for _ in range(4)]
        if self.max_episode_steps:
            horizon = min(horizon, int(self.max_episode_steps))
        horizon = min(horizon, getattr(self.kernels, "episode_steps", horizon))     # EVOPF: one 24-hour day
        with torch.no_grad():
            for i in range(horizon):
                self._eval_action(v)
                v.ctrl.zero_()
                v.step(v.action, rows=self._eval_rows, cap_steps=1, auto_reset=False)
                row = self._eval_rows
                ineq = row[:, c["ineq_viol"][0]:c["ineq_viol"][1]].max(dim=1).values
                eq = row[:, c["eq_viol"][0]:c["eq_viol"][1]].abs().max(dim=1).values
                # finished lanes keep stepping (no reset) and may run off to inf / nan: select, never multiply
                live = alive > 0
                total = torch.where(live, total + row[:, c["reward"][0]], total)
                mean_ineq = torch.where(live, mean_ineq + (ineq - mean_ineq) / (i + 1), mean_ineq)
                mean_eq = torch.where(live, mean_eq + (eq - mean_eq) / (i + 1), mean_eq)
                max_ineq = torch.where(live, torch.maximum(max_ineq, ineq), max_ineq)
                max_eq = torch.where(live, torch.maximum(max_eq, eq), max_eq)
                alive = torch.where(live & (row[:, c["done"][0]] == 0), alive, torch.zeros_like(alive))
        out = []
        for x in (total, mean_ineq, mean_eq, max_ineq, max_eq):
            x = x.cpu().numpy().astype(np.float64)
            out += [x.mean(), x.std()]
        return tuple(out)

    def _print_eval(self, t, res):
        if self.dist.rank != 0 or not _env_int("RPO_VERBOSE", 1):
            return
        rmean, rstd, ineqmean, ineqstd, eqmean, eqstd, maxineqmean, maxineqstd, maxeqmean, maxeqstd = res
        print("\n============================")
        print(f"Eval: epoch {t}, rewards: {rmean:.4f}({rstd:.4f}), mean_ineq_viol: {ineqmean:.4f}({ineqstd:.4f}),"
              f" mean_eq_viol: {eqmean:.4f}({eqstd:.4f}), max_ineq_viol: {maxineqmean:.4f}({maxineqstd:.4f})"
              f" max_eq_viol: {maxeqmean:.4f}({maxeqstd:.4f})")
        print(f"lambda: {self.agent.lamb}, nju: {self.agent.nju}")
        print("============================\n")

    # ------------------------------------------------------------------------------------------ checkpoints
    def save(self, replay=True):
        """Checkpoint for an exact resume (the reference's save/load, agent/ddpg_pa.py:92-99, stores
        parameter generators and cannot be loaded): networks, targets, optimiser moments and step counters, multipliers,
        the env lanes with their episode bookkeeping, the device step counter that keys every Philox stream, and
        (``replay=True``) the filled part of this rank's replay shard.  One directory per rank when data-parallel."""
        d = self._ckpt_dir()
        os.makedirs(d, exist_ok=True)
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        self._check_flags()                                         # never checkpoint parameters of undefined origin
        self.agent.save_model(d)
        self._harvest()
        v, b = self.vec, self.buffer
        filled = min(self._t, b.capacity) * b.n_envs
        state = dict(t=self._t, updates=self._updates, seed=self.seed, num_envs=self.num_envs, world=self.dist.world,
                     internal=v.internal, obs=None if v.obs is v.internal else v.obs, ep_len=v.ep_len, ep_ret=v.ep_ret,
                     ep_count=v.ep_count, ctrl=v.ctrl, rows=b.rows[:filled].clone() if replay else None,
                     pending=self._pending, viol_steps=self.viol_steps, env_steps=self.env_steps,
                     capacity=b.capacity, row_floats=self.kernels.row_floats, algo=type(self).__name__,
                     env=self.kernels.name, projection_mode=self.projection_mode)
        torch.save(state, os.path.join(d, "trainer_state.pth"))

    def load(self, weights_only=False):
        """Restore a checkpoint.  ``weights_only=True``: networks, targets, optimiser state and multipliers only (to
        evaluate, or to start a fresh run from trained weights); otherwise the exact resume of ``save()``, which needs
        the replay shard: a checkpoint written with ``replay=False`` cannot resume training (the sampler would draw
        from a ring of zeros) and is refused."""
        d = self._ckpt_dir()
        path = os.path.join(d, "trainer_state.pth")
        st = torch.load(path, map_location=self.device, weights_only=False) if os.path.exists(path) else None
        if st is not None and not weights_only:
            # validate BEFORE anything is modified
            b = self.buffer
            mine = dict(seed=self.seed, num_envs=self.num_envs, world=self.dist.world, capacity=b.capacity,
                        row_floats=self.kernels.row_floats, algo=type(self).__name__, env=self.kernels.name,
                        projection_mode=self.projection_mode)
            diff = {k: (st[k], v) for k, v in mine.items() if k in st and st[k] != v}
            if diff:
                raise ValueError("checkpoint does not match this trainer (checkpoint, trainer): %s" % diff)
            if st["rows"] is None and st["t"] > 0:
                raise ValueError("checkpoint was saved with replay=False: training cannot resume from an empty replay "
                                 "ring; use load(weights_only=True) to restore the networks only")
        self.agent.load_model(d)
        if st is None or weights_only:
            return
        v, b = self.vec, self.buffer
        v.internal.copy_(st["internal"])
        if st["obs"] is not None:
            v.obs.copy_(st["obs"])
        v.ep_len.copy_(st["ep_len"])
        v.ep_ret.copy_(st["ep_ret"])
        v.ep_count.copy_(st["ep_count"])
        v.ctrl.copy_(st["ctrl"])
        v.stats.zero_()
        if st["rows"] is not None:
            # (the ring stride may differ from the checkpoint's: CartSafe rings went from 24 to 32 floats per row in round 4;
            #  a transition is the first `row_floats` floats of a ring row either way)
            w = min(st["rows"].shape[1], b.rows.shape[1])
            b.rows[:st["rows"].shape[0], :w].copy_(st["rows"][:, :w])
        self._t = self._harvested = int(st["t"])
        self._uclock_ok = False
        self._updates = int(st["updates"])
        b._steps_host = v.steps_host = self._t
        self._pending = list(st["pending"])
        self.viol_steps, self.env_steps = st["viol_steps"], st["env_steps"]
        self.agent.eps = max(self.eps, self.eps_start - self.decay_value * self._t)

    def _ckpt_dir(self):
        return self.work_dir if not self.dist.on else os.path.join(self.work_dir, "rank%d" % self.dist.rank)
