"""Harness that imports the UNMODIFIED reference (``/root/reference``) in the build container.

Test infrastructure only.  It exists to generate the golden fixtures in this directory
(``make_golden.py``); nothing in the product, in ``-m gpu`` tests, in ``smoke()`` or in
``bench.py`` imports it, and it cannot run on the GPU box (``/root/reference`` is absent there).

The reference needs three things this image lacks:

* ``gym==0.19`` -> an in-memory stand-in exposing exactly the call sites on the path
  (``Env``, ``spaces.Box``, ``utils.seeding.np_random``, ``envs.registration.register``,
  ``make`` + ``TimeLimit``, ``logger.warn``);
* ``np.bool`` (removed from numpy >= 1.24, used by ``rpo/algo/agent/ddpg_pa.py:71``);
* ``pypower`` / ``igraph`` (only needed by EVOPF; ``rpo/env/__init__.py:2`` imports it
  unconditionally) -> ``rpo.env`` is pre-registered as an empty package and the classic-control
  sub-package is imported directly.
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "rpo"))


def _build_gym_stub():
    gym = types.ModuleType("gym")

    class Env(object):
        metadata = {}

        def close(self):
            pass

        def seed(self, seed=None):
            return [seed]

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            if shape is not None:
                low = np.broadcast_to(low, shape).copy()
                high = np.broadcast_to(high, shape).copy()
            self.low, self.high, self.dtype, self.shape = low, high, np.dtype(dtype), low.shape

        def contains(self, x):
            # gym 0.19 semantics: castable, same shape, inside [low, high]
            x = np.asarray(x)
            return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                        and np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            return np.random.uniform(self.low, self.high).astype(self.dtype)

    class TimeLimit(object):
        def __init__(self, env, max_episode_steps):
            self.env, self._max_episode_steps, self._elapsed_steps = env, max_episode_steps, None

        def __getattr__(self, name):
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self.env, name)

        def reset(self, **kw):
            self._elapsed_steps = 0
            return self.env.reset(**kw)

        def step(self, action):
            obs, reward, done, info = self.env.step(action)
            self._elapsed_steps += 1
            if self._elapsed_steps >= self._max_episode_steps:
                info["TimeLimit.truncated"] = not done
                done = True
            return obs, reward, done, info

        def close(self):
            return self.env.close()

    registry = {}

    def register(id, entry_point, max_episode_steps=None, **kw):
        registry[id] = (entry_point, max_episode_steps, kw)

    def make(id, **kwargs):
        entry_point, max_steps, kw = registry[id]
        mod_name, cls_name = entry_point.split(":")
        cls = getattr(importlib.import_module(mod_name), cls_name)
        env = cls(**{**kw.get("kwargs", {}), **kwargs})
        return TimeLimit(env, max_steps) if max_steps else env

    def np_random(seed=None):
        rs = np.random.RandomState(seed)
        return rs, seed

    gym.Env = Env
    gym.make = make
    gym.spaces = types.ModuleType("gym.spaces")
    gym.spaces.Box = Box
    gym.logger = types.ModuleType("gym.logger")
    gym.logger.warn = lambda *a, **k: None
    gym.utils = types.ModuleType("gym.utils")
    gym.utils.seeding = types.ModuleType("gym.utils.seeding")
    gym.utils.seeding.np_random = np_random
    gym.envs = types.ModuleType("gym.envs")
    gym.envs.registration = types.ModuleType("gym.envs.registration")
    gym.envs.registration.register = register
    gym.wrappers = types.ModuleType("gym.wrappers")
    gym.wrappers.TimeLimit = TimeLimit
    sys.modules.update({
        "gym": gym, "gym.spaces": gym.spaces, "gym.logger": gym.logger, "gym.utils": gym.utils,
        "gym.utils.seeding": gym.utils.seeding, "gym.envs": gym.envs,
        "gym.envs.registration": gym.envs.registration, "gym.wrappers": gym.wrappers})
    return gym


def load_reference():
    """Import the reference's ``rpo`` package; returns a namespace with the classes on the hot path."""
    if not reference_available():
        raise RuntimeError("the reference is only mounted in the build container")
    for name in list(sys.modules):
        if name == "rpo" or name.startswith("rpo."):
            raise RuntimeError("another 'rpo' package is already imported (%s); run in a clean process" % name)
    if not hasattr(np, "bool"):
        np.bool = bool
    gym = _build_gym_stub()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import rpo  # noqa: F401  (the reference's, first on sys.path)
        env_pkg = types.ModuleType("rpo.env")
        env_pkg.__path__ = [os.path.join(REFERENCE_ROOT, "rpo", "env")]
        sys.modules["rpo.env"] = env_pkg
        cc = importlib.import_module("rpo.env.classic_control")
        env_pkg.CartSafeEnv = cc.CartSafeEnv
        env_pkg.SpringPendulumEnv = cc.SpringPendulumEnv
        algo = importlib.import_module("rpo.algo")
        logger = importlib.import_module("rpo.utils.logger")
        buffer = importlib.import_module("rpo.utils.buffer")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    ns = types.SimpleNamespace(gym=gym, CartSafeEnv=cc.CartSafeEnv, SpringPendulumEnv=cc.SpringPendulumEnv,
                               RPODDPG=algo.RPODDPG, RPOSAC=algo.RPOSAC, Logger=logger.Logger,
                               ReplayBuffer=buffer.ReplayBuffer)
    return ns


def _build_pypower_standin():
    """Data-only stand-in for the un-vendored ``pypower`` (and an empty ``igraph``, used by render() only) so that the
    reference's ``rpo/env/electrical_grid/evopf.py`` can be imported here.  The numbers come from ``oracle/evopf.py``
    (public IEEE-14 tables + the published makeYbus algorithm), therefore fixtures produced through this stand-in
    cross-check the restatement of evopf.py's OWN logic and nothing about pypower: parity for EVOPF stays unpinned."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import evopf as oe
    pp = types.ModuleType("pypower")
    api = types.ModuleType("pypower.api")
    idx_bus = types.ModuleType("pypower.idx_bus")
    idx_gen = types.ModuleType("pypower.idx_gen")
    ppoption = types.ModuleType("pypower.ppoption")
    for k, v in dict(BUS_TYPE=1, PD=2, QD=3, GS=4, BS=5, VM=7, VA=8, VMAX=11, VMIN=12).items():
        setattr(idx_bus, k, v)
    for k, v in dict(PG=1, QG=2, QMAX=3, QMIN=4, MBASE=6, PMAX=8, PMIN=9).items():
        setattr(idx_gen, k, v)

    def case14():
        gen = np.zeros((oe.GEN.shape[0], 21))
        gen[:, :oe.GEN.shape[1]] = oe.GEN
        return {"version": "2", "baseMVA": oe.BASE_MVA, "bus": oe.BUS.copy(), "gen": gen, "branch": oe.BRANCH.copy(),
                "gencost": oe.GENCOST.copy()}

    class _Dense(object):
        def __init__(self, m):
            self.m = m

        def todense(self):
            return np.asmatrix(self.m)

    def makeYbus(base_mva, bus, branch):
        b, br = bus.copy(), branch.copy()
        b[:, 0] += 1                      # evopf.py:261-262 passes 0-based numbering
        br[:, [0, 1]] += 1
        return _Dense(oe.make_ybus(base_mva, b, br)), None, None

    def _unavailable(*a, **k):
        raise NotImplementedError("pypower stand-in: data only")

    api.case14, api.makeYbus, api.opf = case14, makeYbus, _unavailable
    ppoption.ppoption = _unavailable
    pp.api, pp.idx_bus, pp.idx_gen, pp.ppoption = api, idx_bus, idx_gen, ppoption
    sys.modules.update({"pypower": pp, "pypower.api": api, "pypower.idx_bus": idx_bus, "pypower.idx_gen": idx_gen,
                        "pypower.ppoption": ppoption, "igraph": types.ModuleType("igraph")})


def load_reference_evopf(ns=None):
    """Reference EVOPFEnv on the pypower stand-in (see above); returns ``ns`` with ``EVOPFEnv`` added."""
    ns = ns if ns is not None else load_reference()
    _build_pypower_standin()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        eg = importlib.import_module("rpo.env.electrical_grid")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    sys.modules["rpo.env"].EVOPFEnv = eg.EVOPFEnv
    ns.EVOPFEnv = eg.EVOPFEnv
    return ns
