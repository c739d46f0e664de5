"""Minimal stand-in for the parts of ``gym`` (0.19 API) the RPO scripts and envs touch.

The reference pins ``gym==0.19.0`` (README.md:18) and uses: ``gym.Env``, ``gym.spaces.Box`` (+ ``contains``),
``gym.utils.seeding.np_random``, ``gym.envs.registration.register(id, entry_point, max_episode_steps)``,
``gym.make`` (-> ``TimeLimit`` wrapper that sets ``done`` at truncation) and ``gym.logger.warn``.
``install()`` registers this module as ``gym`` ONLY when no real gym can be imported, so the reference's scripts
(`import gym; gym.make("CartSafe-v0")`) run unchanged on a box without the package.
"""
import importlib
import sys
import types

import numpy as np


class Env(object):
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    action_space = None
    observation_space = None

    def step(self, action):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def render(self, mode="human"):
        raise NotImplementedError

    def close(self):
        pass

    def seed(self, seed=None):
        return [seed]

    @property
    def unwrapped(self):
        return self


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None:
            low = np.broadcast_to(low, shape).copy()
            high = np.broadcast_to(high, shape).copy()
        self.low, self.high = low, high
        self.dtype = np.dtype(dtype)
        self.shape = low.shape
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape and np.all(x >= self.low)
                    and np.all(x <= self.high))

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space
        self.metadata = getattr(env, "metadata", {})

    def __getattr__(self, name):
        if name.startswith("_"):          # keeps copy.deepcopy / pickle from recursing into the wrapped env
            raise AttributeError("attempted to get missing private attribute '%s'" % name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def render(self, mode="human", **kwargs):
        return self.env.render(mode, **kwargs)

    def close(self):
        return self.env.close()

    def seed(self, seed=None):
        return self.env.seed(seed)


class TimeLimit(Wrapper):
    """gym 0.19 semantics: ``done = True`` when the step budget is exhausted (truncation is folded into done)."""

    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class _Spec(object):
    def __init__(self, id, entry_point, max_episode_steps, kwargs):
        self.id, self.entry_point, self.max_episode_steps, self.kwargs = id, entry_point, max_episode_steps, kwargs


_REGISTRY = {}


def register(id, entry_point=None, max_episode_steps=None, kwargs=None, **_ignored):
    _REGISTRY[id] = _Spec(id, entry_point, max_episode_steps, kwargs or {})


def make(id, **kwargs):
    if id not in _REGISTRY:
        raise KeyError("No registered env with id: %s" % id)
    spec = _REGISTRY[id]
    if callable(spec.entry_point):
        cls = spec.entry_point
    else:
        mod_name, attr = spec.entry_point.split(":")
        cls = getattr(importlib.import_module(mod_name), attr)
    env = cls(**{**spec.kwargs, **kwargs})
    env.spec = spec
    if spec.max_episode_steps is not None:
        env = TimeLimit(env, max_episode_steps=spec.max_episode_steps)
    return env


def np_random(seed=None):
    """gym.utils.seeding.np_random: (generator, seed).  gym 0.19 hashes the seed before seeding a RandomState; that
    mapping is third-party behaviour the reference never pins (its scripts never seed the env)."""
    return np.random.RandomState(seed), seed


def _as_module():
    gym = types.ModuleType("gym")
    gym.__doc__ = __doc__
    gym.__version__ = "0.19.0+rpo_amd.shim"
    gym.Env, gym.Wrapper, gym.make, gym.register = Env, Wrapper, make, register
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    logger = types.ModuleType("gym.logger")
    logger.warn = lambda *a, **k: None
    logger.info = lambda *a, **k: None
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = np_random
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register, registration.make = register, make
    envs.registration = registration
    wrappers = types.ModuleType("gym.wrappers")
    wrappers.TimeLimit = TimeLimit
    gym.spaces, gym.logger, gym.utils, gym.envs, gym.wrappers = spaces, logger, utils, envs, wrappers
    return {"gym": gym, "gym.spaces": spaces, "gym.logger": logger, "gym.utils": utils,
            "gym.utils.seeding": seeding, "gym.envs": envs, "gym.envs.registration": registration,
            "gym.wrappers": wrappers}


def install():
    """Make ``import gym`` work.  Returns the module that will be used (the real package when it is importable)."""
    if "gym" in sys.modules:
        return sys.modules["gym"]
    try:
        return importlib.import_module("gym")
    except ImportError:
        mods = _as_module()
        sys.modules.update(mods)
        return mods["gym"]
