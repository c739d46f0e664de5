"""Registers the two classic-control ids with a 200-step TimeLimit (rpo/env/classic_control/__init__.py:6-16)."""
from ..base import gym
from .cartpole import CartSafeEnv
from .pendulum import SpringPendulumEnv

for _id, _cls in (("CartSafe-v0", "CartSafeEnv"), ("SpringPendulum-v0", "SpringPendulumEnv")):
    try:
        gym.envs.registration.register(id=_id, entry_point="rpo_amd.env.classic_control:%s" % _cls,
                                       max_episode_steps=200)
    except Exception:           # real gym raises on double registration (e.g. module reloaded)
        pass
