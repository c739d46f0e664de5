"""Registers EVOPF-v0 without a TimeLimit (rpo/env/electrical_grid/__init__.py:1-7): an episode is one 24-hour day."""
from ..base import gym
from .evopf import EVOPFEnv

try:
    gym.envs.registration.register(id="EVOPF-v0", entry_point="rpo_amd.env.electrical_grid:EVOPFEnv")
except Exception:               # real gym raises on double registration
    pass
