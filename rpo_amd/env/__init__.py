"""`from rpo.env import *` surface of the reference (rpo/env/__init__.py:1-2); importing it registers the gym ids."""
from .classic_control import CartSafeEnv, SpringPendulumEnv
from .electrical_grid import EVOPFEnv

__all__ = ["CartSafeEnv", "SpringPendulumEnv", "EVOPFEnv"]
