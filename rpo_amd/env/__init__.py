"""`from rpo.env import *` surface of the reference (rpo/env/__init__.py:1-2).  EVOPF-v0 is not built yet: it needs
pypower's case14 tables, which are absent from the reference tree and from this image."""
from .classic_control import CartSafeEnv, SpringPendulumEnv

__all__ = ["CartSafeEnv", "SpringPendulumEnv"]
