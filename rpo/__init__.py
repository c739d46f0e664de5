"""`rpo` -- the reference's import surface, served by the MI355X-native implementation in `rpo_amd`.

The reference's scripts do `from rpo.algo import RPODDPG`, `from rpo.env import *`,
`from rpo.utils.logger import Logger`, `import gym` (scripts/cart_exp.py:1-7).  Putting this repository on PYTHONPATH
makes those lines resolve to `rpo_amd` unchanged; `gym` falls back to rpo_amd.gym_shim when the real package is not
installed (all scripts import `rpo.*` before `gym`).
"""
import importlib
import sys

from rpo_amd import gym_shim

gym_shim.install()

_ALIASES = ("algo", "algo.agent", "algo.model", "algo.rpo_ddpg", "algo.rpo_sac", "env", "env.classic_control",
            "utils", "utils.logger", "utils.monitor", "utils.buffer", "utils.utils")
for _name in _ALIASES:
    sys.modules["rpo." + _name] = importlib.import_module("rpo_amd." + _name)
for _name in ("algo", "env", "utils"):
    globals()[_name] = sys.modules["rpo." + _name]
