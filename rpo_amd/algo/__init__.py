from .rpo_ddpg import RPODDPG
from .rpo_sac import RPOSAC

__all__ = ["RPODDPG", "RPOSAC"]
