from .baselines import DDPG_LA, SAC_LA
from .rpo_ddpg import RPODDPG
from .rpo_sac import RPOSAC
from .trainer import NonFiniteError

__all__ = ["RPODDPG", "DDPG_LA", "RPOSAC", "SAC_LA", "NonFiniteError"]
