from .pa_agents import Agent, PDDDPG_PA, PDSAC_PA

__all__ = ["Agent", "PDDDPG_PA", "PDSAC_PA"]
