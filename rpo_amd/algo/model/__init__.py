from .nets import (ActionEmbedding, BoxConstraint, DoubleValueAdd, DoubleValueCat, Dual, DualAdam,
                   GaussianSharedPolicy, SharedEmbedding, SharedPolicy, SharedValueAdd, SharedValueCat,
                   StateEmbedding)

__all__ = ["ActionEmbedding", "BoxConstraint", "DoubleValueAdd", "DoubleValueCat", "Dual", "DualAdam",
           "GaussianSharedPolicy", "SharedEmbedding", "SharedPolicy", "SharedValueAdd", "SharedValueCat",
           "StateEmbedding"]
