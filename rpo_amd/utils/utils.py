"""Small helpers that are part of the reference's public surface (rpo/utils/utils.py)."""
from collections import namedtuple

# ReplayBuffer key spec, e.g. Type((state_dim,), np.float32) (rpo/algo/agent/ddpg_pa.py:70-71)
Type = namedtuple("Type", ["shape", "dtype"])


def max_grad(net):
    """Largest gradient entry over a module's parameters (rpo/utils/utils.py:5-10); debug print of train()."""
    best = 0
    for p in net.parameters():
        if p.grad is not None:
            best = max(best, p.grad.max())
    return best
