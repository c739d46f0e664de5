"""rpo_amd -- MI355X-native hot path of Reduced Policy Optimization (RPO).

Host side: Python mirror of the reference's `rpo.algo` trainers and `rpo.env` constraint-oracle interface
(vectorised over N env instances resident in HBM).  Device side: hand-written HIP kernels for gfx950 behind the
C ABI of include/rpo_hip.h (rpo_amd/csrc).  PyTorch-ROCm provides device memory, streams, the actor/critic MLPs
and torch.distributed (RCCL).  There is no CPU fallback: GPU entry points raise when librpo_hip.so is missing.
"""
__version__ = "0.1.0"
