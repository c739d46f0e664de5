#!/usr/bin/env python
"""What bounds the 5-8 us layer launches of the EVOPF windows (csrc/mlp_gemm.h)?  Times, as dependent chains inside a hipGraph,
the forward of the EVOPF actor (57 -> 256 -> 256 -> 14) and of two concatenating critics in one set of launches
((57 | 43) -> 256 | 256 -> 256 -> 1), and the critic's backward rows, at batch 256 -- run it against timing-only builds
(RPO_HIP_LIBRARY=rpo_amd/csrc/librpo_hip_skipN.so, tools/probe/build_stream_variants.sh with KIND=gemm)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from rpo_amd import ops  # noqa: E402
from test_mlp_gpu import (StateEmbedding, ActionEmbedding, SharedPolicy, SharedValueCat, aligned_params, desc_for)  # noqa: E402

DEV = torch.device("cuda")
torch.manual_seed(0)
S, A, E, H, n = 57, 43, 256, 256, 256
actor = aligned_params(SharedPolicy(S, 14, StateEmbedding(S, E, H), E, H, 1, None))
crit = [aligned_params(SharedValueCat(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H)) for _ in range(2)]
da = desc_for(ops, actor, "actor14", S, 0, E, H)
dc = [desc_for(ops, m, "cat", S, A, E, H) for m in crit]
wide = torch.randn(n, S + A, device=DEV)
s, a = wide[:, :S], wide[:, S:]
out_a, x0a, h1a = torch.empty(n, 14, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
bufs = [(torch.empty(n, 1, device=DEV), torch.empty(n, 2 * E, device=DEV), torch.empty(n, H, device=DEV)) for _ in range(2)]
dout = torch.randn(n, 1, device=DEV)
dh, dx0 = torch.empty(n, H, device=DEV), torch.empty(n, 2 * E, device=DEV)
bench.spin_up(DEV, 1.0)
tag = os.path.basename(os.environ.get("RPO_HIP_LIBRARY", "librpo_hip.so"))
if len(sys.argv) > 1:                                            # e.g. mlp_gemm=0: the row-tile kernels
    kv = dict((k, int(v)) for k, _, v in (x.partition("=") for x in sys.argv[1].split(",")))
    ops.tuning(**kv).__enter__()
    tag += " " + sys.argv[1]
fa = bench.time_kernel(lambda: ops.mlp_forward(da, s, None, out_a, x0a, h1a))[0]
fc = bench.time_kernel(lambda: ops.mlp_forward_multi([(d, s, a) + b for d, b in zip(dc, bufs)]))[0]
bc = bench.time_kernel(lambda: ops.mlp_backward(dc[0], s, a, bufs[0][1], bufs[0][2], dout, dh, dx0, None, param_grads=False))[0]
print("%-24s actor forward (3 launches) %6.2f us | Q_targ || Q forward (3 launches) %6.2f us | critic backward rows (2 launches) %6.2f us"
      % (tag, fa, fc, bc), flush=True)
