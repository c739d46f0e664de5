#!/usr/bin/env python
"""DESIGN.md 4.5 at unit level: ONE rpo_mlp_backward call of the large-batch critic (TD prologue -> [zero the split-K scratch] ->
rows kernel -> weights kernel -> ordered reduce) captured by torch.cuda.graph and replayed, against the same call launched
eagerly.  With the shipped library (scratch zeroed by a kernel) everything is clean.  With the old form and slices that carry
floats no kernel writes,

    git apply tools/probe/splitk_pad.patch
    bash tools/probe/build_flag_variant.sh memsetpad -DRPO_SPLITK_ZERO=0 -DRPO_SPLITK_PAD=64 ; git checkout rpo_amd/csrc/mlp_bwd.h
    DBG_SPLITK_PAD=64 RPO_HIP_LIBRARY=$PWD/rpo_amd/csrc/librpo_hip_memsetpad.so python tools/probe/memset_backward_probe.py

the replayed memset NODE leaves the 16-byte group {34572, 0, <low word>, <high word of a device pointer>} in every unwritten
position: `stride` and `gradmax`, bytes 32..47 of the arguments of the reduce launch three nodes behind it (the gradient still
equals the eager one here because this network's span has no padding inside it; in the trainer the critic's has three floats).
PROBE_PRE_FORWARDS=k puts k forward launches in front of the backward inside the graph (same result)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from rpo_amd import ops  # noqa: E402
import test_mlp_gpu as T  # noqa: E402

DEV = torch.device("cuda")
n, S, A, E, H = 1 << 20, 6, 2, 128, 256
torch.manual_seed(1)
se = T.StateEmbedding(S, E, H)
net = T.SharedValueAdd(S, A, se, T.ActionEmbedding(A, E, H), E, H)
T.aligned_params(net)
flat, total = T._flat_grads(net)
d = T.desc_for(ops, net, "add", S, A, E, H)
Z = 256
d.splitk = torch.zeros(Z * ((total + 3) // 4 * 4 + 64), device=DEV)
s, a = torch.randn(n, S, device=DEV), torch.randn(n, A, device=DEV)
out, x0, h1 = torch.empty(n, 1, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
ops.mlp_forward(d, s, a, out, x0, h1)
dh, dx0 = torch.empty(n, H, device=DEV), torch.empty(n, E, device=DEV)
gm = torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV)
q, qn1 = out.view(-1).clone(), torch.randn(n, device=DEV)
wide = torch.randn(n, 2, device=DEV)
reward, done = wide[:, 0:1], (wide[:, 1:2] > 0.5).float()
dq, parts = torch.empty(n, device=DEV), torch.zeros((n + 15) // 16, device=DEV)
td = ops.Td(q, qn1, None, None, reward, done, 0.0, 0.95, dq, parts)


out2 = torch.empty(n, 1, device=DEV)
PRE = int(os.environ.get("PROBE_PRE_FORWARDS", "0"))             # forward launches in front of the backward inside the graph


def body():
    for _ in range(PRE):
        ops.mlp_forward(d, s, a, out2, None, None)
    if PRE:
        ops.mlp_forward(d, s, a, out, x0, h1)
    ops.mlp_backward(d, s, a, x0, h1, None, dh, dx0, None, gradmax=gm, td=td)


flat.zero_()
body()
torch.cuda.synchronize()
eager = flat.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, capture_error_mode="thread_local"):
    body()
for r in range(6):
    flat.zero_()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(flat, eager)
    diff = torch.nonzero(flat != eager).view(-1)
    print("replay %d: gradient %s the eager launch's%s" % (r, "==" if same else "!=", "" if same else
          " at %d positions, e.g. %s" % (diff.numel(), [(int(i), float(flat[i]), float(eager[i])) for i in diff[:4]])), flush=True)
PAD = int(os.environ.get("DBG_SPLITK_PAD", "0"))                 # (a library built with -DRPO_SPLITK_PAD=<this>: floats per slice no kernel writes)
if PAD:
    lo = min(t.grad.data_ptr() for t in d.tensors.values() if t is not None)
    hi = max(t.grad.data_ptr() + 4 * t.numel() for t in d.tensors.values() if t is not None)
    span = (hi - lo) // 4
    stride = (span + 3) // 4 * 4 + PAD
    tail = d.splitk[:Z * stride].view(Z, stride)[:, stride - PAD:].contiguous().view(torch.int32)
    print("the %d floats behind each slice that no kernel writes, distinct 16-byte groups (int32): %s" % (
        PAD, torch.unique(tail.view(-1, 4), dim=0)[:6].tolist()))
ints = d.splitk.view(torch.int32)
small = (ints != 0) & (ints.abs() < 4096)
print("scratch words that look like small integers: %d; distinct 16-byte groups among them: %s" % (
    int(small.sum()), torch.unique(ints.view(-1, 4)[small.view(-1, 4).any(1)], dim=0)[:4].tolist()))
