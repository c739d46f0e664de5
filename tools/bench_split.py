#!/usr/bin/env python
"""Micro-benchmark (GPU box): row-tile forward (16 workgroups at batch 256) against the column-split forward
(128 workgroups per network) + the head kernel that stands for the consumer's prologue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from rpo_amd import ops
from rpo_amd.algo.model import ActionEmbedding, SharedValueAdd, StateEmbedding
from test_mlp_gpu import aligned_params, desc_for
from bench import time_kernel

DEV = "cuda"


def main():
    torch.manual_seed(0)
    S, A, E, H, n = 6, 2, 128, 256, 256
    nets = [aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H)) for _ in range(4)]
    ds = [desc_for(ops, c, "add", S, A, E, H) for c in nets]
    s, a = torch.randn(n, S, device=DEV), torch.randn(n, A, device=DEV)
    outs = [torch.empty(n, 1, device=DEV) for _ in range(4)]
    parts = [torch.zeros(8, n, 2, device=DEV) for _ in range(4)]
    x0, h1 = torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
    t = time_kernel(lambda: ops.mlp_forward(ds[0], s, a, outs[0], x0, h1))[0]
    print("tile  forward, 1 net  (16 WGs)        %7.2f us" % t)
    t = time_kernel(lambda: ops.mlp_forward_multi([(ds[i], s, a, outs[i], None, None) for i in range(2)]))[0]
    print("tile  forward, 2 nets (32 WGs)        %7.2f us" % t)
    t = time_kernel(lambda: ops.mlp_forward_multi([(ds[i], s, a, outs[i], None, None) for i in range(4)]))[0]
    print("tile  forward, 4 nets (64 WGs)        %7.2f us" % t)
    for k in (1, 2, 4):
        t = time_kernel(lambda: ops.mlp_forward_split([(ds[i], s, a, parts[i], x0 if i == 0 else None,
                                                        h1 if i == 0 else None) for i in range(k)]))[0]
        print("split forward, %d net(s) (%3d WGs)      %7.2f us" % (k, 128 * k, t))
    t = time_kernel(lambda: ops.mlp_split_head(ds[0], parts[0], outs[0]))[0]
    print("split head (consumer prologue stand-in) %5.2f us" % t)
    t = time_kernel(lambda: (ops.mlp_forward_split([(ds[0], s, a, parts[0], x0, h1)]), ops.mlp_split_head(ds[0], parts[0], outs[0])))[0]
    print("split forward + head, dependent pair  %7.2f us" % t)


if __name__ == "__main__":
    main()
