#!/usr/bin/env python
"""Micro-benchmark of the MLP kernels (GPU box): per-launch time of forward / backward at the trainer's shapes,
measured as a hipGraph of back-to-back launches between two HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from rpo_amd import ops
from rpo_amd.algo.model import ActionEmbedding, SharedPolicy, SharedValueAdd, StateEmbedding
from test_mlp_gpu import aligned_params, desc_for
from bench import time_kernel

DEV = "cuda"


def main():
    torch.manual_seed(0)
    S, A, E, H = 6, 2, 128, 256
    actor = aligned_params(SharedPolicy(S, 1, StateEmbedding(S, E, H), E, H, 1, None))
    critic = aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H))
    da, dc = desc_for(ops, actor, "actor", S, 0, E, H), desc_for(ops, critic, "add", S, A, E, H)
    for n in (256, 4096, 65536):
        s, a = torch.randn(n, S, device=DEV), torch.randn(n, A, device=DEV)
        out = torch.empty(n, 1, device=DEV)
        x0, h1 = torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
        dh, dx0, dA = torch.empty(n, H, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, A, device=DEV)
        dout = torch.randn(n, 1, device=DEV)
        flops_f = 2.0 * n * (S * E + E * H + H)
        t = time_kernel(lambda: ops.mlp_forward(da, s, None, out), reps=50)[0]
        print("n=%6d actor  fwd (infer)   %8.2f us  %7.2f TFLOP/s" % (n, t, flops_f / t * 1e-6))
        t = time_kernel(lambda: ops.mlp_forward(dc, s, a, out, x0, h1), reps=50)[0]
        print("n=%6d critic fwd (save)    %8.2f us" % (n, t))
        ops.mlp_forward(dc, s, a, out, x0, h1)
        t = time_kernel(lambda: ops.mlp_backward(dc, s, a, x0, h1, dout, dh, dx0, dA), reps=50)[0]
        print("n=%6d critic bwd (2 launches) %6.2f us" % (n, t))
        t = time_kernel(lambda: ops.mlp_backward(dc, s, a, x0, h1, dout, dh, dx0, dA, param_grads=False), reps=50)[0]
        print("n=%6d critic bwd rows only %8.2f us" % (n, t))
        if n <= 4096:
            lin = torch.nn.Linear(E, H).to(DEV)
            xx = torch.randn(n, E, device=DEV)
            t = time_kernel(lambda: lin(xx), reps=50)[0]
            print("n=%6d torch Linear(128,256) %7.2f us (rocBLAS)" % (n, t))


if __name__ == "__main__":
    main()
