#!/usr/bin/env python
"""The GEMM-shaped launches of the large-batch update at n rows (default 2^20), each timed as a hipGraph of back-to-back
launches between two HIP events and priced against the 157.3 TFLOP/s f32 MFMA peak; with `check` the large-n kernels are
first compared bit for bit with the 16-row tile kernels on a slice.

    python tools/probe_mlp_large.py [n=1048576] [check]
    rocprofv3 --kernel-trace --stats ... -- python3 tools/probe_mlp_large.py 1048576 once     # one launch of each (profilers)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402

PEAK = 157.3


def main():
    from rpo_amd import ops
    from rpo_amd.algo.model import ActionEmbedding, SharedPolicy, SharedValueAdd, StateEmbedding
    from test_mlp_gpu import aligned_params, desc_for
    from bench import time_kernel
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1 << 20
    mode = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].isdigit() else "time"
    dev = "cuda"
    torch.manual_seed(0)
    S, A, E, H = 6, 2, 128, 256
    actor = aligned_params(SharedPolicy(S, 1, StateEmbedding(S, E, H), E, H, 1, None))
    critics = [aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H)) for _ in range(2)]
    da = desc_for(ops, actor, "actor", S, 0, E, H)
    dcs = [desc_for(ops, c, "add", S, A, E, H) for c in critics]
    batch = torch.randn(n, 24, device=dev)
    s, a, s2, a2 = batch[:, 0:6], batch[:, 6:8], batch[:, 8:14], batch[:, 14:16]
    out = [torch.empty(n, 1, device=dev) for _ in range(2)]
    x0 = [torch.empty(n, E, device=dev) for _ in range(2)]
    h1 = [torch.empty(n, H, device=dev) for _ in range(2)]
    f_a = 2.0 * n * (S * E + E * H + H)
    f_c = 2.0 * n * ((S + A) * E + E * H + H)

    if mode == "check":
        m = min(n, 8192)
        for kw in (dict(fwd_stream=1, fwd_stream_waves=12), dict(fwd_stream=1, fwd_stream_waves=16), dict(fwd_stream=0)):
            for t in out + x0 + h1:
                t.fill_(float("nan"))
            with ops.tuning(**kw):
                ops.mlp_forward(dcs[0], s, a, out[0], x0[0], h1[0])
                ops.mlp_forward_multi([(dcs[0], s2, a2, out[1], x0[1], h1[1])])
            ro, rx, rh = torch.empty(m, 1, device=dev), torch.empty(m, E, device=dev), torch.empty(m, H, device=dev)
            ops.mlp_forward(dcs[0], s[:m], a[:m], ro, rx, rh)
            ok = [torch.equal(ro, out[0][:m]), torch.equal(rx, x0[0][:m]), torch.equal(rh, h1[0][:m])]
            tail = n - 37
            to, tx, th = torch.empty(37, 1, device=dev), torch.empty(37, E, device=dev), torch.empty(37, H, device=dev)
            ops.mlp_forward(dcs[0], s2[tail:], a2[tail:], to, tx, th)
            ok += [torch.equal(to, out[1][tail:]), torch.equal(tx, x0[1][tail:]), torch.equal(th, h1[1][tail:])]
            print(kw, "bit-equal to the 16-row tiles (out, x0, h1; head of the batch | tail through the multi entry):", ok, flush=True)
            assert all(ok)
        return

    def report(name, fn, flops, reps=10):
        if mode == "once" and False:
            pass
        if mode == "once":
            fn()
            torch.cuda.synchronize()
            return
        us = time_kernel(fn, reps=reps)[0]
        print("%-44s %9.1f us  %7.2f TFLOP/s  %.3f of the f32 MFMA peak" % (name, us, flops / us * 1e-6, flops / us * 1e-6 / PEAK), flush=True)

    for waves in (12, 16):
        with ops.tuning(fwd_stream=1, fwd_stream_waves=waves):
            report("forward actor (nothing saved), stream/%d" % waves, lambda: ops.mlp_forward(da, s, None, out[0]), f_a)
            report("forward critic (x0, h1 saved), stream/%d" % waves, lambda: ops.mlp_forward(dcs[0], s, a, out[0], x0[0], h1[0]), f_c)
            report("forward_multi Q_targ || Q, stream/%d" % waves, lambda: ops.mlp_forward_multi(
                [(dcs[0], s2, a2, out[0], x0[0], h1[0]), (dcs[1], s, a, out[1], x0[1], h1[1])]), 2 * f_c)
    if mode == "fwd":
        return
    with ops.tuning(fwd_stream=0):
        report("forward actor (nothing saved), 64-row tiles", lambda: ops.mlp_forward(da, s, None, out[0]), f_a)
        report("forward critic (x0, h1 saved), 64-row tiles", lambda: ops.mlp_forward(dcs[0], s, a, out[0], x0[0], h1[0]), f_c)
        report("forward_multi Q_targ || Q, 64-row tiles", lambda: ops.mlp_forward_multi(
            [(dcs[0], s2, a2, out[0], x0[0], h1[0]), (dcs[1], s, a, out[1], x0[1], h1[1])]), 2 * f_c)
    # backward: rows + weights in one pass (split-K scratch as the trainer sets it up), and the rows-only form of the policy step
    ops.mlp_forward(dcs[0], s, a, out[0], x0[0], h1[0])
    dout = torch.randn(n, 1, device=dev)
    dh, dx0, dA = torch.empty(n, H, device=dev), torch.empty(n, E, device=dev), torch.empty(n, A, device=dev)
    # ONE flat gradient buffer (like agent/flat.py) + the split-K scratch the trainer sets up (FusedNets.enable_splitk)
    net = critics[0]
    total = sum((q.numel() + 3) // 4 * 4 for q in net.parameters())
    flat = torch.zeros(total + 8, device=dev)
    off = (-flat.data_ptr() // 4) % 4
    for q in net.parameters():
        q.grad = flat[off:off + q.numel()].view(q.shape)
        off += (q.numel() + 3) // 4 * 4
    dcs[0] = desc_for(ops, net, "add", S, A, E, H)
    dcs[0].splitk = torch.zeros(max(2, min(256, n // 4096)) * total, device=dev)
    report("backward critic (param grads)", lambda: ops.mlp_backward(dcs[0], s, a, x0[0], h1[0], dout, dh, dx0, None), 2 * f_c, reps=5)
    # (the dQ/da pass of the policy step: dh -> dx0 -> da, no parameter gradients: one forward's worth of flops)
    report("backward critic rows only (dQ/da)", lambda: ops.mlp_backward(dcs[0], s, a, x0[0], h1[0], dout, dh, dx0, dA, param_grads=False), f_c, reps=5)


if __name__ == "__main__":
    main()
