"""Hand-written f32-MFMA MLP kernels (rpo_mlp_forward / rpo_mlp_backward) against the PyTorch modules they replace.

Tolerance: both sides are float32; the MFMA is an exact fmaf chain, so only summation order differs from rocBLAS /
torch reductions: forward 1e-5 relative, gradients 2e-5 relative to the largest entry of each gradient tensor.
"""
import numpy as np
import pytest
import torch

from rpo_amd.algo.model import (ActionEmbedding, BoxConstraint, DoubleValueAdd, GaussianSharedPolicy, SharedPolicy,
                                SharedValueAdd, SharedValueCat, StateEmbedding)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def aligned_params(module):
    """Re-home the parameters 16-byte aligned on the GPU (what FlatParams does) and give every one a zero .grad."""
    for p in module.parameters():
        buf = torch.zeros(p.numel() + 8, device=DEV)
        off = (-buf.data_ptr() // 4) % 4
        v = buf[off:off + p.numel()].view(p.shape)
        v.copy_(p.data)
        p.data = v
        g = torch.zeros(p.numel() + 8, device=DEV)
        off = (-g.data_ptr() // 4) % 4
        p.grad = g[off:off + p.numel()].view(p.shape)
    return module


def desc_for(ops, net, kind, S, A, E, H):
    if kind in ("actor", "actor14"):
        t = dict(Ws=net.state_embed.embeds[0].weight, bs=net.state_embed.embeds[0].bias, W0=net.affines[0].weight,
                 b0=net.affines[0].bias, W1=net.affines[1].weight, b1=net.affines[1].bias)
        return ops.MlpDesc(t, S, 0, E, H, 1, False, head_dim=14 if kind == "actor14" else 1)
    if kind == "gauss14":
        t = dict(Ws=net.state_embed.embeds[0].weight, bs=net.state_embed.embeds[0].bias, W0=net.affines[0].weight,
                 b0=net.affines[0].bias, W1=net.affine_mean.weight, b1=net.affine_mean.bias,
                 W1b=net.affine_log_std.weight, b1b=net.affine_log_std.bias)
        return ops.MlpDesc(t, S, 0, E, H, 2, False, head_dim=14)
    if kind == "gauss":
        t = dict(Ws=net.state_embed.embeds[0].weight, bs=net.state_embed.embeds[0].bias, W0=net.affines[0].weight,
                 b0=net.affines[0].bias, W1=net.affine_mean.weight, b1=net.affine_mean.bias,
                 W1b=net.affine_log_std.weight, b1b=net.affine_log_std.bias)
        return ops.MlpDesc(t, S, 0, E, H, 2, False)
    t = dict(Ws=net.state_embed.embeds[0].weight, bs=net.state_embed.embeds[0].bias, Wa=net.action_embed.embeds[0].weight,
             ba=net.action_embed.embeds[0].bias, W0=net.affines[0].weight, b0=net.affines[0].bias,
             W1=net.affines[1].weight, b1=net.affines[1].bias)
    return ops.MlpDesc(t, S, A, E, H, 1, kind == "cat")


def gauss_raw(net, s):
    x = net.state_embed(s)
    for affine in net.affines:
        x = affine(torch.relu(x))
    x = torch.relu(x)
    return torch.cat([net.affine_mean(x), net.affine_log_std(x)], dim=1)


CASES = [("actor", 6, 0, 128, 256, 4096), ("actor", 5, 0, 128, 256, 250), ("add", 6, 2, 128, 256, 256),
         ("add", 5, 2, 128, 256, 77), ("cat", 57, 43, 256, 256, 200), ("gauss", 5, 0, 128, 256, 256),
         ("add", 6, 2, 256, 256, 512),
         # multi-output heads (MFMA head): EVOPF-v0 actor (57 -> 256 -> 256 -> 14) and its SAC form (2 x 14)
         ("actor14", 57, 0, 256, 256, 1000), ("actor14", 57, 0, 256, 256, 256), ("gauss14", 57, 0, 256, 256, 77),
         ("actor14", 6, 0, 128, 256, 300),
         # the matrix-core first layer (S <= 6, A <= 4: bias slots | state | action as three k steps) at its edges, and the
         # vector form just beyond them
         ("add", 3, 1, 128, 256, 100), ("actor", 1, 0, 128, 256, 33), ("add", 6, 4, 128, 256, 64), ("add", 7, 2, 128, 256, 48),
         ("add", 6, 5, 128, 256, 48), ("gauss", 2, 0, 128, 256, 17)]


@pytest.mark.parametrize("kind,S,A,E,H,n", CASES)
def test_mlp_forward_backward_match_torch(kind, S, A, E, H, n):
    from rpo_amd import ops
    torch.manual_seed(S * 100 + n)
    se = StateEmbedding(S, E, H)
    if kind == "actor":
        net = SharedPolicy(S, 1, se, E, H, 1, None)
    elif kind == "actor14":
        net = SharedPolicy(S, 14, se, E, H, 1, None)
    elif kind == "gauss14":
        net = GaussianSharedPolicy(S, 14, se, E, H, 1, None)
    elif kind == "gauss":
        net = GaussianSharedPolicy(S, 1, se, E, H, 1, None)
    else:
        cls = SharedValueCat if kind == "cat" else SharedValueAdd
        net = cls(S, A, se, ActionEmbedding(A, E, H), E, H)
    aligned_params(net)
    assert ops.mlp_supported(E, H, kind == "cat")
    d = desc_for(ops, net, kind, S, A, E, H)
    # inputs as strided column views of a wider batch matrix, like the trainer's gathered rows
    wide = torch.randn(n, S + A + 7, device=DEV)
    s = wide[:, 3:3 + S]
    a = wide[:, 3 + S:3 + S + A] if A else None
    a_t = a.clone().requires_grad_() if A else None
    if kind in ("actor", "actor14"):
        ref = net(s)
    elif kind in ("gauss", "gauss14"):
        ref = gauss_raw(net, s)
    else:
        ref = net(s, a_t)
    n_out = ref.shape[1]
    out = torch.empty(n, n_out, device=DEV)
    x0 = torch.empty(n, d.ein, device=DEV)
    h1 = torch.empty(n, H, device=DEV)
    ops.mlp_forward(d, s, a, out, x0, h1)
    np.testing.assert_allclose(out.cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=2e-6)
    out2 = torch.empty(n, n_out, device=DEV)
    if E == 256:
        # wide networks run layer by layer (mlp_gemm.h) whenever the caller brings buffers for the pre-activations -- the
        # trainer always does; without them the row-tile kernel answers, to float32 round-off of the other summation order
        ops.mlp_forward(d, s, a, out2, torch.empty_like(x0), torch.empty_like(h1))
        assert torch.equal(out, out2)
        # hidden layer / heads with the k range split over the four waves of a workgroup (default, csrc/mlp_gemm.h) vs the
        # one-chain kernel: the same sums in another (fixed) order
        x0b, h1b = torch.full_like(x0, float("nan")), torch.full_like(h1, float("nan"))
        with ops.tuning(gemm_ksplit=0):
            ops.mlp_forward(d, s, a, out2, x0b, h1b)
        assert torch.equal(x0, x0b)                           # (first layers: K <= 64, never split)
        np.testing.assert_allclose(h1.cpu().numpy(), h1b.cpu().numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(out.cpu().numpy(), out2.cpu().numpy(), rtol=1e-5, atol=2e-6)
        ops.mlp_forward(d, s, a, out2)
        np.testing.assert_allclose(out2.cpu().numpy(), out.cpu().numpy(), rtol=1e-5, atol=2e-6)
    else:
        ops.mlp_forward(d, s, a, out2)                   # inference form (nothing saved) gives the same bits
        assert torch.equal(out, out2)
        # the first layer as three matrix-core steps (S <= 6, A <= 4; round 5) vs the vector form: the same fmaf chain, so the
        # pre-activations, the hidden layer and the outputs are the same BITS (csrc/mlp_tile.h tile_l1_mfma)
        x0v, h1v = torch.full_like(x0, float("nan")), torch.full_like(h1, float("nan"))
        with ops.tuning(l1_mfma=0):
            ops.mlp_forward(d, s, a, out2, x0v, h1v)
        assert torch.equal(x0, x0v) and torch.equal(h1, h1v) and torch.equal(out, out2)
        assert float(x0.abs().max()) > 0

    dout = torch.randn(n, n_out, device=DEV) / n
    ref.backward(dout)
    want = {k: (None if t is None else t.grad.clone()) for k, t in d.tensors.items()}
    for t in d.tensors.values():
        if t is not None:
            t.grad.zero_()
    dh = torch.empty(n, H, device=DEV)
    dx0 = torch.empty(n, d.ein, device=DEV)
    da = torch.empty(n, A, device=DEV) if A else None
    gm = torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV)         # the workgroups spread over 16 slots: the norm is the max
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, gradmax=gm)
    # the weights pass also leaves the inf-norm of what it wrote (clip_grad_norm_ without a second pass): exact
    assert float(gm.max()) == max(float(t.grad.abs().max()) for t in d.tensors.values() if t is not None)
    for k, t in d.tensors.items():
        if t is None:
            continue
        w = want[k].cpu().numpy()
        np.testing.assert_allclose(t.grad.cpu().numpy(), w, rtol=2e-5, atol=2e-5 * np.abs(w).max() + 1e-12, err_msg=k)
    if A:
        w = a_t.grad.cpu().numpy()
        np.testing.assert_allclose(da.cpu().numpy(), w, rtol=2e-5, atol=2e-5 * np.abs(w).max())
    # accumulation semantics (+=) and the reduced modes
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da)
    np.testing.assert_allclose(d.tensors["W0"].grad.cpu().numpy(), 2 * want["W0"].cpu().numpy(), rtol=3e-5,
                               atol=4e-5 * np.abs(want["W0"].cpu().numpy()).max())
    for t in d.tensors.values():
        if t is not None:
            t.grad.zero_()
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, param_grads=True, first_layer_state_only=True)
    np.testing.assert_allclose(d.tensors["Ws"].grad.cpu().numpy(), want["Ws"].cpu().numpy(), rtol=2e-5,
                               atol=2e-5 * np.abs(want["Ws"].cpu().numpy()).max())
    assert float(d.tensors["W0"].grad.abs().max()) == 0.0
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, param_grads=False)
    assert float(d.tensors["W0"].grad.abs().max()) == 0.0


@pytest.mark.parametrize("kind,S,A,E,H,n", [("add", 6, 2, 128, 256, 40000), ("actor", 6, 0, 128, 256, 16384),
                                            ("cat", 57, 43, 256, 256, 20011), ("gauss", 5, 0, 128, 256, 70000)])
def test_split_k_weights_pass_large_batch(kind, S, A, E, H, n):
    """Batches of >= RPO_SPLITK_FROM rows with a scratch buffer in rpo_mlp_grad: the weights pass splits the batch over up
    to 64 slices (mlp_bwd_weights_splitk_kernel + splitk_reduce_kernel) instead of walking it with ~50 workgroups.  Same
    gradients as torch (2e-5 of each tensor's largest entry, like the plain pass), the inf-norm of what was written, += onto
    existing gradients, bitwise reproducible from call to call, and untouched when the batch is small."""
    from rpo_amd import ops
    torch.manual_seed(n)
    se = StateEmbedding(S, E, H)
    if kind == "actor":
        net = SharedPolicy(S, 1, se, E, H, 1, None)
    elif kind == "gauss":
        net = GaussianSharedPolicy(S, 1, se, E, H, 1, None)
    else:
        net = (SharedValueCat if kind == "cat" else SharedValueAdd)(S, A, se, ActionEmbedding(A, E, H), E, H)
    aligned_params(net)
    # ONE flat gradient buffer (like agent/flat.py): the scratch holds slices of the span from its first to its last tensor
    total = sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
    flat = torch.zeros(total + 8, device=DEV)
    off = (-flat.data_ptr() // 4) % 4
    for p in net.parameters():
        p.grad = flat[off:off + p.numel()].view(p.shape)
        off += (p.numel() + 3) // 4 * 4
    d = desc_for(ops, net, kind, S, A, E, H)
    s = torch.randn(n, S, device=DEV)
    a = torch.randn(n, A, device=DEV) if A else None
    a_t = a.clone().requires_grad_() if A else None
    ref = net(s) if kind == "actor" else (gauss_raw(net, s) if kind == "gauss" else net(s, a_t))
    n_out = ref.shape[1]
    out, x0, h1 = torch.empty(n, n_out, device=DEV), torch.empty(n, d.ein, device=DEV), torch.empty(n, H, device=DEV)
    ops.mlp_forward(d, s, a, out, x0, h1)
    dout = torch.randn(n, n_out, device=DEV) / n
    ref.backward(dout)
    want = {k: t.grad.clone() for k, t in d.tensors.items() if t is not None}
    flat.zero_()
    dh, dx0 = torch.empty(n, H, device=DEV), torch.empty(n, d.ein, device=DEV)
    da = torch.empty(n, A, device=DEV) if A else None
    gm = torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV)
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, gradmax=gm)         # no scratch: the one-owner-per-batch pass
    plain = flat.clone()
    d.splitk = torch.full((64 * total,), 7.0, device=DEV)                    # (dirty: the launch zeroes what it uses)
    runs = []
    for _ in range(2):
        flat.zero_()
        gm.zero_()
        ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, gradmax=gm)
        runs.append(flat.clone())
    assert torch.equal(runs[0], runs[1])                                      # fixed orders: bitwise reproducible
    assert not torch.equal(runs[0], plain)                                    # (another association than the plain pass)
    assert float(gm.max()) == float(flat.abs().max())
    # The yardstick is the plain pass (pinned against torch at the batch sizes of the update, above): at tens of thousands of
    # rows torch's own float32 GEMMs are off by up to 4e-3 of a tensor's largest entry against float64 (measured: dW0 at
    # 40 000 rows; both passes here are within 3e-6 of float64 there), so torch only bounds the comparison from afar
    view = {k: (t.grad.data_ptr() - flat.data_ptr()) // 4 for k, t in d.tensors.items() if t is not None}
    for k, t in d.tensors.items():
        if t is None:
            continue
        ref_k = plain[view[k]:view[k] + t.numel()].view(t.shape).cpu().numpy()
        np.testing.assert_allclose(t.grad.cpu().numpy(), ref_k, rtol=1e-5, atol=1e-5 * np.abs(ref_k).max() + 1e-12, err_msg=k)
        w = want[k].cpu().numpy()
        np.testing.assert_allclose(t.grad.cpu().numpy(), w, rtol=0, atol=1e-2 * np.abs(w).max() + 1e-12, err_msg=k)
    ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da)                     # += onto the existing gradients
    np.testing.assert_allclose(flat.cpu().numpy(), 2 * runs[0].cpu().numpy(), rtol=1e-5, atol=2e-5 * float(runs[0].abs().max()))
    # a small batch ignores the scratch: the bits of the plain pass
    m = 300
    flat.zero_()
    ops.mlp_backward(d, s[:m], None if a is None else a[:m], x0[:m], h1[:m], dout[:m].contiguous(), dh[:m], dx0[:m],
                     None if da is None else da[:m])
    small = flat.clone()
    d.splitk = None
    flat.zero_()
    ops.mlp_backward(d, s[:m], None if a is None else a[:m], x0[:m], h1[:m], dout[:m].contiguous(), dh[:m], dx0[:m],
                     None if da is None else da[:m])
    assert torch.equal(small, flat)


def test_mlp_tanh_box_epilogue():
    from rpo_amd import ops
    torch.manual_seed(1)
    box = BoxConstraint(np.array([-10.0], dtype=np.float32), np.array([10.0], dtype=np.float32), device=DEV)
    net = aligned_params(SharedPolicy(6, 1, StateEmbedding(6, 128, 256), 128, 256, 1, box))
    d = desc_for(ops, net, "actor", 6, 0, 128, 256)
    s = torch.randn(1000, 6, device=DEV) * 3
    out = torch.empty(1000, 1, device=DEV)
    ops.mlp_forward(d, s, None, out, out_mode=1, scale=10.0, base=0.0)
    np.testing.assert_allclose(out.cpu().numpy(), net(s).detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_forward_multi_large_n_equals_single_launches():
    """From 12288 rows both entry points move to 64-row tiles: same bits as the single launches."""
    from rpo_amd import ops
    torch.manual_seed(5)
    S, A, E, H, n = 6, 2, 128, 256, 20011
    nets = [aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H)) for _ in range(3)]
    descs = [desc_for(ops, m, "add", S, A, E, H) for m in nets]
    wide = torch.randn(n, 16, device=DEV)
    ins = [(wide[:, 0:6], wide[:, 6:8]), (wide[:, 8:14], wide[:, 14:16]), (wide[:, 0:6], wide[:, 14:16])]
    new = lambda: (torch.empty(n, 1, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV))
    single, multi = [new() for _ in range(3)], [new() for _ in range(3)]
    for d, (s, a), m in zip(descs, ins, single):
        ops.mlp_forward(d, s, a, *m)
    ops.mlp_forward_multi([(d, s, a) + m for d, (s, a), m in zip(descs, ins, multi)])
    for one, many in zip(single, multi):
        for x, y in zip(one, many):
            assert torch.equal(x, y)
    ref = nets[1](ins[1][0], ins[1][1]).detach()
    np.testing.assert_allclose(multi[1][0].cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_forward_multi_equals_single_launches():
    """rpo_mlp_forward_multi (gridDim.y = network): four same-shaped critics on two different input pairs, bitwise equal to
    four rpo_mlp_forward launches, pre-activations included; and the TD prologue of rpo_mlp_backward (rpo_td) against
    rpo_td_huber + rpo_mlp_backward."""
    from rpo_amd import ops
    torch.manual_seed(3)
    S, A, E, H, n = 6, 2, 128, 256, 200
    nets = [aligned_params(SharedValueAdd(S, A, StateEmbedding(S, E, H), ActionEmbedding(A, E, H), E, H)) for _ in range(4)]
    descs = [desc_for(ops, m, "add", S, A, E, H) for m in nets]
    wide = torch.randn(n, 24, device=DEV)
    s1, a1, s2, a2 = wide[:, 0:6], wide[:, 6:8], wide[:, 8:14], wide[:, 14:16]
    ins = [(s2, a2), (s2, a2), (s1, a1), (s1, a1)]
    single = []
    for d, (s, a) in zip(descs, ins):
        out, x0, h1 = torch.empty(n, 1, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
        ops.mlp_forward(d, s, a, out, x0, h1)
        single.append((out, x0, h1))
    multi = [(torch.empty(n, 1, device=DEV), torch.empty(n, E, device=DEV) if k >= 2 else None,
              torch.empty(n, H, device=DEV) if k >= 2 else None) for k in range(4)]
    ops.mlp_forward_multi([(d, s, a) + m for d, (s, a), m in zip(descs, ins, multi)])
    for k in range(4):
        assert torch.equal(single[k][0], multi[k][0])
        if k >= 2:
            assert torch.equal(single[k][1], multi[k][1]) and torch.equal(single[k][2], multi[k][2])
    # TD prologue (SAC form: min of the targets, entropy term) for critic 2 (index 2)
    qn1, qn2, q = single[0][0].view(-1), single[1][0].view(-1), single[2][0].view(-1)
    logp = torch.randn(n, device=DEV)
    reward, done = wide[:, 16:17], (wide[:, 17:18] > 0.5).float()
    loss, dq = torch.zeros(1, device=DEV), torch.empty(n, device=DEV)
    ops.td_huber(q, None, qn1, qn2, logp, 0.2, reward, done, 0.95, loss, dq, None, None)
    d = descs[2]
    grads_a = {}
    for mode in ("dout", "td"):
        for t in d.tensors.values():
            if t is not None:
                t.grad.zero_()
        dh, dx0, da = torch.empty(n, H, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, A, device=DEV)
        if mode == "dout":
            ops.mlp_backward(d, s1, a1, single[2][1], single[2][2], dq.view(n, 1), dh, dx0, da)
        else:
            dq2, parts = torch.empty(n, device=DEV), torch.zeros((n + 15) // 16, device=DEV)
            td = ops.Td(q, qn1, qn2, logp, reward, done, 0.2, 0.95, dq2, parts)
            ops.mlp_backward(d, s1, a1, single[2][1], single[2][2], None, dh, dx0, da, td=td)
            assert torch.equal(dq2, dq)
            np.testing.assert_allclose(float(parts.sum()), float(loss), rtol=1e-5)
        got = {k: t.grad.clone() for k, t in d.tensors.items() if t is not None}
        got["da"] = da.clone()
        if mode == "dout":
            grads_a = got
        else:
            for k in got:
                assert torch.equal(got[k], grads_a[k]), k


@pytest.mark.parametrize("kind,S,A,n", [("actor", 6, 0, 256), ("add", 6, 2, 256), ("add", 5, 2, 77), ("gauss", 5, 0, 256),
                                         ("gauss", 6, 0, 250)])
def test_split_forward_is_bitwise_the_tile_forward(kind, S, A, n):
    """rpo_mlp_forward_split + rpo_mlp_split_head (16 row tiles x 8 column groups of workgroups, head partials added by
    the consumer in the tile kernel's wave order) == rpo_mlp_forward, bit for bit: outputs and both saved
    pre-activations; two networks in one launch as well."""
    from rpo_amd import ops
    E, H = 128, 256
    torch.manual_seed(S * 10 + n)
    se = StateEmbedding(S, E, H)
    if kind == "actor":
        net = SharedPolicy(S, 1, se, E, H, 1, None)
    elif kind == "gauss":
        net = GaussianSharedPolicy(S, 1, se, E, H, 1, None)
    else:
        net = SharedValueAdd(S, A, se, ActionEmbedding(A, E, H), E, H)
    aligned_params(net)
    d = desc_for(ops, net, kind, S, A, E, H)
    assert ops.mlp_split_supported(d)
    wide = torch.randn(n, S + A + 7, device=DEV)
    s = wide[:, 3:3 + S]
    a = wide[:, 3 + S:3 + S + A] if A else None
    n_out = d.n_out
    out, x0, h1 = torch.empty(n, n_out, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
    mode = (1, 2.5, 0.25) if kind == "actor" else (0, 1.0, 0.0)
    ops.mlp_forward(d, s, a, out, x0, h1, *mode)
    part, part2 = torch.zeros(8, n, 2, device=DEV), torch.zeros(8, n, 2, device=DEV)
    out_s, x0_s, h1_s = torch.empty(n, n_out, device=DEV), torch.zeros(n, E, device=DEV), torch.zeros(n, H, device=DEV)
    ops.mlp_forward_split([(d, s, a, part, x0_s, h1_s), (d, s, a, part2, None, None)])
    ops.mlp_split_head(d, part, out_s, *mode)
    assert torch.equal(out, out_s) and torch.equal(x0, x0_s) and torch.equal(h1, h1_s)
    assert torch.equal(part, part2)


def _flat_grads(net):
    """ONE flat gradient buffer behind the network's parameters (like agent/flat.py) -> (flat, {name: (offset, shape)})."""
    total = sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
    flat = torch.zeros(total + 8, device=DEV)
    off = (-flat.data_ptr() // 4) % 4
    for p in net.parameters():
        p.grad = flat[off:off + p.numel()].view(p.shape)
        off += (p.numel() + 3) // 4 * 4
    return flat, total


@pytest.mark.parametrize("kind,S,A,n", [("add", 6, 2, 40007), ("add", 5, 2, 16384), ("actor", 6, 0, 20011), ("gauss", 5, 0, 33000)])
def test_streaming_backward_matches_the_two_pass_kernels(kind, S, A, n):
    """Large batches, 128 -> 256 networks (round 5, csrc/mlp_bwd_stream.h): the backward as two streaming launches -- rows
    kernel with W0 in LDS (dx0, first-layer gradients on MFMA, optionally da), weights kernel (dW0, db0, dW1, db1) -- against
    the rows pass + split-K weights pass of round 3 (rpo_tuning bwd_stream = 0, bwd_onepass = 0): every parameter gradient to
    1e-5 of its tensor's largest entry (other summation orders; the single-head rows kernel multiplies by dout after the
    k-sum), the inf-norm of what was written, bitwise reproducible from call to call, += onto existing gradients; the TD /
    Huber prologue as a launch of its own (same dq and loss shares, bit for bit); the policy step's forms: da alone, and da
    with the state part of a shared embedding's first-layer gradients."""
    from rpo_amd import ops
    torch.manual_seed(n)
    E, H = 128, 256
    se = StateEmbedding(S, E, H)
    if kind == "actor":
        net = SharedPolicy(S, 1, se, E, H, 1, None)
    elif kind == "gauss":
        net = GaussianSharedPolicy(S, 1, se, E, H, 1, None)
    else:
        net = SharedValueAdd(S, A, se, ActionEmbedding(A, E, H), E, H)
    aligned_params(net)
    flat, total = _flat_grads(net)
    d = desc_for(ops, net, kind, S, A, E, H)
    d.splitk = torch.full((max(2, min(256, n // 512)) * total,), 3.0, device=DEV)      # (dirty: the launch zeroes what it uses)
    s = torch.randn(n, S, device=DEV)
    a = torch.randn(n, A, device=DEV) if A else None
    n_out = 2 if kind == "gauss" else 1
    out, x0, h1 = torch.empty(n, n_out, device=DEV), torch.empty(n, E, device=DEV), torch.empty(n, H, device=DEV)
    ops.mlp_forward(d, s, a, out, x0, h1)
    dout = torch.randn(n, n_out, device=DEV) / n
    dh, dx0 = torch.empty(n, H, device=DEV), torch.empty(n, E, device=DEV)
    gm = torch.zeros(ops.CONST["RPO_GRADMAX_LEN"], device=DEV)

    def run(stream, **kw):
        flat.zero_()
        gm.zero_()
        with ops.tuning(bwd_stream=stream, bwd_onepass=0):
            ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, None, gradmax=gm, **kw)
        return flat.clone(), float(gm.max())
    ref, ref_max = run(0)
    got, got_max = run(1)
    again, _ = run(1)
    assert torch.equal(got, again)                                           # fixed orders: bitwise reproducible
    view = {k: (t.grad.data_ptr() - flat.data_ptr()) // 4 for k, t in d.tensors.items() if t is not None}
    for k, t in d.tensors.items():
        if t is None:
            continue
        r = ref[view[k]:view[k] + t.numel()].cpu().numpy()
        g = got[view[k]:view[k] + t.numel()].cpu().numpy()
        assert np.abs(r).max() > 0, k
        np.testing.assert_allclose(g, r, rtol=1e-5, atol=1e-5 * np.abs(r).max() + 1e-12, err_msg=k)
    assert got_max == float(got.abs().max()) and abs(got_max - ref_max) <= 1e-5 * ref_max
    with ops.tuning(bwd_stream=1):                                           # += onto the existing gradients
        ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, None)
    np.testing.assert_allclose(flat.cpu().numpy(), 2 * got.cpu().numpy(), rtol=1e-5, atol=2e-5 * float(got.abs().max()))
    if kind != "add":
        return
    # ---- TD / Huber prologue (critic update)
    q, qn1 = out.view(-1).clone(), torch.randn(n, device=DEV)
    wide = torch.randn(n, 2, device=DEV)
    reward, done = wide[:, 0:1], (wide[:, 1:2] > 0.5).float()
    res = {}
    for stream in (0, 1):
        flat.zero_()
        dq, parts = torch.empty(n, device=DEV), torch.zeros((n + 15) // 16, device=DEV)
        td = ops.Td(q, qn1, None, None, reward, done, 0.0, 0.95, dq, parts)
        with ops.tuning(bwd_stream=stream, bwd_onepass=0):
            ops.mlp_backward(d, s, a, x0, h1, None, dh, dx0, None, td=td)
        res[stream] = (dq.clone(), parts.clone(), flat.clone())
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[1][2].cpu().numpy(), res[0][2].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(res[0][2].abs().max()))
    # ---- the policy step's pass through the critic: dQ/da alone, and with the state part of the first layer (shared embedding)
    for kw in (dict(param_grads=False), dict(param_grads=True, first_layer_state_only=True)):
        das = {}
        for stream in (0, 1):
            flat.zero_()
            da = torch.full((n, A), float("nan"), device=DEV)
            with ops.tuning(bwd_stream=stream, bwd_onepass=0):
                ops.mlp_backward(d, s, a, x0, h1, dout, dh, dx0, da, **kw)
            das[stream] = (da.clone(), flat.clone())
        np.testing.assert_allclose(das[1][0].cpu().numpy(), das[0][0].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(das[0][0].abs().max()))
        np.testing.assert_allclose(das[1][1].cpu().numpy(), das[0][1].cpu().numpy(), rtol=1e-5, atol=1e-5 * float(das[0][1].abs().max()) + 1e-30)
        if kw["param_grads"]:
            ws = d.tensors["Ws"]
            assert float(ws.grad.abs().max()) > 0 and float(d.tensors["W0"].grad.abs().max()) == 0      # the state part only
