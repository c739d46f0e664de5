"""Debugging aid and reproducer (round 6): which launch leaves a non-zero gradient in the PADDING floats of the
flat parameter buffer?  With the shipped library (split-K scratch zeroed by a kernel) nothing does; with the old form,

    bash tools/probe/build_flag_variant.sh memset -DRPO_SPLITK_ZERO=0
    RPO_HIP_LIBRARY=$PWD/rpo_amd/csrc/librpo_hip_memset.so python tools/probe/dbg_padding.py

the hipGraph run shows garbage behind the critic's head bias after its first replayed window (iteration 5), the eager run never.
DBG_INSPECT=1 additionally looks into the slices' scratch between iterations (which makes the symptom disappear).  The post
mortem after the first bad iteration prints what the scratch holds; with slices padded by floats that NO kernel writes,

    git apply tools/probe/splitk_pad.patch && bash tools/probe/build_flag_variant.sh memsetpad -DRPO_SPLITK_ZERO=0 -DRPO_SPLITK_PAD=64
    git checkout rpo_amd/csrc/mlp_bwd.h
    DBG_SPLITK_PAD=64 RPO_HIP_LIBRARY=$PWD/rpo_amd/csrc/librpo_hip_memsetpad.so python tools/probe/dbg_padding.py

it shows what the replayed memset node left there: the 16-byte group {12, 0, 1, 12} (int32) instead of zeros."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("RPO_VERBOSE", "0")
import torch  # noqa: E402
import bench  # noqa: E402
from rpo_amd import ops  # noqa: E402

dev = torch.device("cuda")
n = 4096
for use_graph in (False, True):
    tr = bench.make_trainer(n, dev, 3000, capacity=64, workload="cart_ddpg", torch_seed=123, seed=7000, batch_size=256 * n,
                            use_graph=use_graph)
    tr.vec.reset()
    fl = tr.agent.flat
    pad = torch.ones(fl.total, dtype=torch.bool, device=dev)
    for mod in (tr.agent.actor, tr.agent.critic, tr.agent.nju):
        for p_ in mod.parameters():
            o = fl.offset.get(id(p_))
            if o is not None:
                pad[o:o + p_.numel()] = False
    pad_idx = torch.nonzero(pad).view(-1)
    # wrap the backward entry point: padding of the gradient right after every call
    orig = ops.mlp_backward

    def wrapped(*a, **kw):
        r = orig(*a, **kw)
        if not torch.cuda.is_current_stream_capturing():
            g = fl.grad[pad_idx]
            if bool((g != 0).any()):
                print("   after mlp_backward(n=%d, param_grads=%s, da=%s, state_only=%s): grad padding" % (
                    a[1].shape[0], kw.get("param_grads", True), kw.get("da") is not None, kw.get("first_layer_state_only", False)),
                    [(int(i), float(x)) for i, x in zip(pad_idx.tolist(), g.tolist()) if x != 0], flush=True)
            for name, d in tr.fused.descs.items():
                if getattr(d, "splitk", None) is not None:
                    pass
        return r
    ops.mlp_backward = wrapped
    first = None
    for it in range(48):
        tr.run_steps(1)
        torch.cuda.synchronize()
        if first is None and os.environ.get("DBG_INSPECT", "0") == "1":
            for name, d in tr.fused.descs.items():
                sk = getattr(d, "splitk", None)
                if sk is None:
                    continue
                grads = [t.grad for t in d.tensors.values() if t is not None and t.grad is not None]
                lo = min(g.data_ptr() for g in grads)
                hi = max(g.data_ptr() + 4 * g.numel() for g in grads)
                span = (hi - lo) // 4
                stride = (span + 3) // 4 * 4 + int(os.environ.get('DBG_SPLITK_PAD', '0'))
                Z = sk.numel() // stride
                view = sk[:Z * stride].view(Z, stride)
                base_idx = (lo - fl.grad.data_ptr()) // 4
                cols = [int(i) - base_idx for i in pad_idx.tolist() if base_idx <= int(i) < base_idx + span]
                if cols:
                    sub = view[:, cols]
                    nz = torch.nonzero(sub)
                    if nz.numel():
                        print("   it %d: scratch of %s (Z=%d, span=%d, base %d): non-zero PADDING entries at (slice, flat index):" % (it, name, Z, span, base_idx),
                              [(int(a), cols[int(b)] + base_idx, float(sub[a, b])) for a, b in nz[:8].tolist()], flush=True)
        v = fl.data[pad_idx]
        if bool((v != 0).any()) and first is None:
            first = it
            print("use_graph=%s: padding of the PARAMETERS non-zero after iteration %d:" % (use_graph, it),
                  [(int(i), float(x)) for i, x in zip(pad_idx.tolist(), v.tolist()) if x != 0], flush=True)
            # post mortem (nothing was allocated or inspected before this point): what do the slices' scratch copies, the
            # gradient and the moments hold at the padding positions NOW?
            for name, d in tr.fused.descs.items():
                sk = getattr(d, "splitk", None)
                if sk is None:
                    continue
                grads = [t.grad for t in d.tensors.values() if t is not None and t.grad is not None]
                lo = min(g.data_ptr() for g in grads)
                hi = max(g.data_ptr() + 4 * g.numel() for g in grads)
                span = (hi - lo) // 4
                stride = (span + 3) // 4 * 4 + int(os.environ.get('DBG_SPLITK_PAD', '0'))
                Z = sk.numel() // stride
                view = sk[:Z * stride].view(Z, stride)
                base_idx = (lo - fl.grad.data_ptr()) // 4
                cols = [int(i) - base_idx for i in pad_idx.tolist() if base_idx <= int(i) < base_idx + span]
                sub = view[:, cols] if cols else view[:, :0]
                nz = torch.nonzero(sub)
                print("   post mortem: scratch of %s (Z=%d, span=%d, base %d, ptr %% 4096 = %d): %d non-zero padding entries %s; "
                      "non-finite anywhere: %d" % (name, Z, span, base_idx, sk.data_ptr() % 4096, nz.shape[0],
                                                   [(int(a), cols[int(b)] + base_idx, float(sub[a, b])) for a, b in nz[:6].tolist()],
                                                   int((~torch.isfinite(view)).sum())), flush=True)
                den = (view != 0) & (view.abs() < 1e-37)
                dcols = torch.nonzero(den.any(0)).view(-1)
                print("   post mortem: %s scratch: %d denormal entries in %d columns %s; slice 0 around the head bias: %s; as int32: %s" % (
                    name, int(den.sum()), dcols.numel(), (dcols[:12] + base_idx).tolist(),
                    view[0, max(0, 33660 - base_idx):max(0, 33672 - base_idx)].tolist(),
                    view[0, max(0, 33660 - base_idx):max(0, 33672 - base_idx)].view(torch.int32).tolist()), flush=True)
                if stride > (span + 3) // 4 * 4:
                    tail = view[:, (span + 3) // 4 * 4:].contiguous().view(torch.int32)
                    print("   post mortem: %s scratch, the %d floats behind each slice that NO kernel writes, as int32: slice 0 %s ... slice %d %s;"
                          " distinct 16-byte groups: %s" % (name, tail.shape[1], tail[0, :16].tolist(), Z - 1, tail[-1, :8].tolist(),
                                                           torch.unique(tail.view(-1, 4), dim=0)[:6].tolist()), flush=True)
            print("   post mortem: grad padding", fl.grad[pad_idx].tolist(), flush=True)
            if os.environ.get("RPO_GRAPH_AUDIT", "0") == "1":      # what did the capture RECORD for the memset nodes?
                import ctypes

                class MemsetParams(ctypes.Structure):
                    _fields_ = [("dst", ctypes.c_void_p), ("elementSize", ctypes.c_uint), ("height", ctypes.c_size_t),
                                ("pitch", ctypes.c_size_t), ("value", ctypes.c_uint), ("width", ctypes.c_size_t)]
                hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
                scr = {int(d.splitk.data_ptr()): name for name, d in tr.fused.descs.items() if getattr(d, "splitk", None) is not None}
                for key, e in tr._graphs.entries.items():
                    if e.get("graph") is None:
                        continue
                    graph = ctypes.c_void_p(int(e["graph"].raw_cuda_graph()))
                    cnt = ctypes.c_size_t(0)
                    hip.hipGraphGetNodes(graph, None, ctypes.byref(cnt))
                    nodes = (ctypes.c_void_p * max(1, cnt.value))()
                    hip.hipGraphGetNodes(graph, nodes, ctypes.byref(cnt))
                    seen = {}
                    for i in range(cnt.value):
                        t = ctypes.c_int(-1)
                        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
                        if t.value != 2:
                            continue
                        mp = MemsetParams()
                        rc = hip.hipGraphMemsetNodeGetParams(ctypes.c_void_p(nodes[i]), ctypes.byref(mp))
                        k = (rc, scr.get(int(mp.dst or 0), hex(int(mp.dst or 0))), mp.elementSize, mp.width, mp.height, mp.pitch, mp.value)
                        seen[k] = seen.get(k, 0) + 1
                    # the window's nodes in dependency order, kernels by name
                    class Dim3(ctypes.Structure):
                        _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]

                    class KernelParams(ctypes.Structure):
                        _fields_ = [("blockDim", Dim3), ("extra", ctypes.c_void_p), ("func", ctypes.c_void_p), ("gridDim", Dim3),
                                    ("kernelParams", ctypes.c_void_p), ("sharedMemBytes", ctypes.c_uint)]
                    hip.hipKernelNameRefByPtr.restype = ctypes.c_char_p
                    ne = ctypes.c_size_t(0)
                    hip.hipGraphGetEdges(graph, None, None, ctypes.byref(ne))
                    fr, to = (ctypes.c_void_p * max(1, ne.value))(), (ctypes.c_void_p * max(1, ne.value))()
                    hip.hipGraphGetEdges(graph, fr, to, ctypes.byref(ne))
                    succ = {}
                    indeg = {int(nodes[i]): 0 for i in range(cnt.value)}
                    for i in range(ne.value):
                        succ.setdefault(int(fr[i]), []).append(int(to[i]))
                        indeg[int(to[i])] += 1
                    order, ready = [], [h for h, d_ in indeg.items() if d_ == 0]
                    while ready:
                        h = ready.pop(0)
                        order.append(h)
                        for t2 in succ.get(h, []):
                            indeg[t2] -= 1
                            if indeg[t2] == 0:
                                ready.append(t2)
                    for pos, h in enumerate(order):
                        t = ctypes.c_int(-1)
                        hip.hipGraphNodeGetType(ctypes.c_void_p(h), ctypes.byref(t))
                        if t.value == 0:
                            kp = KernelParams()
                            hip.hipGraphKernelNodeGetParams(ctypes.c_void_p(h), ctypes.byref(kp))
                            nm = hip.hipKernelNameRefByPtr(ctypes.c_void_p(kp.func), None)
                            desc = "%s grid %d block %d lds %d" % ((nm or b"?").decode()[:90], kp.gridDim.x * kp.gridDim.y, kp.blockDim.x, kp.sharedMemBytes)
                        else:
                            desc = "node type %d" % t.value
                        print("      %2d (%d successors) %s" % (pos, len(succ.get(h, [])), desc), flush=True)
                    print("   window %r: node kinds %s; memset nodes (rc, dst, elementSize, width, height, pitch, value) x count: %s" % (
                        key, e.get("node_kinds"), seen), flush=True)
    print("use_graph=%s: first non-zero padding at iteration %s" % (use_graph, first), flush=True)
    ops.mlp_backward = orig
    del tr
    torch.cuda.empty_cache()
