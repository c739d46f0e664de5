#!/usr/bin/env python
"""tools/probe/memset_probe.hip inside a Python process that is bound to the HIP runtime PyTorch ships (the runtime the trainer's
captured windows run on -- rpo_amd/_lib.py binds librpo_hip.so to it), optionally after torch has initialised the device and
with the capture taken by torch.cuda.graph around the library's own launches."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rpo_amd import _lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libmemset_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                           os.path.join(HERE, "memset_probe.hip"), "-o", so])
_lib._bind_to_torch_hip_runtime()
print("torch", torch.__version__, "hip", torch.version.hip)
x = torch.zeros(1, device="cuda")                               # torch owns the device / primary context
lib = ctypes.CDLL(so)
rc = lib.memset_probe_main()
print("rc", rc)
lib.memset_probe_layouts.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.memset_probe_layouts(None, 6))

# ---- the same launches captured by torch.cuda.graph (torch's capture stream, private memory pool, instantiation and launch)
lib.memset_probe_body.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                  ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
Z, span = 256, 34564
for pad, passes, node in ((0, 2, 1), (64, 2, 1), (64, 24, 1), (64, 2, 0)):
    stride = span + pad
    scratch = torch.zeros(Z * stride, device="cuda")
    out = torch.zeros(span, device="cuda")
    other = torch.zeros(3 * 1000 * 1000 + 13, dtype=torch.uint8, device="cuda")

    def body():
        rc = lib.memset_probe_body(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), scratch.data_ptr(), out.data_ptr(),
                                   other.data_ptr(), other.numel(), Z, stride, span, passes, node)
        assert rc == 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    out.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    replays = 20
    for r in range(replays):
        g.replay()
        if r % 4 == 0:
            torch.cuda.synchronize()
            _ = float(out[5])                                    # (host activity between the launches, like the trainer's polling)
    torch.cuda.synchronize()
    idx = torch.arange(span, device="cuda")
    never = (idx % 97) == 3
    bad = int((out[never] != 0).sum())
    tail = scratch.view(Z, stride)[:, span:].contiguous().view(torch.int32) if pad else None
    print("torch.cuda.graph, %s, %d passes, %d unwritten floats per slice: %d never-written positions non-zero%s" % (
        "memset node" if node else "zero kernel", passes, pad, bad,
        "" if tail is None else "; unwritten tail groups: %s" % torch.unique(tail.view(-1, 4), dim=0)[:4].tolist()), flush=True)
