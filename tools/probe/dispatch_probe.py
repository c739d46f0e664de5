#!/usr/bin/env python
"""How does the cost of a launch grow with its workgroups -- per workgroup or per wavefront?  Same total thread count in
workgroups of 64 .. 1024 threads (tools/probe/dispatch_probe.hip), hipGraph of 100 launches, HIP events."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from rpo_amd import _lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libdispatch_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "dispatch_probe.hip"), "-o", so])
_lib._bind_to_torch_hip_runtime()
lib = ctypes.CDLL(so)
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(1 << 16, device="cuda")


def launch(t, g, work):
    rc = lib.probe_launch(t, g, work, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


for work in (0, 200):
    print("work loop = %d" % work)
    for threads in (32768, 131072, 262144):
        row = []
        for t in (64, 128, 256, 512, 1024):
            g = threads // t
            us = bench.time_kernel(lambda: launch(t, g, work))[0]
            row.append("%4d x %-5d %6.2f us" % (t, g, us))
        print("  %7d threads: " % threads + " | ".join(row))
