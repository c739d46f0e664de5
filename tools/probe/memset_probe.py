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
