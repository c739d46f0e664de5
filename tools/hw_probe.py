#!/usr/bin/env python
"""Where does the dispatcher put the workgroups of a launch?  (rpo_hw_probe)

    python tools/hw_probe.py                 # the grids of the headline iteration's launches
    python tools/hw_probe.py gx gy gz threads [lds_bytes]
"""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rpo_amd import _lib  # noqa: E402


def probe(gx, gy, gz, threads, lds=4, spin=200):
    lib = _lib.load()
    out = torch.zeros(gx * gy * gz, dtype=torch.int32, device="cuda")
    rc = lib.rpo_hw_probe(gx, gy, gz, threads, lds, spin, ctypes.c_void_p(out.data_ptr()), None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    v = out.cpu().numpy()
    xcc, hw = v >> 16, v & 0xffff
    cu = [(int(x), int(h >> 13) & 7, int(h >> 12) & 1, int(h >> 8) & 15) for x, h in zip(xcc, hw)]
    simd = [c + (int(h >> 4) & 3,) for c, h in zip(cu, hw)]
    per_cu = collections.Counter(cu)
    per_simd = collections.Counter(simd)
    waves = (threads + 63) // 64
    hist = collections.Counter(per_cu.values())
    print("grid (%d, %d, %d) x %d threads, %d B LDS: %d workgroups on %d CUs (%d XCDs); workgroups per CU: %s; first waves per SIMD: max %d" % (
        gx, gy, gz, threads, lds, len(v), len(per_cu), len(set(xcc)), dict(sorted(hist.items())), max(per_simd.values())))
    return per_cu


if __name__ == "__main__":
    if len(sys.argv) > 4:
        a = [int(x) for x in sys.argv[1:]]
        probe(a[0], a[1], a[2], a[3], a[4] if len(a) > 4 else 4)
    else:
        for name, g in (("rollout 4096 lanes", (256, 1, 1, 512, 40000)), ("bwd_b DDPG", (165, 1, 1, 256, 4096)), ("bwd_b SAC", (165, 2, 1, 256, 4096)),
                        ("adam 34564", (136, 1, 1, 256, 4)), ("critic_front DDPG", (8, 16, 4, 256, 30000)),
                        ("evopf 1024 x 64", (1024, 1, 1, 64, 6500)), ("evopf 256 x 256", (256, 1, 1, 256, 26000)),
                        ("evopf batch 64 x 256", (64, 1, 1, 256, 26000)), ("pm workers", (8, 16, 1, 256, 30000))):
            print(name, end=": ")
            probe(*g)
