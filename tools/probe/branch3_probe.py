#!/usr/bin/env python
"""Three captured branches (tools/probe/branch_probe.hip spin kernels): would the critic's backward + Adam pay off as a THIRD
branch of the EVOPF windows?  Per iteration (us):  main: a(3) -> m(22) -> p(50) -> q(16) -> [two-branch form: b(8) -> w(10) ->
adam(6)] ;  side (rollout, forked behind a, joined before the next a): r(22) -> pr(46) -> st(13) ;  three-branch form: tail forked
behind q: b(8) -> w(10) -> adam(6), joined before the NEXT q, and a 3 us clock bump on main behind q."""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rpo_amd import _lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libbranch_probe.so")
if not os.path.exists(so):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "branch_probe.hip"), "-o", so])
_lib._bind_to_torch_hip_runtime()
lib = ctypes.CDLL(so)
lib.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
ITER, PER = 16, 12
stamps = torch.zeros(2 * ITER * PER, dtype=torch.int64, device="cuda")
side, tail = torch.cuda.Stream(), torch.cuda.Stream()


def k(slot, us):
    assert lib.probe_spin(stamps.data_ptr(), slot, int(us * 100), 1, torch.cuda.current_stream().cuda_stream) == 0


def window(three):
    main = torch.cuda.current_stream()
    tail_pending = False
    for i in range(ITER):
        b = i * PER
        k(b + 0, 3)                                              # a: sample
        forked = torch.cuda.Event()
        forked.record(main)
        k(b + 1, 22)                                             # pi_targ forward
        k(b + 2, 50)                                             # projection
        if three and tail_pending:
            main.wait_stream(tail)
        k(b + 3, 16)                                             # Q_targ || Q forward
        if three:
            qdone = torch.cuda.Event()
            qdone.record(main)
            k(b + 4, 3)                                          # clock bump
            tail.wait_event(qdone)
            with torch.cuda.stream(tail):
                k(b + 5, 8), k(b + 6, 10), k(b + 7, 6)
            tail_pending = True
        else:
            k(b + 5, 8), k(b + 6, 10), k(b + 7, 6)
        side.wait_event(forked)
        with torch.cuda.stream(side):
            k(b + 8, 22), k(b + 9, 46), k(b + 10, 13)
        main.wait_stream(side)
    if three:
        main.wait_stream(tail)


for three in (False, True):
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        window(three)
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        window(three)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    acc = []
    for _ in range(20):
        g.replay()
        torch.cuda.synchronize()
        acc.append(stamps.cpu().numpy().reshape(ITER, PER, 2).astype(np.float64) / 100.0)
    t = np.mean(acc, 0)
    it = (t[-1, 0, 0] - t[2, 0, 0]) / (ITER - 3)
    print("%s branches: %.1f us per iteration (sum of the main chain's kernels: %d us)" % ("three" if three else "two", it, 3 + 22 + 50 + 16 + (3 if three else 24)))
