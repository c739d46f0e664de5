#!/usr/bin/env python
"""What a second captured branch costs in a hipGraph on this stack (device timestamps, 100 MHz clock).
Topology per iteration, like trainer._overlapped_window:  main: a -> [fork] -> m1 .. m4 -> [join] ;  side: s1 .. s3.
Prints, averaged over the iterations of one replay: kernel-to-kernel gap inside a chain, the fork latency (start of s1 after
the end of a), the start of m1 after a, and the join latency (start of the next a after the later of m4 / s3)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rpo_amd import _lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "libbranch_probe.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                           os.path.join(HERE, "branch_probe.hip"), "-o", so])
_lib._bind_to_torch_hip_runtime()
lib = ctypes.CDLL(so)
lib.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
ITER = 16
PER = 8                                   # a, m1..m4, s1..s3
stamps = torch.zeros(2 * ITER * PER, dtype=torch.int64, device="cuda")


def k(slot, us, wgs=1):
    rc = lib.probe_spin(stamps.data_ptr(), slot, int(us * 100), wgs, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def window(main_us, side_us, branch):
    """branch: 0 serial; 1 side captured first (fork = side.wait_stream(main) right behind a); 2 main's next launch captured
    first, the side waits for an event recorded behind a; 3 like 2, and the side's launches are captured after ALL of main's."""
    main = torch.cuda.current_stream()
    side = window.side
    for i in range(ITER):
        b = i * PER
        k(b, 3)                                              # a (the sampling launch)

        def side_chain():
            with torch.cuda.stream(side):
                for j in range(3):
                    k(b + 5 + j, side_us)
        if branch == 1:
            side.wait_stream(main)
            side_chain()
        elif branch >= 2:
            ev = torch.cuda.Event()
            ev.record(main)
        for j in range(4):
            k(b + 1 + j, main_us)
            if branch == 2 and j == 0:
                side.wait_event(ev)
                side_chain()
        if branch == 3:
            side.wait_event(ev)
            side_chain()
        if branch:
            main.wait_stream(side)
        else:
            for j in range(3):
                k(b + 5 + j, side_us)


window.side = torch.cuda.Stream()


def run(main_us, side_us, branch):
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        window(main_us, side_us, branch)
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        window(main_us, side_us, branch)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    acc = []
    for _ in range(20):
        g.replay()
        torch.cuda.synchronize()
        acc.append(stamps.cpu().numpy().reshape(ITER, PER, 2).astype(np.float64) / 100.0)      # us
    t = np.mean(acc, 0)
    t -= t[0, 0, 0]
    it = (t[-1, 0, 0] - t[1, 0, 0]) / (ITER - 2)
    chain_gap = np.mean([t[i, j + 1, 0] - t[i, j, 1] for i in range(1, ITER) for j in (1, 2, 3)])
    m1 = np.mean(t[1:, 1, 0] - t[1:, 0, 1])
    if branch:
        fork = np.mean(t[1:, 5, 0] - t[1:, 0, 1])
        side_gap = np.mean([t[i, j + 1, 0] - t[i, j, 1] for i in range(1, ITER) for j in (5, 6)])
        last = np.maximum(t[1:-1, 4, 1], t[1:-1, 7, 1])
        join = np.mean(t[2:, 0, 0] - last)
        side_end = np.mean(t[1:-1, 7, 1] - t[1:-1, 0, 1])
        main_end = np.mean(t[1:-1, 4, 1] - t[1:-1, 0, 1])
        print("[%d] main 4 x %4.0f us | side 3 x %4.0f us: iteration %6.1f us; gap in chain %.1f (side %.1f), m1 starts %.1f after a, "
              "s1 (fork) %.1f after a; main ends %.1f, side ends %.1f after a; next a starts %.1f after the later one (join)"
              % (branch, main_us, side_us, it, chain_gap, side_gap, m1, fork, main_end, side_end, join))
    else:
        print("serial: main 4 x %4.0f us + 3 x %4.0f us: iteration %6.1f us; gap in chain %.1f" % (main_us, side_us, it, chain_gap))


for main_us, side_us in ((25, 25), (25, 10), (10, 25), (5, 5)):
    for branch in (0, 1, 2, 3):
        run(main_us, side_us, branch)
