"""Build librpo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python rpo_amd/csrc/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["cartsafe.hip", "pendulum.hip", "evopf.hip", "replay.hip", "train_ops.hip", "mlp.hip", "fused.hip", "nsplit.hip"]
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join("..", "..", "include", "rpo_hip.h")]   # (every header: a stale object is worse than a rebuild)
TARGET = os.path.join(HERE, "librpo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("HIPCC_EXTRA", "").split()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(TARGET, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TARGET] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return TARGET


if __name__ == "__main__":
    build(force="--force" in sys.argv)
