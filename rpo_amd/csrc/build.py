"""Build librpo_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python rpo_amd/csrc/build.py [--force] [--asan]

``--asan`` (SURVEY 5 "race detection / sanitizers"): the HOST side of every translation unit -- the C-ABI entry points, their
argument validation, the launch plumbing -- instrumented with AddressSanitizer + UndefinedBehaviorSanitizer into a separate
``librpo_hip_asan.so`` (objects under asan/); the device code is compiled as usual (-fno-gpu-sanitize: GPU ASan / XNACK are not
available on the target pool).  tests/test_sanitizers.py runs the ABI tests against it with the ASan runtime preloaded, on the
CPU box only.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["cartsafe.hip", "pendulum.hip", "evopf.hip", "replay.hip", "train_ops.hip", "mlp.hip", "fused.hip", "nsplit.hip",
           "rollout_stream.hip", "mlp_bwd_stream.hip"]
# per-file flags: the streaming rollout and the streaming backward are compiled without SLP vectorisation (see their headers)
FILE_FLAGS = {"rollout_stream.hip": ["-fno-slp-vectorize"], "mlp_bwd_stream.hip": ["-fno-slp-vectorize"]}
HEADERS = sorted(f for f in os.listdir(HERE) if f.endswith(".h")) + [os.path.join("..", "..", "include", "rpo_hip.h")]   # (every header: a stale object is worse than a rebuild)
TARGET = os.path.join(HERE, "librpo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("HIPCC_EXTRA", "").split()


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


SANITIZE = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-gpu-sanitize", "-g"]
TARGET_ASAN = os.path.join(HERE, "librpo_hip_asan.so")


def asan_runtime():
    """The AddressSanitizer runtime of hipcc's clang (to LD_PRELOAD into a Python process that loads librpo_hip_asan.so)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(HIPCC)), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.check_output([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()


def build(force=False, verbose=True, asan=False):
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    obj_dir = os.path.join(HERE, "asan") if asan else HERE
    os.makedirs(obj_dir, exist_ok=True)
    target = TARGET_ASAN if asan else TARGET
    flags = FLAGS + (SANITIZE if asan else [])
    cmds = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmds.append([HIPCC] + flags + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])
    if cmds:                                                    # translation units are independent: RPO_BUILD_JOBS at a time (4)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max(1, int(os.environ.get("RPO_BUILD_JOBS", "4")))) as pool:
            list(pool.map(run, cmds))
    if force or _stale(target, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", target] + objs + \
            (["-fsanitize=address,undefined", "-shared-libsan"] if asan else [])
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return target


if __name__ == "__main__":
    build(force="--force" in sys.argv, asan="--asan" in sys.argv)
