"""AddressSanitizer + UndefinedBehaviorSanitizer build of the library's HOST side ("race detection /
sanitizers"): `rpo_amd/csrc/build.py --asan` instruments the C-ABI entry points, their argument
validation and the launch plumbing of every translation unit into a separate librpo_hip_asan.so (the device code is compiled
as usual: GPU ASan / XNACK are not available on the target pool), and tests/test_abi.py -- exports, struct layouts, the
argument validation of the split stages, the NULL / empty-argument sweep over EVERY entry point -- runs against it in a child
process with the ASan runtime preloaded.  CPU box only; the first run builds the library (~2.5 minutes, cached in-tree like
the normal build)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900, method="thread")
def test_abi_tests_pass_under_asan_and_ubsan():
    from rpo_amd.csrc import build
    if not os.path.exists(build.HIPCC):
        pytest.skip("hipcc not found")
    lib = build.build(verbose=False, asan=True)
    rt = build.asan_runtime()
    assert os.path.exists(rt), rt
    syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    assert "__asan_init" in syms and "__ubsan_handle" in syms          # the host side really is instrumented
    env = dict(os.environ, LD_PRELOAD=rt, RPO_HIP_LIBRARY=lib,
               # (CPython and torch "leak" by design at exit; everything else stays on: overflows, use-after-free, UB traps)
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_abi.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert " passed" in out
    # ... and the child really loaded the instrumented library
    probe = subprocess.run([sys.executable, "-c", "from rpo_amd import _lib; _lib.load(); "
                            "print([l.split()[-1] for l in open('/proc/self/maps') if 'librpo_hip' in l][0])"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert probe.stdout.strip().endswith("librpo_hip_asan.so"), probe.stdout + probe.stderr
