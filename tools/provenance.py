"""Provenance of committed GPU evidence: the ABI version and a hash of the kernel sources the rows were produced
with, so that tests/test_statistical_evidence.py can tell evidence of the CURRENT kernels from stale files.  Regenerate with

    bash tools/collect_statistics.sh r06        (through gpurun; ~25 GPU-minutes)  and copy gpurun_out/statistics_r06/* to profiles/
"""
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    """sha256 over the kernel sources (rpo_amd/csrc/*.hip, *.h and include/rpo_hip.h), comments and blank lines stripped -- a
    comment edit does not make evidence stale, a change of code does."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rpo_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h"))) + [os.path.join(ROOT, "include", "rpo_hip.h")]
    for path in files:
        with open(path) as f:
            text = f.read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        text = "\n".join(line.strip() for line in text.splitlines() if line.strip())
        h.update(os.path.basename(path).encode() + b"\0" + text.encode() + b"\0")
    return h.hexdigest()[:16]


def abi_version():
    with open(os.path.join(ROOT, "include", "rpo_hip.h")) as f:
        return int(re.search(r"#define\s+RPO_ABI_VERSION\s+(\d+)", f.read()).group(1))


def stamp():
    return {"abi": abi_version(), "csrc_sha16": csrc_sha16()}
