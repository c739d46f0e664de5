"""DESIGN.md states the CURRENT design: at most 600 lines, and every entry point, kernel, switch and file it
names exists in the tree -- so that a stale name (round 5's DESIGN carried switches that had been deleted two rounds earlier)
fails the CPU suite instead of misleading a reader.  The round-by-round narrative lives in docs/HISTORY.md."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree_text():
    out = []
    for base, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", "profiles", "docs", ".pytest_cache", "asan")]
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".sh")) or f in ("INTEGRATION.md", "README.md"):
                with open(os.path.join(base, f), errors="replace") as fh:
                    out.append(fh.read())
    return "\n".join(out)


def test_design_is_short_and_names_only_what_exists():
    with open(os.path.join(ROOT, "DESIGN.md")) as f:
        text = f.read()
    assert len(text.splitlines()) <= 600
    assert os.path.exists(os.path.join(ROOT, "docs", "HISTORY.md"))
    tree = _tree_text()
    code = set(re.findall(r"`([^`\n]+)`", text))
    missing = []
    for tok in sorted(code):
        # C-ABI entry points / device functions / kernels / switches / constants: rpo_..., RPO_..., *_kernel, *.hip / *.h / *.py files
        for name in re.findall(r"\b(?:rpo_[a-z_0-9]+|RPO_[A-Z_0-9]+|[a-z_0-9]+_kernel)\b", tok):
            base = name.rstrip("_")
            if name.endswith("_") or "*" in tok[max(0, tok.find(name) - 1):tok.find(name) + len(name) + 2]:
                ok = base in tree                               # a family (`rpo_split_*`, `rpo_*_step`): the stem exists
            else:
                ok = re.search(r"\b%s\b" % re.escape(name), tree) is not None
            if not ok and name not in ("rpo_amd",):
                missing.append(name)
        for path in re.findall(r"\b((?:rpo_amd|tests|tools|oracle|docs|include|profiles)/[\w./-]+\.(?:py|h|hip|sh|md|json|npz))\b", tok):
            if "*" not in path and "N" not in os.path.basename(path) and not os.path.exists(os.path.join(ROOT, path)):
                missing.append(path)
        for fn in re.findall(r"\b([a-z_]+\.(?:hip|h))\b", tok):
            if "/" not in tok and not os.path.exists(os.path.join(ROOT, "rpo_amd", "csrc", fn)) and not os.path.exists(os.path.join(ROOT, "include", fn)):
                missing.append(fn)
    assert not missing, sorted(set(missing))
    # tests it cites by name exist
    for t in set(re.findall(r"`(test_[a-z_0-9]+)`", text)):
        assert re.search(r"def %s\b" % t, tree), t


def test_the_library_issues_nothing_a_capture_turns_into_a_memset_or_memcpy_node():
    """a hipMemsetAsync captured into the trainer's windows filled the split-K scratch with a stale pattern at
    replay.  Every launch of the library is a kernel; the one remaining hipMemsetAsync is the reproducer's form behind
    RPO_SPLITK_ZERO=0."""
    d = os.path.join(ROOT, "rpo_amd", "csrc")
    calls = []
    for fn in sorted(os.listdir(d)):
        if not fn.endswith((".hip", ".h")):
            continue
        with open(os.path.join(d, fn)) as f:
            for no, line in enumerate(f, 1):
                code = line.split("//")[0]
                for m in re.findall(r"\bhipMem(?:set|cpy)\w*\s*\(", code):
                    calls.append((fn, no, m))
    assert [c[0] for c in calls] == ["mlp_bwd.h"] and calls[0][2].startswith("hipMemsetAsync"), calls
    with open(os.path.join(d, "mlp_bwd.h")) as f:
        text = f.read()
    assert "#define RPO_SPLITK_ZERO 1" in text
