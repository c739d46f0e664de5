#!/usr/bin/env python
"""Kernel timeline of the steady-state iterations in a rocprofv3 rocpd database: the dispatches between two consecutive
launches of an anchor kernel (default: the env-step / rollout kernel that opens every iteration), for each of the most
common launch sequences, averaged over its last `n` occurrences.

    python tools/rocpd_timeline.py /tmp/prof/x_results.db [anchor-substring] [n]
"""
import collections
import sqlite3
import sys


def main(path, anchor="act_project", n=200):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    short = lambda s: s.replace("(anonymous namespace)::", "").split("(")[0][:70]      # noqa: E731
    names = [short(r[0]) for r in rows]
    cuts = [i for i, s in enumerate(names) if anchor in s]
    iters = [(cuts[j], cuts[j + 1]) for j in range(len(cuts) - 1)]
    seqs = collections.Counter(tuple(names[a:b]) for a, b in iters)
    print("# source: %s; anchor '%s'; %d anchor launches" % (path, anchor, len(cuts)))
    for seq, cnt in seqs.most_common(4):
        if len(seq) > 200:
            continue
        sel = [(a, b) for a, b in iters if tuple(names[a:b]) == seq][-n:]
        k = len(seq)
        dur = [0.0] * k
        gap = [0.0] * k
        for a, b in sel:
            for i in range(k):
                dur[i] += (rows[a + i][2] - rows[a + i][1]) / 1e3
                gap[i] += (rows[a + i + 1][1] - rows[a + i][2]) / 1e3
        m = float(len(sel))
        print("\n# sequence seen %d times (%d launches); averages over the last %d: kernel time %.1f us, span %.1f us"
              % (cnt, k, len(sel), sum(dur) / m, (sum(dur) + sum(gap)) / m))
        print("%-72s %10s %12s" % ("kernel", "avg_us", "gap_after_us"))
        for i in range(k):
            print("%-72s %10.2f %12.2f" % (seq[i], dur[i] / m, gap[i] / m))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "act_project", int(sys.argv[3]) if len(sys.argv) > 3 else 200)
