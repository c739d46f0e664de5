#!/usr/bin/env python
"""Raw kernel rows of a few consecutive steady-state iterations in a rocprofv3 rocpd database (start relative to the first
row, duration, gap to the next start-ordered row, queue / stream when the view has them):
    python tools/rocpd_window.py /tmp/prof/x_results.db [anchor-substring] [anchors] [skip-from-end]"""
import sqlite3
import sys


def main(path, anchor="evopf_step_kernel", count=20, back=40):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    extra = [c for c in ("queue_id", "stream_id", "tid") if c in cols]
    rows = cur.execute("select name, start, end%s from kernels order by start" % "".join(", " + c for c in extra)).fetchall()
    short = lambda s: s.replace("(anonymous namespace)::", "").replace("rpo_mlp_dev::", "").split("(")[0][:48]      # noqa: E731
    cuts = [i for i, r in enumerate(rows) if anchor in r[0]]
    a, b = cuts[-back], cuts[-back + count]
    t0 = rows[a][1]
    print("# %s: rows %d..%d, columns: start_us dur_us end_us %s name" % (path, a, b, " ".join(extra)))
    for i in range(a, b):
        r = rows[i]
        print("%9.2f %7.2f %9.2f %s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3,
                                            " ".join(str(x) for x in r[3:]), short(r[0])))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *[int(x) for x in sys.argv[3:5]])
