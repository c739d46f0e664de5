#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes
DIR/NAME_results.db on this image) into the per-kernel table that `--stats` would print: calls, total / average / min /
max duration and share of GPU kernel time.

    python tools/rocpd_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_bench_kernel_stats.txt
"""
import sqlite3
import sys


def main(path, top=60):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = float(sum(r[2] for r in rows))
    print("# source: %s" % path)
    print("# kernels: %d dispatches, %.3f ms total GPU kernel time" % (sum(r[1] for r in rows), total / 1e6))
    print("%-100s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows[:top]:
        print("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name[:100], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
