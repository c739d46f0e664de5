#!/usr/bin/env python
"""Per-kernel PMC counter averages from a rocprofv3 rocpd database (rocprofv3 --pmc ... -d DIR -o NAME)."""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    src = "counters_collection" if "counters_collection" in views else "pmc_events"
    cols = [d[1] for d in cur.execute("pragma table_info(%s)" % src)]
    print("# %s columns: %s" % (src, cols))
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    cname = "counter_name" if "counter_name" in cols else ("pmc_name" if "pmc_name" in cols else None)
    val = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    grid = "grid_size" if "grid_size" in cols else ("grid_x" if "grid_x" in cols else None)
    if cname is None or val is None:
        for row in cur.execute("select * from %s limit 5" % src):
            print(row)
        return
    q = "select %s, %s, %s, count(*), avg(%s), min(%s), max(%s) from %s group by 1, 2, 3 order by 1, 3, 2" % (
        name_col, cname, grid or "0", val, val, val, src)
    for r in cur.execute(q):
        print("%-70s %-22s grid=%-9s n=%-4d avg=%-14.1f min=%-14.1f max=%.1f" % (str(r[0])[:70], r[1], r[2], r[3], r[4], r[5], r[6]))


if __name__ == "__main__":
    main(sys.argv[1])
