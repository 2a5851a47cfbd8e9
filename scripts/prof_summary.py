#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (ROCm 7.2 writes SQLite): per-kernel time stats and, when the
run collected PMC counters, the per-dispatch average of each counter per kernel."""
import sqlite3
import sys


def main(db, title=""):
    c = sqlite3.connect(db)
    cur = c.cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                            "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    print("# %s" % title)
    print("# kernel | calls | total_us | avg_us | min_us | max_us | pct")
    for r in rows[:25]:
        print("%s | %d | %.1f | %.1f | %.1f | %.1f | %.2f" % (r[0][:120], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    try:
        cols = [x[1] for x in cur.execute("pragma table_info(counters_collection)")]
    except sqlite3.Error:
        cols = []
    if cols:
        name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
        if name_col and "counter_name" in cols and "value" in cols:
            q = ("select %s, counter_name, count(*), avg(value), sum(value) from counters_collection group by 1, 2 order by 5 desc" % name_col)
            rows = list(cur.execute(q))
            if rows:
                print("# PMC: kernel | counter | dispatches | avg per dispatch | sum")
                for r in rows[:40]:
                    print("%s | %s | %d | %.1f | %.1f" % (r[0][:100], r[1], r[2], r[3], r[4]))
        else:
            print("# counters_collection columns:", cols)


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
