#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 results database (rocpd SQLite): calls, total, average,
minimum and maximum duration in microseconds. `python tools/rocpd_stats.py x_results.db [--csv out.csv]`"""
import csv
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    return [(n, c, t / 1e3, a / 1e3, lo / 1e3, hi / 1e3) for n, c, t, a, lo, hi in rows]


def main():
    rows = stats(sys.argv[1])
    if "--csv" in sys.argv:
        with open(sys.argv[sys.argv.index("--csv") + 1], "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us"])
            for r in rows:
                w.writerow([r[0], r[1]] + [f"{x:.3f}" for x in r[2:]])
    for n, c, t, a, lo, hi in rows:
        print(f"{c:6d} {t:12.1f} {a:10.2f} {lo:10.2f} {hi:10.2f}  {n[:110]}")


if __name__ == "__main__":
    main()
