#!/usr/bin/env python
"""Kernel statistics from a rocprofv3 rocpd SQLite database (the default output format of
rocprofv3 --kernel-trace in ROCm 7.2) -> the same table `--stats` prints, as CSV/markdown.
usage: python tools/rocpd_stats.py <results.db> [out.csv] [--per-step N]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, t, a, mn, mx, 100.0 * t / total))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text[:6000])


if __name__ == "__main__":
    main()
