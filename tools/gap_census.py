#!/usr/bin/env python
"""Where the main stream of a training step is NOT running a kernel (rocprofv3 rocpd database of `bench.py --train-only`).
A step = the dispatches between two consecutive adam_kernel launches; main stream = the one with the most kernel time.
Prints the step's length, the main stream's kernel time, a histogram of the gaps between its consecutive kernels and the
largest gaps with their neighbours and what the other streams ran meanwhile.
usage: gap_census.py trace_results.db [steps_from_the_end=3] [top=25]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN3osn\d+([a-z0-9_]+?)(?:ILi|E|I)", name)
    return (m.group(1) if m else re.sub(r"\(.*\)$", "", name))[:40]


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    rows = cur.execute("select d.%s, d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start" % key).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[3]]
    if len(adam) < back + 1:
        print("not enough steps in the trace")
        return
    for s in range(back):
        i0, i1 = adam[-2 - s], adam[-1 - s]
        step = rows[i0 + 1:i1 + 1]
        t_begin, t_end = rows[i0][2], rows[i1][2]
        tot = {}
        for k, a, b, _ in step:
            tot[k] = tot.get(k, 0) + (b - a)
        main_key = max(tot.items(), key=lambda kv: kv[1])[0]
        mk = [(a, b, n) for k, a, b, n in step if k == main_key]
        others = [(k, a, b, n) for k, a, b, n in step if k != main_key]
        gaps = []
        prev_end, prev_name = t_begin, "adam_kernel(prev step)"
        for a, b, n in mk:
            if a > prev_end:
                gaps.append((a - prev_end, prev_end, a, prev_name, n))
            if b > prev_end:
                prev_end, prev_name = b, n
        print("== step -%d: %.3f ms from Adam to Adam; main stream %s: %d kernels, %.3f ms of kernel time, %.3f ms idle in %d gaps; "
              "other streams: %s" % (s + 1, (t_end - t_begin) / 1e6, main_key, len(mk), tot[main_key] / 1e6,
                                     sum(g[0] for g in gaps) / 1e6, len(gaps),
                                     ", ".join("%s %.3f ms" % (k, v / 1e6) for k, v in tot.items() if k != main_key)))
        edges = [(0, 2), (2, 4), (4, 8), (8, 20), (20, 50), (50, 1e9)]
        for lo, hi in edges:
            sel = [g for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
            print("   gaps %4g - %-6s us: %4d, %.3f ms" % (lo, ("%g" % hi) if hi < 1e8 else "inf", len(sel), sum(g[0] for g in sel) / 1e6))
        if s == 0:
            print("   the %d largest gaps (us | at ms from the step's start | after -> before | other streams meanwhile):" % top)
            for g, a, b, pn, nn in sorted(gaps, reverse=True)[:top]:
                mean = {}
                for k, oa, ob, on in others:
                    ov = min(ob, b) - max(oa, a)
                    if ov > 0:
                        mean[short(on)] = mean.get(short(on), 0) + ov
                desc = ", ".join("%s %.0f" % (k, v / 1e3) for k, v in sorted(mean.items(), key=lambda kv: -kv[1])[:4])
                print("   %7.1f | %6.3f | %s -> %s | %s" % (g / 1e3, (a - t_begin) / 1e6, short(pn), short(nn), desc or "-"))


main()
