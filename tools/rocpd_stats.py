#!/usr/bin/env python
"""Per-kernel statistics (calls, total / average duration) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes NAME_results.db on this ROCm), as CSV.
usage: rocpd_stats.py trace_results.db [steps]   (steps: divide the calls / totals to per-step figures)
       rocpd_stats.py trace_results.db --dispatches PATTERN [N]   (durations in us of the last N dispatches of the kernels
                                                                   whose name contains PATTERN, in launch order)
       rocpd_stats.py trace_results.db --by-position PATTERN PER_STEP [LABELS.json]   (the dispatches of the kernels matching
                                                                   PATTERN, in launch order, folded modulo PER_STEP -- a step issues
                                                                   them in a fixed order, so position p is ONE launch shape; LABELS:
                                                                   a JSON list of PER_STEP strings, tools/tl_launch_sequence.py)
       rocpd_stats.py trace_results.db --timeline PATTERN [BEFORE_US [AFTER_US]]   (every dispatch around the last PATTERN launch)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def dispatches(db, pattern, n):
    rows = db.cursor().execute("select s.kernel_name, d.start, d.end - d.start from rocpd_kernel_dispatch d join "
                               "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    sel = [(short(k), dur / 1e3) for k, _, dur in rows if pattern in k][-n:]
    print("# last %d dispatches of kernels matching %r (us, launch order)" % (len(sel), pattern))
    for k, us in sel:
        print("%-60s %9.2f" % (k[:60], us))


def by_position(db, pattern, per_step, labels):
    rows = db.cursor().execute("select s.kernel_name, d.start, d.end - d.start from rocpd_kernel_dispatch d join "
                               "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    sel = [dur / 1e3 for k, _, dur in rows if pattern in k]
    print("# %d dispatches of kernels matching %r = %d per step x %.2f steps" % (len(sel), pattern, per_step, len(sel) / float(per_step)))
    if len(sel) % per_step:
        print("# WARNING: not a multiple of %d -- the first %d dispatches are dropped" % (per_step, len(sel) % per_step))
        sel = sel[len(sel) % per_step:]
    print("# position  launches  mean_us  min_us  max_us  label")
    for p in range(per_step):
        v = sel[p::per_step]
        print("%3d %6d %9.2f %9.2f %9.2f  %s" % (p, len(v), sum(v) / len(v), min(v), max(v), labels[p] if labels and p < len(labels) else ""))


def timeline(db, pattern, before, after):
    """Every dispatch (stream, start and end in us relative to the anchor's start, name) from `before` us before to `after`
    us after the start of the LAST dispatch whose kernel name contains `pattern`."""
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    key = "stream_id" if "stream_id" in cols else "queue_id"
    rows = cur.execute("select d.%s, d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start" % key).fetchall()
    anchors = [a for _, a, _, n in rows if pattern in n]
    if not anchors:
        print("no dispatch matches", pattern)
        return
    t0 = anchors[-1]
    print("# stream  start_us  end_us  dur_us  kernel   (0 = start of the last %r dispatch)" % pattern)
    for k, a, b, n in rows:
        if t0 - before * 1e3 <= a <= t0 + after * 1e3:
            print("%3s %9.1f %9.1f %7.1f  %s" % (k, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, short(n)[:90]))


def per_stream(db, steps):
    """Kernel time and launches per stream / queue (ms per step), and the union (busy time of the device)."""
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    key = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    print("# columns:", ",".join(cols))
    rows = cur.execute("select %s, start, end from rocpd_kernel_dispatch order by start" % (key or "0")).fetchall()
    tot = {}
    for k, a, b in rows:
        t = tot.setdefault(k, [0, 0.0])
        t[0] += 1
        t[1] += (b - a)
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("%s %s: %.1f launches/step, %.3f ms/step" % (key, k, n / steps, t / steps / 1e6))
    busy, cur_end = 0.0, None
    cur_start = None
    for _, a, b in rows:                      # union of the intervals
        if cur_end is None or a > cur_end:
            if cur_end is not None:
                busy += cur_end - cur_start
            cur_start, cur_end = a, b
        else:
            cur_end = max(cur_end, b)
    if cur_end is not None:
        busy += cur_end - cur_start
    print("device busy (union of all kernels): %.3f ms/step; sum of kernels %.3f ms/step" % (busy / steps / 1e6, sum(t for _, t in tot.values()) / steps / 1e6))
    # which kernels of the other streams run while the busiest stream is idle (exposed side-stream work)
    main_key = max(tot.items(), key=lambda kv: kv[1][1])[0]
    named = cur.execute("select d.%s, d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                        "on d.kernel_id = s.id order by d.start" % key).fetchall() if key else []
    main_iv = [(a, b) for k, a, b, _ in named if k == main_key]
    exposed = {}
    j = 0
    for k, a, b, name in named:
        if k == main_key:
            continue
        # part of [a, b) not covered by any main-stream kernel
        while j < len(main_iv) and main_iv[j][1] <= a:
            j += 1
        t, i, un = a, j, 0
        while t < b:
            if i >= len(main_iv) or main_iv[i][0] >= b:
                un += b - t
                break
            if main_iv[i][0] > t:
                un += main_iv[i][0] - t
            t = max(t, main_iv[i][1])
            i += 1
        if un > 0:
            e = exposed.setdefault(short(name)[:70], [0, 0.0])
            e[0] += 1
            e[1] += un
    print("side-stream time while stream %s is idle: %.3f ms/step" % (main_key, sum(v[1] for v in exposed.values()) / steps / 1e6))
    for name, (n, t) in sorted(exposed.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-70s %6.1f launches/step %8.3f ms/step" % (name, n / steps, t / steps / 1e6))


def main():
    db = sqlite3.connect(sys.argv[1])
    if len(sys.argv) > 3 and sys.argv[2] == "--dispatches":
        return dispatches(db, sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 64)
    if len(sys.argv) > 4 and sys.argv[2] == "--by-position":
        import json
        labels = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
        return by_position(db, sys.argv[3], int(sys.argv[4]), labels)
    if len(sys.argv) > 3 and sys.argv[2] == "--timeline":
        return timeline(db, sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 500.0, float(sys.argv[5]) if len(sys.argv) > 5 else 500.0)
    if len(sys.argv) > 2 and sys.argv[2] == "--streams":
        return per_stream(db, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = db.cursor()
    rows = cur.execute("select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                       "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                       "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    print("kernel,calls_per_step,ms_per_step,avg_us,min_us,max_us,percent")
    for name, n, t, lo, hi in rows:
        print("\"%s\",%.1f,%.4f,%.2f,%.2f,%.2f,%.2f" % (short(name), n / steps, t / steps / 1e6, t / n / 1e3, lo / 1e3, hi / 1e3,
                                                         100.0 * t / total))
    print("\"TOTAL\",%.1f,%.4f,,,," % (sum(r[1] for r in rows) / steps, total / steps / 1e6))


if __name__ == "__main__":
    main()
