#!/usr/bin/env python
"""Average the per-dispatch counters of rocprofv3 --pmc passes per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if not any(t in name for t in ("spconv", "query", "bn_", "wgrad_tl", "wgrad_w1", "stem_")):
                continue
            short = name.split("(")[0].replace("void ", "").replace("osn::", "")
            grid = row.get("Grid_Size", row.get("Grid_Size_X", ""))
            key = short + " grid=" + str(grid)
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for key in sorted(acc):
        print(key)
        for c in sorted(acc[key]):
            v = acc[key][c]
            print("    %-28s %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))


if __name__ == "__main__":
    main()
