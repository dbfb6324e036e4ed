#!/usr/bin/env python
"""Group a tools/rocpd_stats.py CSV into the step's phases (launches and ms per step)."""
import csv
import sys


def group(n):
    if 'spconv_tl' in n or 'tl_reduce_parts' in n: return 'conv fwd/dgrad TL'
    if 'dense_kernel' in n: return 'conv 1x1 (dense kernel)'
    if 'spconv_ws' in n: return 'conv fwd/dgrad WS (small maps, 2^3 fine side)'
    if 'spconv_fwd_x6' in n or 'spconv_fwd_kernel' in n or 'spconv_fwd_pipe' in n or 'stem_fwd' in n: return 'conv fwd/dgrad x6+stem'
    if 'reduce_partial_rows' in n or 'fixup_units' in n: return 'conv partial reduce/fixup'
    if 'wgrad_tl_kernel' in n: return 'wgrad TL'
    if 'wgrad_tl_reduce' in n or 'reduce_items' in n: return 'wgrad reduce'
    if 'spconv_wgrad' in n or 'wgrad_plan' in n: return 'wgrad old (stem, 1x1)'
    if 'weight_prep' in n: return 'weight prep'
    if 'bn_' in n or 'col_reduce' in n: return 'BN'
    if 'pair_' in n or 'tile_lists' in n: return 'maps: tile/pair lists'
    if 'kmap' in n or 'rocprim' in n or 'hash_insert' in n or 'unique' in n or 'scan_' in n or 'table_renumber' in n: return 'maps: coords + tables + sort'
    if 'fillBuffer' in n or 'copyBuffer' in n: return 'memset/memcpy'
    if n.startswith('_ZN2at') or 'at::' in n: return 'torch'
    return 'other:' + n[:40]


def main():
    rows = list(csv.reader(open(sys.argv[1])))[1:-1]
    groups = {}
    for r in rows:
        a = groups.setdefault(group(r[0]), [0.0, 0.0])
        a[0] += float(r[1])
        a[1] += float(r[2])
    tot = [0.0, 0.0]
    for k, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        print("%-32s %7.1f launches  %7.3f ms" % (k, c, t))
        tot[0] += c
        tot[1] += t
    print("%-32s %7.1f launches  %7.3f ms" % ("TOTAL", tot[0], tot[1]))


if __name__ == "__main__":
    main()
