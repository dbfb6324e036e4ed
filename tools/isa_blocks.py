#!/usr/bin/env python
"""Instruction mix per basic block of one kernel in a hipcc -save-temps .s file (MFMA-bearing blocks only):
isa_blocks.py file.s <kernel-name-substring>"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = None
    for n, l in enumerate(lines):
        if l.startswith("_ZN") and key in l.split(":")[0] and l.split(":")[0].endswith(l.split(":")[0]):
            start = n
            break
    if start is None:
        raise SystemExit("kernel not found")
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        if re.match(r"\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            blocks.append(cur)
        elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
            cur[1].append(l.strip())
    for name, ins in blocks:
        mf = [k for k, i in enumerate(ins) if i.startswith("v_mfma")]
        if not mf and "-a" not in sys.argv:
            continue
        cnt = collections.Counter()
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_mfma"): cnt["mfma"] += 1
            elif op.startswith("v_accvgpr"): cnt["accmov"] += 1
            elif op.startswith("v_"): cnt["valu"] += 1
            elif op.startswith("ds_read") or op.startswith("ds_load"): cnt["ds_read"] += 1
            elif op.startswith("ds_write") or op.startswith("ds_store"): cnt["ds_write"] += 1
            elif op.startswith("ds_"): cnt["ds_other"] += 1
            elif op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): cnt["vmem"] += 1
            elif op == "s_waitcnt": cnt["waitcnt"] += 1
            elif op == "s_nop": cnt["nop"] += 1
            elif op.startswith("s_"): cnt["salu"] += 1
            else: cnt["other"] += 1
        print(name, len(ins), dict(cnt))
        if mf:
            gaps = collections.Counter(mf[k + 1] - mf[k] - 1 for k in range(len(mf) - 1))
            print("   first / last mfma at", mf[0], mf[-1], " instructions between consecutive mfmas:", sorted(gaps.items()))
            print("   waits:", collections.Counter(i for i in ins if i.startswith("s_waitcnt")).most_common(10))


if __name__ == "__main__":
    main()
