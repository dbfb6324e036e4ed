#!/bin/bash
# A/B of the product library against a variant build (OSN_LIB_PATH): weight-gradient micro + training step.  usage: ab_lib.sh <tag> <variant.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in "" $2; do
    echo "== lib=${lib:-product} rep $rep" >> $O/ab.txt
    OSN_LIB_PATH=$lib MODE=wgrad SHAPES=hot REPS=20 timeout 100 python tools/micro_tl.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  wgrad', d['shape'], 'tl_us %.1f' % d['tl_us'], 'rel diff vs fp32-MFMA kernel %.2e' % d['max_rel_diff'])" >> $O/ab.txt
    OSN_LIB_PATH=$lib timeout 150 python bench.py --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/detail.json 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  step %.3f ms' % d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
