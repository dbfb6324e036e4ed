#!/bin/bash
# A/B of the product library against a variant build (OSN_LIB_PATH): convolution + weight-gradient micro, training step, and one PMC pass
# (L2 miss traffic of the tile-list kernels).  usage: ab_lib.sh <tag> <variant.so>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
  for lib in "" $2; do
    echo "== lib=${lib:-product} rep $rep" >> $O/ab.txt
    OSN_LIB_PATH=$lib SHAPES=hot REPS=20 timeout 100 python tools/micro_tl.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  conv ', d['shape'], 'tl_us %.1f' % d['tl_us'], 'rel diff vs first-generation kernel %.2e' % d['max_rel_diff'])" >> $O/ab.txt
    OSN_LIB_PATH=$lib MODE=wgrad SHAPES=hot REPS=20 timeout 100 python tools/micro_tl.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  wgrad', d['shape'], 'tl_us %.1f' % d['tl_us'])" >> $O/ab.txt
    OSN_LIB_PATH=$lib timeout 150 python bench.py --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/detail.json 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  step %.3f ms' % d['ms_per_step'])" >> $O/ab.txt
  done
done
cd /tmp
for lib in "" $R/$2; do
  [ "$lib" = "$R/" ] && continue
  name=$(basename ${lib:-product})
  OSN_LIB_PATH=$lib SHAPES=hot REPS=2 timeout 90 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc_$name -o p -- python $R/tools/micro_tl.py > $O/pmc_$name.log 2>&1
  echo "== PMC lib=$name (exit $?)" >> $O/ab.txt
  python $R/tools/pmc_summary.py $O/pmc_$name | grep -A2 "spconv_tl_kernel" >> $O/ab.txt
  rm -rf $O/pmc_$name
done
cat $O/ab.txt
