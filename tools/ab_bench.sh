#!/bin/bash
# Same-box A/B of bench.py argument sets on the training step: ab_bench.sh <tag> "<bench args>" ["<bench args>" ...]   ("-" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  a="$v"; [ "$v" = "-" ] && a=""
  timeout 300 python bench.py --steps 40 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/ab_detail.json $a > $O/ab.tmp 2>$O/ab.err || tail -5 $O/ab.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/ab.tmp') if l.startswith('{')][-1]; print('%-50s %.3f ms/step  loss %.7f' % ('[$v]', d['ms_per_step'], d['loss']))" >> $O/ab.txt
done
done
cat $O/ab.txt
