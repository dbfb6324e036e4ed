#!/bin/bash
# A/B of environment knobs on the training step: gpu_knobs.sh <tag> "<env assignments>" ["<env assignments>" ...]   ("-" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
for rep in 1 2; do
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e=""
  env $e timeout 200 python bench.py --steps 40 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/knob_detail.json > $O/knob.tmp 2>$O/knob.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/knob.tmp') if l.startswith('{')][-1]; print('%-70s %.3f ms/step  loss %.7f' % ('[$v]', d['ms_per_step'], d['loss']))" >> $O/knobs.txt
done
done
cat $O/knobs.txt
