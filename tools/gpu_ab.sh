#!/bin/bash
# A/B of bench.py's training step: gpu_ab.sh <tag> <rounds> -- "<env> | <bench args>" ...   (each variant = 'ENV=.. ENV=.. | --flag ..')
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; R_N=$2; shift 2
mkdir -p $O
cd $R
export TMPDIR=/tmp
for rep in $(seq 1 $R_N); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    envs="${v%%|*}"; flags="${v#*|}"
    env $envs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only $flags > $O/ab_${i}_$rep.json 2>> $O/ab.err
    echo "AB[$envs|$flags] $(python -c "
import json
for l in open('$O/ab_${i}_$rep.json'):
    if l.startswith('{'): print(round(json.loads(l)['ms_per_step'], 3))")"
  done
done
