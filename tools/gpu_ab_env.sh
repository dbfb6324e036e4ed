#!/bin/bash
# A/B of the training step under environment switches, in one GPU call.  usage: gpu_ab_env.sh <tag> "<ENV=a ...>" "<ENV=b ...>" [...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
i=0
for envs in "$@"; do
  i=$((i+1))
  for rep in 1 2; do
    env $envs timeout 300 python bench.py --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/detail_${i}_$rep.json > $O/run_${i}_$rep.json 2> $O/run_${i}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/run_${i}_$rep.json").read().strip().splitlines()[-1]); print("[$envs] rep $rep: %.3f ms/step  loss %.6f" % (d["ms_per_step"], d["loss"]))
except Exception as e:
    print("[$envs] rep $rep failed:", e); print(open("$O/run_${i}_$rep.err").read()[-1500:])
PY
  done
done
