#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3m
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "executor_equals or (unet_vs_oracle and 18A-64-True) or s100k_minkunet or tile_ordered" 2>&1 | tail -n 4
ab() {
  tag=$1; shift
  env $ENVV timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only "$@" > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$ENVV $*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
for rep in 1 2; do
ENVV="OSN_X=1" ab mid$rep
ENVV="OSN_TL_MID_MIN_ROWS=1000000000" ab nomid$rep
ENVV="OSN_TL_MID_MIN_ROWS=30000" ab mid30k$rep
done
