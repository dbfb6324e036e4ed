#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i
mkdir -p $O
cd $R
export TMPDIR=/tmp
ab() {
  tag=$1; shift
  env $ENVV timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only "$@" > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$ENVV $*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
ENVV="OSN_PREFETCH_PRIORITY=low" ab pyr_lowprio
ENVV="OSN_SIDE_STREAM=0" ab pyr_noside
ENVV="OSN_MAPS_STREAMS=1" ab pyr_maps1
ENVV="OSN_SIDE_STREAM=0 OSN_MAPS_STREAMS=1" ab pyr_noside_maps1
ENVV="OSN_EXECUTOR=0" ab pyr_modules
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
rm -rf $O/prof
head -n 25 $O/stats.csv | cut -c1-50,95-180
