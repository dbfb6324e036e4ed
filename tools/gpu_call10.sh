#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
export TMPDIR=/tmp
ab() {
  tag=$1; shift
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only "$@" > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
for rep in 1 2 3; do
ab pyr$rep
ab nopf$rep --no-prefetch-pyramid
done
timeout 600 python -m pytest tests/test_gpu_unet.py tests/test_gpu_coords.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "executor_equals or prefetched or all_maps" 2>&1 | tail -n 3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
rm -rf $O/prof
cat $O/groups.txt; tail -n 20 $O/streams.txt | cut -c1-150
