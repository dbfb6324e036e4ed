#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_coords.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "all_maps or executor_equals or frozen or (unet_vs_oracle and 18A-64-True-executor) or prefetched or tile_ordered or s100k_minkunet" 2>&1 | tail -n 30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 8 $O/pytest.log
ab() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
ab default OSN_X=1
ab maps1 OSN_MAPS_STREAMS=1
ab default2 OSN_X=2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
rm -rf $O/prof
tail -n 3 $O/groups.txt; tail -n 22 $O/streams.txt
