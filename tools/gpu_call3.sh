#!/bin/bash
# round 3, call 3: new executor features -- targeted parity tests, A/B of the knobs (wall + rocprof kernel time)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_unet.py tests/test_gpu_spconv.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "batchnorm or executor_equals or frozen or (unet_vs_oracle and 18A-64-True-executor) or wgrad_tile_list or relu_add_cat" 2>&1 | tail -n 30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 12 $O/pytest.log
ab() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
ab noside OSN_WGRAD_SIDE_STREAM=0
ab default OSN_X=1
ab tlsmall OSN_TL_SMALL_MAX_ROWS=4096
ab tlsmall16k OSN_TL_SMALL_MAX_ROWS=16384
ab noside_tlsmall OSN_WGRAD_SIDE_STREAM=0 OSN_TL_SMALL_MAX_ROWS=4096
prof() {
  tag=$1; shift
  cd /tmp
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof_$tag.json 2> $O/prof_$tag.err
  python $R/tools/rocpd_stats.py $O/prof_$tag/trace_results.db 13 > $O/stats_$tag.csv
  python $R/tools/group_stats.py $O/stats_$tag.csv > $O/groups_$tag.txt
  rm -rf $O/prof_$tag
  echo "== $tag"; cat $O/groups_$tag.txt
  cd $R
}
prof noside OSN_WGRAD_SIDE_STREAM=0
prof noside_tlsmall OSN_WGRAD_SIDE_STREAM=0 OSN_TL_SMALL_MAX_ROWS=4096
