#!/bin/bash
# round 3, call 5: one-call multi-stream maps -- bit-exact tests, A/B, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_coords.py tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "all_maps or one_sync or executor_equals or frozen or (unet_vs_oracle and 18A-64-True) or prefetched or tile_ordered or s100k_minkunet or batch8_s100k" 2>&1 | tail -n 30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 12 $O/pytest.log
ab() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
ab maps1 OSN_MAPS_STREAMS=1
ab maps2 OSN_MAPS_STREAMS=2
ab maps3 OSN_MAPS_STREAMS=3
ab maps4 OSN_MAPS_STREAMS=4
ab maps3b OSN_MAPS_STREAMS=3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
rm -rf $O/prof
cat $O/groups.txt; tail -n 6 $O/streams.txt
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3e/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
r=d['roofline']; print(r['kernel'], r['frac'], r['avg_launch_us'], r['launches_per_step'], r.get('shape_launches_per_step_both_passes'))
for k,v in d['phases'].items():
    if isinstance(v, dict) and 'ms' in v: print(k, round(v['ms'],3))
PY
