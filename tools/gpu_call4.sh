#!/bin/bash
# round 3, call 4: side-stream shortcut stages -- parity (repeated, to give a race a chance to show), A/B, per-stream kernel time, full bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x --count 1 -k "executor_equals or frozen or (unet_vs_oracle and executor) or prefetched or tile_ordered or s100k_minkunet or batch8_training" 2>&1 | tail -n 30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 12 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_unet.py -m gpu -q -p no:cacheprovider -x -k "executor_equals or tile_ordered" 2>&1 | tail -n 2
done
ab() {
  tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$tag.json 2>> $O/ab.err
  echo "AB $tag [$*] $(python -c "import json;d=json.loads(open('$O/ab_$tag.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
}
ab noside OSN_SIDE_STREAM=0
ab default OSN_X=1
ab default2 OSN_X=2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
rm -rf $O/prof
cat $O/groups.txt $O/streams.txt
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for k,v in d['phases'].items():
    if isinstance(v, dict) and 'ms' in v: print(k, round(v['ms'],3))
PY
