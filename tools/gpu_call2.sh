#!/bin/bash
# round 3, call 2: the network executor on hardware -- parity tests, bench, per-step kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_fullsize.py tests/test_gpu_dense.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=8 -x -s -k "not query_at and not adversarial and not batchnorm and not voxel" 2>&1 | tail -n 80 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 40 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
tail -c 1500 $O/bench.json; tail -n 5 $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof_bench.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
rm -f $O/prof/trace_results.db
cat $O/groups.txt; tail -c 400 $O/prof_bench.json
cd $R
for v in "" "--prefetch-maps" "--prefetch-maps-threaded"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only $v > $O/ab.json 2>> $O/ab.err
  echo "AB [$v] $(python -c "import json;d=json.loads(open('$O/ab.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
done
OSN_EXECUTOR=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab.json 2>> $O/ab.err
echo "AB [OSN_EXECUTOR=0] $(python -c "import json;d=json.loads(open('$O/ab.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],3))")"
