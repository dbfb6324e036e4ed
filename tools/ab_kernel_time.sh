#!/bin/bash
# usage: tools/ab_kernel_time.sh TAG [ENV=VAL ...]   -> prints the per-step kernel time of 10 + 3 training steps under rocprofv3
# (kernel-time sums are reproducible to ~1 %; wall-clock step times on the shared boxes vary by +-5 % run to run)
tag=$1; shift
out=/root/repo/gpurun_out/ab/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $out/bench.json 2> $out/bench.err
python /root/repo/tools/rocpd_stats.py $out/trace_results.db 13 > $out/stats.csv
echo "$tag $* : $(tail -1 $out/stats.csv)  wall $(python -c "import json;print(round(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['ms_per_step'],2))")"
rm -f $out/trace_results.db
