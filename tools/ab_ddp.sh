#!/bin/bash
# kernel-time / launch statistics of the step under the one-rank DDP wrapping (bench.py --dist-single) next to the plain step
out=/root/repo/gpurun_out/ab/ddp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $out -o trace -- python /root/repo/bench.py --dist-single --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $out/bench.json 2> $out/bench.err
python /root/repo/tools/rocpd_stats.py $out/trace_results.db 13 > $out/stats.csv
tail -1 $out/stats.csv
rm -f $out/trace_results.db
