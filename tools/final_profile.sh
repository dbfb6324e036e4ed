#!/bin/bash
# The bench command itself under rocprofv3: python bench.py (defaults) -> per-kernel stats over the whole run and the
# individual dispatch durations of the dominant kernel, to set beside the bench line's HIP-event average.
out=/root/repo/gpurun_out/final
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out -o trace -- python /root/repo/bench.py > $out/bench.json 2> $out/bench.err
python /root/repo/tools/rocpd_stats.py $out/trace_results.db > $out/kernel_stats_whole_run.csv
python /root/repo/tools/rocpd_stats.py $out/trace_results.db --dispatches spconv_tl_kernel 400 > $out/tl_dispatches.txt
rm -f $out/trace_results.db
head -5 $out/kernel_stats_whole_run.csv; grep -c . $out/tl_dispatches.txt
