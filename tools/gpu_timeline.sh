#!/bin/bash
# timeline of the step boundary: dispatches around the last stem forward launch (start of the last profiled step's forward pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o trace -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only "$@" > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --timeline stem_fwd_kernel 900 400 > $O/timeline.txt
rm -rf $O/prof
wc -l $O/timeline.txt
