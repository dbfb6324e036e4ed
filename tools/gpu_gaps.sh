#!/bin/bash
# main-stream gap census of the training step under rocprofv3   usage: gpu_gaps.sh <tag> [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
shift
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
echo "bench under rocprof exit $?"
python $R/tools/gap_census.py $O/prof/trace_results.db 3 40 > $O/gap_census.txt 2>&1
rm -rf $O/prof
cat $O/gap_census.txt
