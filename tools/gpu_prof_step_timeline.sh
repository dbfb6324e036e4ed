#!/bin/bash
# kernel trace of the training step with the dispatches around the step's seams (loss, Adam -> next forward) as timelines.  usage: gpu_prof_step_timeline.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --train-only > $O/prof.json 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 13 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 13 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --timeline adam_kernel 400 700 > $O/timeline_adam.txt
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --timeline loss_rows_kernel 300 500 > $O/timeline_loss.txt
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --timeline adam_kernel 9000 100 > $O/timeline_step.txt
rm -rf $O/prof
cat $O/groups.txt; tail -n 6 $O/streams.txt | cut -c1-150
