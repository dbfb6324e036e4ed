#!/bin/bash
# kernel trace of the inference pass (maps + eval-mode forward): per-kernel and per-group launches / ms per pass.  usage: gpu_prof_infer.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
PASSES=20 timeout 300 python $R/tools/infer_loop.py > $O/plain.txt 2>&1
PASSES=20 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/tools/infer_loop.py > $O/prof.txt 2> $O/prof.err
python $R/tools/rocpd_stats.py $O/prof/trace_results.db 23 > $O/stats.csv
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --streams 23 > $O/streams.txt
python $R/tools/group_stats.py $O/stats.csv > $O/groups.txt
python $R/tools/rocpd_stats.py $O/prof/trace_results.db --timeline stem_fwd4 1400 3200 > $O/timeline_last_pass.txt
rm -rf $O/prof
cat $O/plain.txt $O/groups.txt; tail -n 6 $O/streams.txt | cut -c1-150
