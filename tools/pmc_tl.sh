#!/bin/bash
# HBM traffic of the second-generation kernels: rocprofv3 PMC passes (counters + kernel trace only, separate passes for the
# read and the write side) over tools/micro_tl.py on the S100k scene's hot shapes.  Output: gpurun_out/pmc_tl/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_tl
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for mode in fwd wgrad; do
  for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
    i=$((i+1))
    if [ $mode = fwd ]; then envs="SHAPES=hot REPS=2"; else envs="MODE=wgrad SHAPES=hot REPS=2"; fi
    env $envs timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${mode}_p$i -o p -- python $R/tools/micro_tl.py > $O/${mode}_p$i.log 2>&1
    echo "$mode pass $i exit $?"
  done
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1; head -150 $O/summary.txt
