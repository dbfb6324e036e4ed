#!/bin/bash
# L2 hit / miss and fetch size of the weight-gradient probe under each work-item plan (one process per plan).  usage: pmc_w1.sh <tag> "plans"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for plan in ${2:-0 1 3}; do
  for set in "FETCH_SIZE TCC_HIT_sum"; do   # (three TCC counters in one pass exceed the hardware: rocprofv3 aborts and hangs)
    VARIANTS=2:1:0:$plan SHAPES=hot REPS=3 timeout 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$plan -o p -- python $R/tools/micro_w1.py > $O/p$plan.log 2>&1
    echo "plan $plan exit $?"
    python $R/tools/pmc_summary.py $O/p$plan > $O/pmc_plan$plan.txt 2>&1
    grep -A4 "wgrad_w1\|wgrad_tl_kernel" $O/pmc_plan$plan.txt
    rm -rf $O/p$plan
  done
done
