#!/bin/bash
# Round 6, experiment 1: does Z-order of the pair arrays turn the weight gradient's 27-fold row re-use into L2 hits?
# (kill criterion of VERDICT r5 item 2: TCC hit rate of the region-pinned strided plan must exceed 50 %.)   usage: r6_e1.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for order in tile morton; do
  ORDER=$order VARIANTS=2:1:0:0,2:1:0:1,2:1:0:3 SHAPES=hot REPS=20 timeout 300 python $R/tools/micro_w1.py > $O/time_$order.log 2>&1
  echo "time $order exit $?"; tail -2 $O/time_$order.log | cut -c1-1500
done
for bm in 32 128; do
  ORDER=morton BM=$bm VARIANTS=2:1:0:1 SHAPES=hot REPS=20 timeout 300 python $R/tools/micro_w1.py > $O/time_morton_bm$bm.log 2>&1
  echo "time morton bm=$bm exit $?"; tail -1 $O/time_morton_bm$bm.log | cut -c1-900
done
for order in tile morton; do
  for plan in 0 1; do
    ORDER=$order VARIANTS=2:1:0:$plan SHAPES=hot REPS=3 timeout 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/p_${order}_$plan -o p -- python $R/tools/micro_w1.py > $O/pmc_${order}_$plan.log 2>&1
    echo "pmc $order plan $plan exit $?"
    python $R/tools/pmc_summary.py $O/p_${order}_$plan > $O/pmc_${order}_plan$plan.txt 2>&1
    grep -A3 "wgrad_w1\|wgrad_tl_kernel" $O/pmc_${order}_plan$plan.txt
    rm -rf $O/p_${order}_$plan
  done
done
