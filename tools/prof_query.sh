#!/bin/bash
# The query kernels under rocprofv3: kernel-trace stats of tools/micro_query.py and two PMC passes (read side / write side) for
# the HBM traffic per launch.   usage: prof_query.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for lib in "" $R/tools/probes/bin/libosn_qprio1.so $R/tools/probes/bin/libosn_qprio3.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  for rep in 1 2; do
    echo "== lib=${lib:-product} rep $rep" >> $O/micro_query_ab.txt
    OSN_LIB_PATH=$lib SHAPES=wide timeout 120 python $R/tools/micro_query.py >> $O/micro_query_ab.txt 2>&1
  done
done
cat $O/micro_query_ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o trace -- python $R/tools/micro_query.py > $O/micro_query_under_rocprofv3.txt 2>&1
python $R/tools/rocpd_stats.py $O/kt/trace_results.db > $O/query_kernel_stats.csv
rm -rf $O/kt
i=0
for set in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_p$i -o p -- python $R/tools/micro_query.py > $O/pmc_p$i.log 2>&1
  echo "pmc pass $i exit $?"
done
python $R/tools/pmc_summary.py $O > $O/pmc_query_summary.txt 2>&1
rm -rf $O/pmc_p1 $O/pmc_p2
grep -i query $O/query_kernel_stats.csv | cut -c1-200
cat $O/pmc_query_summary.txt | head -60
