#!/bin/bash
# rocprofv3 PMC passes over tools/micro_conv.py (separate passes, counters only + kernel trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python $R/tools/micro_conv.py > $O/micro.log 2>&1; tail -n 5 $O/micro.log
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  REPS=1 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p$i -- python $R/tools/micro_conv.py > $O/p$i.log 2>&1
  echo "pass $i exit $?"
done
ls -R $O | head -40
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | head -60
