#!/bin/bash
# rocprofv3 PMC passes over tools/micro_conv.py (counters only + kernel trace, separate passes).
#   PMC_ONLY="fwd_x6 wgrad"   which micro_conv launches to profile (default: both)
#   PMC_SETS=traffic          only the two HBM-traffic passes (default: all four counter sets)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
if [ "${PMC_SETS:-all}" = "traffic" ]; then
  SETS=("FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum")
else
  SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
        "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"
        "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum")
fi
i=0
for only in ${PMC_ONLY:-fwd_x6 wgrad}; do
  for set in "${SETS[@]}"; do
    i=$((i+1))
    ONLY=$only REPS=2 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${only}_p$i -o p -- python $R/tools/micro_conv.py > $O/${only}_p$i.log 2>&1
    echo "$only pass $i exit $?"
  done
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | head -120
