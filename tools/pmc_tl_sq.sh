#!/bin/bash
# SQ / pipe-utilisation counters of the tile-list kernels (rocprofv3 --pmc passes, counters + kernel trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_tl_sq
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  SHAPES=hot REPS=2 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/fwd_p$i -o p -- python $R/tools/micro_tl.py > $O/fwd_p$i.log 2>&1
  echo "fwd pass $i exit $?"
  MODE=wgrad SHAPES=hot REPS=2 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/wgrad_p$i -o p -- python $R/tools/micro_tl.py > $O/wgrad_p$i.log 2>&1
  echo "wgrad pass $i exit $?"
done
python $R/tools/pmc_summary.py $O > $O/summary.txt 2>&1; grep -A18 "spconv_tl_kernel\|wgrad_tl_kernel<3, 3>" $O/summary.txt | head -80
