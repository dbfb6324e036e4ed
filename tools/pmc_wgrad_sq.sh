#!/bin/bash
# Round 6: what do the weight gradient's waves wait for?  SQ / TA / TCP counters of the product kernel (and the one-wave probe) on the
# level-0 3^3 96 -> 96 map, product pair order and Z-order.   usage: pmc_wgrad_sq.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TC_STALL_sum" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  for order in tile morton; do
    ORDER=$order VARIANTS=2:1:0:1 SHAPES=hot REPS=3 timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${order}_p$i -o p -- python $R/tools/micro_w1.py > $O/${order}_p$i.log 2>&1
    echo "$order pass $i exit $?"
  done
done
for order in tile morton; do
  mkdir -p $O/all_$order; cp -r $O/${order}_p* $O/all_$order/ 2>/dev/null
  python $R/tools/pmc_summary.py $O/all_$order > $O/summary_$order.txt 2>&1
  rm -rf $O/all_$order
done
rm -rf $O/tile_p*/ $O/morton_p*/
grep -A48 "wgrad_tl_kernel<3, 3, true>" $O/summary_tile.txt | head -60
