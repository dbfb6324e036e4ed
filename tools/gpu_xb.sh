#!/bin/bash
# One-launch batch-norm kernels (csrc/bn.hip, osn_bn_forward_train3): parity tests, stand-alone micro under rocprofv3 (kernel
# durations), and the training step with the kernels off / on / on up to 60 000 rows.   usage: gpu_xb.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
timeout 420 python -m pytest tests/test_gpu_bn_xb.py tests/test_golden_unet.py -m gpu -x -q -p no:cacheprovider > $O/pytest_xb.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -5 $O/pytest_xb.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/mb -o trace -- python $R/tools/micro_bn.py 30 > $O/micro_bn.txt 2> $O/micro_bn.err; echo "micro exit $?"
python $R/tools/rocpd_stats.py $O/mb/trace_results.db > $O/micro_bn_kernels.csv 2>> $O/micro_bn.err
rm -rf $O/mb
cat $O/micro_bn.txt
cd $R; timeout 120 python tools/micro_bn.py 30 graph > $O/micro_bn_graph.txt 2> $O/micro_bn_graph.err; echo "micro graph exit $?"; cat $O/micro_bn_graph.txt; cd /tmp
grep -i "bn_\|col_reduce" $O/micro_bn_kernels.csv | cut -c1-160
cd $R
for v in "0 16384" "1 16384" "1 60000" "0 16384" "1 16384"; do
  set -- $v
  OSN_BN_XB=$1 OSN_BN_XB_MAX_ROWS=$2 timeout 200 python bench.py --steps 30 --warmup 8 --train-only --no-cpu-baseline --no-kernel-events --detail $O/xb_detail.json > $O/xb.tmp 2>$O/xb_$1_$2.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/xb.tmp') if l.startswith('{')][-1]; print('OSN_BN_XB=$1 OSN_BN_XB_MAX_ROWS=$2   %.3f ms/step  loss %.7f' % (d['ms_per_step'], d['loss']))" >> $O/xb_steps.txt
done
cat $O/xb_steps.txt
echo "total $(( $(date +%s) - t0 )) s"
