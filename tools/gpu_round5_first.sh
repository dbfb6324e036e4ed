#!/bin/bash
# First GPU call of round 5: what round 4 prepared but could not run (its GPU budget was spent).   usage: gpu_round5_first.sh <tag>
#   1. parity of the buffer-resource gathers (tile-list conv + pair-array weight gradient, OSN_TL_BUFGATHER=1) on the kernels' own tests
#   2. the training step with each prepared knob, two rounds (tools/gpu_knobs.sh: ~3 s per run)
#   3. a gap census of the new default tree
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
t0=$(date +%s)
OSN_TL_BUFGATHER=1 timeout 400 python -m pytest tests/test_gpu_spconv.py -m gpu -x -q -p no:cacheprovider -k "tl or ws" > $O/pytest_bufgather.log 2>&1
echo "buffer-gather parity exit $? in $(( $(date +%s) - t0 )) s"; tail -3 $O/pytest_bufgather.log
bash tools/gpu_knobs.sh $1 "-" "OSN_TL_BUFGATHER=1" "OSN_PREP_OVERLAP=1" "OSN_SIDE_PRIORITY=low" "OSN_PL_ITEMS=384" "OSN_PL_ITEMS=256" "OSN_TL_BUFGATHER=1 OSN_PREP_OVERLAP=1"
bash tools/gpu_gaps.sh $1
echo "total $(( $(date +%s) - t0 )) s"
