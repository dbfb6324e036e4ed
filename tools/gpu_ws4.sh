#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3u
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and ws) or (adversarial and ws) or executor_equals or (unet_vs_oracle and 18A-64-True) or executor" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$i.json 2> $O/ab_$i.err; python -c "
import json; d=json.loads(open('$O/ab_$i.json').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'])"
OSN_WS_MAX_ROWS=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab0_$i.json 2> $O/ab0_$i.err; python -c "
import json; d=json.loads(open('$O/ab0_$i.json').read().strip().splitlines()[-1]); print('step (no partial-row ws)', d['ms_per_step'], d['value'])"
done
tail -3 $O/ab_1.err
