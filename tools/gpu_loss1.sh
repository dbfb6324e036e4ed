#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3w
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dense.py -m gpu -x -q -p no:cacheprovider -k "distill" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$i.json 2> $O/ab_$i.err; python -c "
import json; d=json.loads(open('$O/ab_$i.json').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'], d['loss'])"
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only --torch-loss > $O/abt_$i.json 2> $O/abt_$i.err; python -c "
import json; d=json.loads(open('$O/abt_$i.json').read().strip().splitlines()[-1]); print('step (torch loss)', d['ms_per_step'], d['value'], d['loss'])"
done
tail -3 $O/ab_1.err
