#!/bin/bash
# round 3, call 1: new parity tests (with durations) + a bench baseline with the batch8 phase
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_dense.py tests/test_gpu_unet.py -m gpu -q --timeout 900 -p no:cacheprovider --durations=25 -x -s 2>&1 | tail -n 120 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 70 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench exit $?"
tail -c 2500 $O/bench.json; tail -n 5 $O/bench.err
