#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dense.py -m gpu -x -q -p no:cacheprovider -k "flat_adam" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
