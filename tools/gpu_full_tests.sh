#!/bin/bash
# the driver's round-end GPU tier: full GPU test suite + smoke
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -25 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $O/smoke.log
