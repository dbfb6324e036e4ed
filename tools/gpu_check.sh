#!/bin/bash
# One GPU call: the full GPU test suite, smoke, and the default bench line (headline + detail file).   usage: gpu_check.sh <tag> [pytest -k expr]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
export TMPDIR=/tmp
t0=$(date +%s)
if [ -n "$2" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "$2" > $O/pytest_gpu.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1
fi
echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -5 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
t0=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $? in $(( $(date +%s) - t0 )) s"
cp bench_detail.json $O/bench_detail.json 2>/dev/null
tail -c 3000 $O/bench.json; echo; wc -c $O/bench.json; tail -3 $O/bench.err
