#!/bin/bash
# One GPU-box session: parity tests -> smoke -> bench -> rocprofv3 kernel trace.
# Everything lands in gpurun_out/ (merged back by gpurun).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu" | tee $O/pytest_gpu.log
timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider ${PYTEST_ARGS:-} 2>&1 | tail -n 400 >> $O/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest_gpu.log
tail -n 60 $O/pytest_gpu.log
if [ "${RUN_DIAG:-0}" = "1" ]; then
  echo "== diag"
  timeout 600 python tools/diag_train_grad.py > $O/diag.log 2>&1; echo "diag exit $?" >> $O/diag.log; tail -n 120 $O/diag.log
fi
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log; tail -n 5 $O/smoke.log
echo "== bench"
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -c 3000 $O/bench.json; tail -n 20 $O/bench.err
if [ "${SKIP_PROF:-0}" != "1" ]; then
  echo "== rocprofv3"
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r01 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/prof_bench.json 2> $O/prof.err
  echo "rocprof exit $?"
  ls -R $O/prof | head -20
  f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -n 40 "$f"
fi
