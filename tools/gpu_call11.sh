#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_dense.py -m gpu -q --timeout 600 -p no:cacheprovider -k "loader or ensemble or fused_head" 2>&1 | tail -n 30 > $O/pytest.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest.log
tail -n 30 $O/pytest.log
