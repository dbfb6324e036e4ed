#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "(forward_and_gradients and tl) or tile_list or adversarial" 2>&1 | tail -n 6
SHAPES=all REPS=10 timeout 300 python tools/micro_tl.py > $O/micro.jsonl 2>$O/micro.err
python - <<'PY'
import json
for l in open('gpurun_out/r3l/micro.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d['n_out'], 'x6 %.1f us  tl %.1f us  tl %.1f TF' % (d['x6_us'], d['tl_us'], d['tl_TF']))
PY
