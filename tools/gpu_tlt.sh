#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and tl) or tile_list or (adversarial and tl) or executor_equals" 2>&1 | tail -2
SHAPES=hot REPS=20 timeout 200 python tools/micro_tl.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d.get('bm'), 'tl %.1f us  %.1f TF diff %.1e' % (d['tl_us'], d['tl_TF'], d['max_rel_diff']))"
SHAPES=l1 REPS=20 timeout 200 python tools/micro_tl.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d.get('bm'), 'tl %.1f us  %.1f TF diff %.1e' % (d['tl_us'], d['tl_TF'], d['max_rel_diff']))"
bash tools/gpu_ab.sh r4q 2 "OSN_X=1|" | grep AB
