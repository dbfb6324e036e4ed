#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_fullsize.py tests/test_gpu_unet.py tests/test_gpu_dense.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and (ws or key5 or key6 or small-key3)) or (adversarial and ws) or executor_equals or fused_head" 2>&1 | tail -2
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from openscene_amd import ops
dev = torch.device('cuda', 0)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n, cin, cout in ((100999, 96, 768), (100999, 768, 96), (100999, 128, 96), (52125, 128, 96)):
    x = torch.randn(n, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.05
    wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
    print('%6d rows %3d -> %3d: dense %.1f us' % (n, cin, cout, timed(lambda: ops.dense_fwd(x, wf, cout))))
PY
SHAPES=k2 REPS=20 timeout 300 python tools/micro_tl.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['shape'], d['n_out'], 'ws %.1f us diff %.1e' % (d.get('ws_us',0), d.get('ws_rel_diff',-1)))"
bash tools/gpu_ab.sh r4r 2 "OSN_X=1|" | grep AB
