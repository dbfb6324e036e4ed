#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_unet.py tests/test_gpu_dense.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and key5) or (forward_and_gradients and key6) or (forward_and_gradients and small-key3) or executor_equals or fused_head or known_answer" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
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
for n, cin, cout in ((100999, 96, 768), (100999, 768, 96), (100999, 128, 96), (100999, 96, 128), (52125, 128, 96), (13393, 192, 128), (3326, 256, 128), (730, 128, 256), (100999, 96, 20)):
    x = torch.randn(n, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.05
    wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
    w6 = ops.weight_prep_x6(w)
    a = ops.dense_fwd(x, wf, cout); b = ops.spconv_fwd_x6(x, w6, None, n)
    ref = x.double() @ w.double()
    print('%6d rows %3d -> %3d: dense %.1f us (%.0f TF), x6 %.1f us; rel err dense %.1e x6 %.1e' % (n, cin, cout, timed(lambda: ops.dense_fwd(x, wf, cout)),
          2.0 * n * cin * cout / timed(lambda: ops.dense_fwd(x, wf, cout)) / 1e6, timed(lambda: ops.spconv_fwd_x6(x, w6, None, n)),
          ((a.double() - ref).abs().max() / ref.abs().max()).item(), ((b.double() - ref).abs().max() / ref.abs().max()).item()))
PY
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$i.json 2> $O/ab_$i.err; python -c "
import json
for l in open('$O/ab_$i.json'):
    if l.startswith('{'): d=json.loads(l); print('step', d['ms_per_step'], d['value'])"
done
