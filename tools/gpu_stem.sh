#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_unet.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and 3-32) or executor_equals or known_answer" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from openscene_amd import ops, synthetic as syn
from openscene_amd.sparse import CoordinateManager
dev = torch.device('cuda', 0)
vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
nbr = cm.kmap(1, 1, 5)[0]
n = nbr.shape[1]
x = torch.randn(n, 3, device=dev); g = torch.randn(n, 32, device=dev)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
a = ops.stem_conv_wgrad(x, g, nbr, 125)
b = ops.spconv_wgrad(x, g, nbr, 125, None)
print('stem wgrad: new %.1f us, table kernel %.1f us, rel diff %.2e' % (timed(lambda: ops.stem_conv_wgrad(x, g, nbr, 125)), timed(lambda: ops.spconv_wgrad(x, g, nbr, 125, None)), ((a - b).abs().max() / b.abs().max()).item()))
PY
for i in 1 2; do
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --train-only > $O/ab_$i.json 2> $O/ab_$i.err; python -c "
import json; d=json.loads(open('$O/ab_$i.json').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['value'])"
done
