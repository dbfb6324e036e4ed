#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_unet.py -m gpu -x -q -p no:cacheprovider -k "(forward_and_gradients and 3-32) or executor_equals or known_answer" 2>&1 | tail -2
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
x = torch.randn(n, 3, device=dev); w = torch.randn(125, 3, 32, device=dev)
def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print('stem fwd: %.1f us' % timed(lambda: ops.stem_conv_fwd(x, w, nbr, n)))
PY
bash tools/gpu_ab.sh r4o 2 "OSN_X=1|" | grep AB
