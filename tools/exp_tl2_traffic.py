#!/usr/bin/env python
"""What bounds the tile-list kernels?  The LDS-DMA kernel on the S100k level-0 96 -> 96 conv with (a) everything, (b) every
B-fragment load redirected to ONE 1 KB block (L1-resident: no fragment traffic from L2), (c) no DMA gathers, (d) neither.
Results are garbage in (b)-(d); only the time matters."""
import ctypes, os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dev = torch.device("cuda", 0)
_lib.load()
set_tl2 = ctypes.CDLL(_lib.LIB_PATH).osn_dbg_set_tl2
vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
for si, cin, cout in ((1, 96, 96), (2, 96, 96), (1, 96, 128)):
    n = cm.size(si)
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    tiles = cm.kmap_tiles(si, si, 3)[0]
    tl = ops.tile_lists(tiles[1], out_rows=tiles[0])
    wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
    row = {"shape": "s%d 3^3 %d->%d" % (si, cin, cout)}
    for name, v in (("tl_round2", 0), ("tl2", 1), ("tl2_no_fragment_traffic", 3), ("tl2_no_gathers", 5), ("tl2_neither", 7)):
        set_tl2(v)
        row[name + "_us"] = round(timed(lambda: ops.spconv_fwd_tl(x, wf, tl, n, 27, cout)), 1)
    set_tl2(1)
    print(json.dumps(row), flush=True)
