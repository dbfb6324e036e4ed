#!/usr/bin/env python
"""Round-4 micro-benchmark: the LDS-DMA tile-list kernel (spconv_tl2_kernel) against the round-2 kernel on the S100k
scene's convolutions -- same lists, same weight image, HIP-event times of back-to-back launches, and BITWISE comparison
of the outputs (same conversions, same product order).  REPS=n."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def timed(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(os.environ.get("REPS", "20"))
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    set_tl2 = ctypes.CDLL(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else os.path.join(ROOT, "openscene_amd", "lib", "libopenscene_amd.so")).osn_dbg_set_tl2
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    shapes = [(1, 1, 3, 96, 96), (1, 1, 3, 128, 96), (1, 1, 3, 96, 128), (2, 2, 3, 96, 96), (2, 2, 3, 128, 96), (2, 2, 3, 96, 128),
              (4, 4, 3, 128, 128), (4, 4, 3, 192, 128), (4, 4, 3, 128, 192), (2, 2, 3, 32, 32), (4, 4, 3, 64, 64), (2, 2, 3, 64, 32),
              (8, 8, 3, 256, 256), (1, 1, 1, 96, 768), (1, 1, 1, 128, 96), (2, 1, 2, 96, 96), (1, 2, 2, 32, 32)]
    if os.environ.get("SHAPES") == "hot":
        shapes = shapes[:4]
    for si, so, ks, cin, cout in shapes:
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        g = torch.Generator().manual_seed(cin * 1000 + cout)
        x = torch.randn(n_in, cin, generator=g).to(dev)
        w = (torch.randn(K, cin, cout, generator=g) * 0.05).to(dev)
        row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out}
        if K > 1:
            nbr = cm.kmap(si, so, ks)[0]
            pairs = int(ops.kmap_count(nbr).sum())
            tiles = cm.kmap_tiles(si, so, ks)[0]
            tbl, rows = (tiles[1], tiles[0]) if tiles is not None else (nbr, None)
            tl = ops.tile_lists(tbl, out_rows=rows)
            row["bm"] = tl.bm
        else:
            tl, pairs = None, n_out
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        res = {}
        for name, on in (("tl", 0), ("tl2", 1), ("tl_again", 0), ("tl2_again", 1)):
            set_tl2(on)
            res[name] = timed(lambda: ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout), reps)
            if name == "tl":
                a = ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout).clone()
            elif name == "tl2":
                b = ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout).clone()
                b2 = ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout).clone()
        set_tl2(1)
        fl = 2.0 * pairs * cin * cout
        row.update({"pairs": pairs, "tl_us": round(min(res["tl"], res["tl_again"]), 1), "tl2_us": round(min(res["tl2"], res["tl2_again"]), 1),
                    "tl2_TF": round(fl / min(res["tl2"], res["tl2_again"]) / 1e6, 1),
                    "bitwise_equal": bool(torch.equal(a, b)), "reproducible": bool(torch.equal(b, b2)),
                    "max_rel_diff": ((a - b).abs().max() / a.abs().max()).item(), "finite": bool(torch.isfinite(b).all())})
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
