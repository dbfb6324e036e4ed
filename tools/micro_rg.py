#!/usr/bin/env python
"""Round 6: the register-gather convolution (csrc/spconv_rg.hip) against the first-generation split-bf16 kernel and the tile-list kernel on
the narrow layers of the S100k scene (32 / 64 channels, 52 k- and 13 k-row levels); results against each other.  REPS=n"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 2)


def main():
    reps = int(os.environ.get("REPS", "30"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    for si, so, ks, cin, cout in ((2, 2, 3, 32, 32), (4, 4, 3, 32, 64), (4, 4, 3, 64, 32), (4, 4, 3, 64, 64), (1, 2, 2, 32, 32), (2, 4, 2, 32, 32),
                                  (1, 1, 3, 32, 32), (1, 1, 3, 64, 64), (2, 2, 3, 64, 64), (2, 2, 3, 32, 64), (2, 2, 3, 64, 32), (1, 1, 3, 32, 64)):
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        x = torch.randn(n_in, cin, device=dev)
        w = torch.randn(K, cin, cout, device=dev) * 0.05
        nbr = cm.kmap(si, so, ks)[0]
        tiles = cm.kmap_tiles(si, so, ks)[0]
        tbl, rows, gm = (tiles[1], tiles[0], tiles[2]) if tiles is not None else (nbr, None, None)
        wp6 = ops.weight_prep_x6(w)
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        pairs = int(ops.kmap_count(nbr).sum())
        row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out, "pairs": pairs, "sorted_table": tiles is not None}
        a = ops.spconv_fwd_x6(x, wp6, tbl, n_out, out_rows=rows, gmask=gm)
        b = ops.spconv_fwd_rg(x, wf, nbr, n_out, cout)
        c = ops.spconv_fwd_rg(x, wf, tbl, n_out, cout, out_rows=rows)
        scale = a.abs().max().item()
        row["rg_vs_x6_maxrel"] = (a - b).abs().max().item() / scale
        row["rg_sorted_vs_x6_maxrel"] = (a - c).abs().max().item() / scale
        row["rg_reproducible"] = bool(torch.equal(b, ops.spconv_fwd_rg(x, wf, nbr, n_out, cout)))
        row["x6_us"] = timed(lambda: ops.spconv_fwd_x6(x, wp6, tbl, n_out, out_rows=rows, gmask=gm), reps)
        row["rg_plain_table_us"] = timed(lambda: ops.spconv_fwd_rg(x, wf, nbr, n_out, cout), reps)
        row["rg_sorted_table_us"] = timed(lambda: ops.spconv_fwd_rg(x, wf, tbl, n_out, cout, out_rows=rows), reps)
        if K > 1 and ops.tl_eligible(K, cin, cout, n_in):
            tl = ops.tile_lists(tbl, out_rows=rows)
            row["tl_us"] = timed(lambda: ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout), reps)
        fl = 2.0 * pairs * cin * cout
        row["rg_TF"] = round(fl / min(row["rg_plain_table_us"], row["rg_sorted_table_us"]) / 1e6, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
