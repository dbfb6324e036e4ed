#!/usr/bin/env python
"""Micro-benchmark: first-generation split-bf16 kernel vs the tile-list kernel on the S100k scene's dominant
convolutions (HIP-event times, back-to-back launches), plus the list-build cost.  REPS=n, SHAPES=all|hot."""
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
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(os.environ.get("REPS", "5"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    res = []
    # (in stride, out stride, ksize, cin, cout)
    shapes = [(1, 1, 3, 96, 96), (1, 1, 3, 128, 96), (1, 1, 3, 96, 128), (2, 2, 3, 96, 96), (2, 2, 3, 128, 96),
              (2, 2, 3, 32, 32), (4, 4, 3, 64, 64), (4, 4, 3, 128, 128), (4, 4, 3, 192, 128),
              (1, 1, 1, 96, 768), (1, 1, 1, 128, 96), (2, 1, 2, 96, 96), (1, 2, 2, 32, 32), (1, 2, 2, 96, 96)]
    if os.environ.get("SHAPES", "all") == "hot":
        shapes = shapes[:2]
    if os.environ.get("SHAPES", "all") == "deep":
        shapes = [(8, 8, 3, 128, 128), (8, 8, 3, 256, 128), (8, 8, 3, 128, 256), (8, 8, 3, 256, 256), (16, 16, 3, 256, 256),
                  (16, 16, 3, 128, 256), (4, 4, 3, 64, 64), (4, 4, 3, 128, 128), (4, 8, 2, 64, 64), (8, 4, 2, 128, 128),
                  (8, 16, 2, 128, 128), (16, 8, 2, 256, 128)]
    if os.environ.get("SHAPES", "all") == "k2":
        shapes = [(2, 1, 2, 96, 96), (1, 2, 2, 32, 32), (1, 2, 2, 96, 96), (4, 2, 2, 128, 96), (2, 4, 2, 32, 32), (8, 4, 2, 128, 128),
                  (4, 8, 2, 64, 64), (16, 8, 2, 256, 128), (8, 16, 2, 128, 128)]
    if os.environ.get("SHAPES", "all") == "l1":
        shapes = [s for s in shapes if s[0] == 2 and s[1] == 2]
    for si, so, ks, cin, cout in shapes:
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        x = torch.randn(n_in, cin, device=dev)
        w = torch.randn(K, cin, cout, device=dev) * 0.05
        row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out}
        if K > 1:
            nbr = cm.kmap(si, so, ks)[0]
            pairs = int(ops.kmap_count(nbr).sum())
            tiles = cm.kmap_tiles(si, so, ks)[0]
            tbl, rows, gm = (tiles[1], tiles[0], tiles[2]) if tiles is not None else (nbr, None, None)
            bm_env = int(os.environ["BM"]) if "BM" in os.environ else None
            t_list = timed(lambda: ops.tile_lists(tbl, out_rows=rows, bm=bm_env), reps)
            tl = ops.tile_lists(tbl, out_rows=rows, bm=bm_env)
            row["bm"] = tl.bm
        else:
            nbr = tbl = rows = gm = tl = None
            pairs = n_out
            t_list = 0.0
        wp6 = ops.weight_prep_x6(w)
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        t_old = timed(lambda: ops.spconv_fwd_x6(x, wp6, tbl, n_out, out_rows=rows, gmask=gm), reps)
        t_new = timed(lambda: ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout), reps)
        a = ops.spconv_fwd_x6(x, wp6, tbl, n_out, out_rows=rows, gmask=gm)
        b = ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout)
        fl = 2.0 * pairs * cin * cout
        row.update({"pairs": pairs, "x6_us": t_old, "tl_us": t_new, "tl_TF": fl / t_new / 1e6, "x6_TF": fl / t_old / 1e6,
                    "lists_us": t_list, "prep_tl_us": timed(lambda: ops.weight_prep_tl(w, want_dgrad=False), reps),
                    "max_rel_diff": (a - b).abs().max().item() / a.abs().max().item()})
        direct = K == 8 and n_out > n_in
        if K > 1 and (direct or (n_out <= 20000 and n_in <= 20000)):
            ops.pair_lists(tl)
            t_ws = timed(lambda: ops.spconv_fwd_ws(x, wf, tl, nbr, n_out, K, cout, direct=direct), reps)
            c = ops.spconv_fwd_ws(x, wf, tl, nbr, n_out, K, cout, direct=direct)
            row.update({"ws_us": t_ws, "ws_TF": fl / t_ws / 1e6, "ws_rel_diff": (a - c).abs().max().item() / a.abs().max().item()})
        res.append(row)
        print(json.dumps(row), flush=True)




def wgrad_main():
    """MODE=wgrad: fp32-MFMA weight gradient vs the pair-array / split-bf16 kernel."""
    reps = int(os.environ.get("REPS", "5"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    shapes = [(1, 1, 3, 96, 96), (1, 1, 3, 128, 96), (2, 2, 3, 96, 96), (2, 2, 3, 128, 96), (2, 2, 3, 32, 32),
              (4, 4, 3, 64, 64), (4, 4, 3, 128, 128), (8, 8, 3, 256, 256), (16, 16, 3, 256, 256), (1, 1, 1, 96, 768),
              (1, 2, 2, 32, 32), (2, 1, 2, 96, 96)]
    if os.environ.get("SHAPES", "all") == "hot":
        shapes = shapes[:2]
    for si, so, ks, cin, cout in shapes:
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        x = torch.randn(n_in, cin, device=dev)
        g = torch.randn(n_out, cout, device=dev)
        row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out}
        if K > 1:
            nbr = cm.kmap(si, so, ks)[0]
            cnt = cm.kmap_counts(si, so, ks)
            swap = so < si
            if swap:
                tiles = cm.kmap_tiles(so, si, ks)[0]
                base = cm.kmap(so, si, ks)[0]
            else:
                tiles = cm.kmap_tiles(si, so, ks)[0]
                base = nbr
            tl = ops.tile_lists(tiles[1], out_rows=tiles[0]) if tiles is not None else ops.tile_lists(base)
            row["pairs_build_us"] = timed(lambda: (setattr(tl, "pairs", None), ops.pair_lists(tl)), reps)
            pairs = int(cnt.sum())
            t_old = timed(lambda: ops.spconv_wgrad(x, g, nbr, K, cnt), reps)
            a = ops.spconv_wgrad(x, g, nbr, K, cnt)
        else:
            tl, swap, pairs = None, False, n_out
            t_old = timed(lambda: ops.spconv_wgrad(x, g, None, 1), reps)
            a = ops.spconv_wgrad(x, g, None, 1)
        t_new = timed(lambda: ops.spconv_wgrad_tl(x, g, tl, K, swap=swap), reps)
        b = ops.spconv_wgrad_tl(x, g, tl, K, swap=swap)
        fl = 2.0 * pairs * cin * cout
        row.update({"pairs": pairs, "old_us": t_old, "tl_us": t_new, "tl_TF": fl / t_new / 1e6, "old_TF": fl / t_old / 1e6,
                    "max_rel_diff": (a - b).abs().max().item() / a.abs().max().item()})
        print(json.dumps(row), flush=True)


if os.environ.get("MODE") == "wgrad":
    main = wgrad_main


def stem_main():
    """MODE=stem: generic kernels vs the dedicated stem kernels (5^3, 3 -> 32, level 0)."""
    reps = int(os.environ.get("REPS", "5"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    n = cm.size(1)
    nbr = cm.kmap(1, 1, 5)[0]
    cnt = cm.kmap_counts(1, 1, 5)
    x = torch.ones(n, 3, device=dev)
    w = torch.randn(125, 3, 32, device=dev) * 0.05
    g = torch.randn(n, 32, device=dev)
    a, b = ops.spconv_fwd(x, w, nbr, n), ops.stem_conv_fwd(x, w, nbr, n)
    print(json.dumps({"fwd_old_us": timed(lambda: ops.spconv_fwd(x, w, nbr, n), reps),
                      "fwd_stem_us": timed(lambda: ops.stem_conv_fwd(x, w, nbr, n), reps),
                      "wgrad_old_us": timed(lambda: ops.spconv_wgrad(x, g, nbr, 125, cnt), reps),
                      "wgrad_stem_us": timed(lambda: ops.stem_conv_wgrad(x, g, nbr, 125), reps),
                      "fwd_diff": (a - b).abs().max().item() / a.abs().max().item(),
                      "kmap125_us": timed(lambda: ops.kmap_build(cm._tables[1], cm._coords[1], 5, 1, with_counts=True), reps),
                      "kmap27_us": timed(lambda: ops.kmap_build(cm._tables[1], cm._coords[1], 3, 1, with_counts=True), reps),
                      "sort27_us": timed(lambda: ops.kmap_sort(cm.kmap(1, 1, 3)[0], cm.kmap_counts(1, 1, 3)), reps)}))


if os.environ.get("MODE") == "stem":
    main = stem_main


if __name__ == "__main__":
    main()
