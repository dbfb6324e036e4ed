#!/usr/bin/env python
"""Experiment: how the tile-ordered forward conv scales with channel width at L0 (fixed overhead vs MFMA work)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import ops, synthetic as syn
from openscene_amd.sparse import CoordinateManager
from tools.micro_conv import timed


def main():
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    n = cm.size(1)
    nbr = cm.kmap(1, 1, 3)[0]
    order, tbl, gm = ops.kmap_sort(nbr)
    cnt = ops.kmap_count(nbr)
    for cin, cout in ((32, 32), (64, 64), (96, 96), (128, 128), (32, 96), (96, 32), (128, 96)):
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        g = torch.randn(n, cout, device=dev)
        t_units = timed(lambda: ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm), 3)
        t_nounit = timed(lambda: ops.spconv_fwd(x, w, tbl, n, out_rows=order), 3)
        t_plain = timed(lambda: ops.spconv_fwd(x, w, nbr, n), 3)
        t_wg = timed(lambda: ops.spconv_wgrad(x, g, nbr, 27, cnt), 3)
        print("%3d->%3d  fwd units %.1f us | tile-ordered no units %.1f | hash order %.1f | wgrad %.1f" % (cin, cout, t_units, t_nounit, t_plain, t_wg))
    # pure streaming reference: copy of a [N, 96] matrix
    a = torch.randn(n, 96, device=dev); b = torch.empty_like(a)
    print("copy [N,96] fp32: %.1f us" % timed(lambda: b.copy_(a), 5))
    idx = torch.randperm(n, device=dev)
    print("row gather [N,96] by random index (torch): %.1f us" % timed(lambda: torch.index_select(a, 0, idx), 5))


if __name__ == "__main__":
    main()
