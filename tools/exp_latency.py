#!/usr/bin/env python
"""Experiment: is the conv bound by the latency of the random row gathers?  Same tables, but every
valid entry redirected to a 1024-row window (all gathers hit L1/L2)."""
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
    for cin, cout in ((96, 96), (128, 128)):
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        g = torch.randn(n, cout, device=dev)
        for name, win in (("real", None), ("window 1024 rows", 1024), ("window 64 rows", 64)):
            t_, n_ = tbl, nbr
            if win:
                t_ = torch.where(tbl >= 0, tbl % win, tbl).contiguous()
                n_ = torch.where(nbr >= 0, nbr % win, nbr).contiguous()
            tf = timed(lambda: ops.spconv_fwd(x, w, t_, n, out_rows=order, gmask=gm), 3)
            tw = timed(lambda: ops.spconv_wgrad(x, g, n_, 27, cnt), 3)
            print("%d->%d  %-18s fwd %.1f us   wgrad %.1f us" % (cin, cout, name, tf, tw))


if __name__ == "__main__":
    main()
