#!/usr/bin/env python
"""Experiment: does a (spatial block, occupancy mask) tile order beat the pure mask order?
(gather locality in L2 vs offset-skip efficiency).  Builds the orders with torch ops."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import ops, synthetic as syn
from openscene_amd.sparse import CoordinateManager
from tools.micro_conv import timed


def group_masks(mask_sorted):
    n = mask_sorted.shape[0]
    pad = (-n) % 32
    g = torch.cat([mask_sorted, mask_sorted.new_zeros(pad)]).reshape(-1, 32)
    out = g[:, 0].clone()
    for j in range(1, 32):
        out |= g[:, j]
    return out.int()


def main():
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = torch.from_numpy(syn.batch_coords([vox])).to(dev)
    cm = CoordinateManager(coords)
    for stride, cin, cout in ((1, 96, 96), (2, 128, 96)):
        n = cm.size(stride)
        nbr = cm.kmap(stride, stride, 3)[0]
        c = cm.coords(stride).long()
        K = 27
        mask = ((nbr >= 0).long() << torch.arange(K, device=dev).reshape(K, 1)).sum(0)
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        g = torch.randn(n, cout, device=dev)
        cnt = ops.kmap_count(nbr)
        ref = ops.spconv_fwd(x, w, nbr, n)
        print("stride %d N=%d %d->%d" % (stride, n, cin, cout))
        for name, blk in (("mask only", None), ("block64+mask", 64), ("block32+mask", 32), ("block16+mask", 16), ("block8+mask", 8), ("morton only", 0)):
            if blk is None:
                key = mask
            else:
                b = blk * stride if blk else 4 * stride
                q = c[:, 1:] // b
                bid = (q[:, 0] * 4096 + q[:, 1]) * 4096 + q[:, 2]
                key = bid * (1 << 27) + (mask if blk else 0)
            order = torch.argsort(key, stable=True).int()
            tbl = nbr[:, order.long()].contiguous()
            gm = group_masks(mask[order.long()])
            act = (tbl >= 0)[:, : n // 128 * 128].reshape(27, -1, 128).any(2).sum(0).float().mean().item()
            out = ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm)
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            t = timed(lambda: ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm), 3)
            print("   %-14s active offsets/tile %.1f   fwd %.1f us   (err %.1e)" % (name, act, t, err))
        # wgrad with rows visited in spatial order? (table columns permuted = same pairs, different order)
        t0 = timed(lambda: ops.spconv_wgrad(x, g, nbr, 27, cnt), 3)
        print("   wgrad hash order %.1f us" % t0)


if __name__ == "__main__":
    main()
