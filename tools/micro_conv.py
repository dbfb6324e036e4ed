#!/usr/bin/env python
"""Micro-benchmark of the dominant convolution launches on the S100k scene (for rocprofv3
--pmc passes): level-0 3^3 conv 96->96 forward (tile-ordered map), its weight gradient, and the
level-1 128->96 pair.  Prints HIP-event times."""
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
    reps = int(os.environ.get("REPS", "3"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    only = os.environ.get("ONLY", "")
    shapes = ((1, 96, 96), (2, 128, 96), (1, 128, 96))
    if only:
        shapes = shapes[:1]
    for stride, cin, cout in shapes:
        n = cm.size(stride)
        nbr = cm.kmap(stride, stride, 3)[0]
        order, tbl, gm = ops.kmap_sort(nbr, cm.kmap_counts(stride, stride, 3))
        pairs = int(ops.kmap_count(nbr).sum())
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        g = torch.randn(n, cout, device=dev)
        if only == "fwd":
            print("fwd(tile-ordered) %.1f us" % timed(lambda: ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm), reps))
            continue
        if only == "fwd_x6":
            wp = ops.weight_prep_x6(w)
            print("fwd_x6(tile-ordered) %.1f us" % timed(
                lambda: ops.spconv_fwd_x6(x, wp, tbl, n, out_rows=order, gmask=gm), reps))
            continue
        if only == "wgrad":
            cnt = ops.kmap_count(nbr)
            print("wgrad(balanced) %.1f us" % timed(lambda: ops.spconv_wgrad(x, g, nbr, 27, cnt), reps))
            continue
        wp = ops.weight_prep_x6(w)
        t_x6 = timed(lambda: ops.spconv_fwd_x6(x, wp, tbl, n, out_rows=order, gmask=gm), reps)
        t_prep = timed(lambda: ops.weight_prep_x6(w), reps)
        ref = ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm)
        got = ops.spconv_fwd_x6(x, wp, tbl, n, out_rows=order, gmask=gm)
        print("   bf16x6 fwd %.1f us (+ weight prep %.1f us), max|d| vs fp32 kernel = %.2e of max" % (
            t_x6, t_prep, (got - ref).abs().max().item() / ref.abs().max().item()))
        t_fwd_sorted = timed(lambda: ops.spconv_fwd(x, w, tbl, n, out_rows=order, gmask=gm), reps)
        t_fwd_plain = timed(lambda: ops.spconv_fwd(x, w, nbr, n), reps)
        cnt = ops.kmap_count(nbr)
        t_wgrad = timed(lambda: ops.spconv_wgrad(x, g, nbr, 27, cnt), reps)
        t_wgrad_u = timed(lambda: ops.spconv_wgrad(x, g, nbr, 27), reps)
        fl = 2.0 * pairs * cin * cout
        print("stride %d  N=%d pairs=%d  %d->%d : fwd(tile-ordered) %.1f us = %.1f TF exact | fwd(hash order) %.1f us | "
              "wgrad(balanced) %.1f us = %.1f TF | wgrad(uniform) %.1f us" % (
                  stride, n, pairs, cin, cout, t_fwd_sorted, fl / t_fwd_sorted / 1e6, t_fwd_plain, t_wgrad,
                  fl / t_wgrad / 1e6, t_wgrad_u))


if __name__ == "__main__":
    main()
