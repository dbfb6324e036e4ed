#!/usr/bin/env python
"""Round 6: batch-norm statistics from the tile-list convolution's epilogue -- parity and time against the separate passes.
forward : [conv] [col_reduce, finalize, apply]           vs  [conv + per-tile sums] [finalize, apply]
backward: [dgrad conv] [col_reduce, finalize, apply]     vs  [dgrad conv + mask + per-tile sums] [finalize, apply]
HIP-event times over back-to-back launches on the S100k scene.   REPS=n"""
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


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    reps = int(os.environ.get("REPS", "30"))
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    K = 27
    for stride, cin, cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (4, 128, 128)):
        n = cm.size(stride)
        tiles = cm.kmap_tiles(stride, stride, 3)[0]
        tl = ops.tile_lists(tiles[1], out_rows=tiles[0]) if tiles is not None else ops.tile_lists(cm.kmap(stride, stride, 3)[0])
        row = {"shape": "s%d k3 %d->%d" % (stride, cin, cout), "rows": n, "tiles": tl.n_tiles,
               "stats_ok": ops.tl_stats_ok(n, n, K, cin, cout, tl.bm)}
        if not row["stats_ok"]:
            print(json.dumps(row), flush=True)
            continue
        torch.manual_seed(0)
        w = torch.randn(K, cin, cout, device=dev) * 0.05
        wf, wb = ops.weight_prep_tl(w, flip=True)
        feats = torch.randn(n, cin, device=dev)
        gamma, beta = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.3
        rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
        # ---------------- forward
        def fwd_plain():
            x = ops.spconv_fwd_tl(feats, wf, tl, n, K, cout)
            return (x,) + ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1)
        part = torch.empty(tl.n_tiles, 2, cout, dtype=torch.float64, device=dev)

        def fwd_fused():
            x = ops.spconv_fwd_tl(feats, wf, tl, n, K, cout, bn_partial=part)
            return (x,) + ops.bn_forward_train_partials(x, part, gamma, beta, 1e-5, None, True, rm, rv, 0.1)
        xa, ya, ma, va = fwd_plain()
        xb, yb, mb, vb = fwd_fused()
        row.update({"fwd_x_bitwise": bool(torch.equal(xa, xb)), "fwd_y_maxdiff": (ya - yb).abs().max().item(), "fwd_mean_rel": rel(mb, ma),
                    "fwd_var_rel": rel(vb, va), "fwd_y_bitwise": bool(torch.equal(ya, yb))})
        row["fwd_conv_us"] = timed(lambda: ops.spconv_fwd_tl(feats, wf, tl, n, K, cout), reps)
        row["fwd_conv_stats_us"] = timed(lambda: ops.spconv_fwd_tl(feats, wf, tl, n, K, cout, bn_partial=part), reps)
        row["fwd_plain_us"] = timed(fwd_plain, reps)
        row["fwd_fused_us"] = timed(fwd_fused, reps)
        # ---------------- backward: the convolution below (cin2 = cout -> cout2) sends its input gradient to THIS batch norm
        x, y, mean, var = xa, ya, ma, va
        gy = torch.randn(n, cout, device=dev)                      # gradient at the NEXT conv's output (cout -> cout, same map)
        w2 = torch.randn(K, cout, cout, device=dev) * 0.05
        _, wb2 = ops.weight_prep_tl(w2, flip=True)
        gres = torch.randn(n, cout, device=dev)
        for tag, relu, use_y, extra in (("t1_mask_from_x", True, False, ()), ("t2_two_sources_mask_y", True, True, (gres,)), ("no_relu", False, False, ())):
            def bwd_plain():
                gin = ops.spconv_fwd_tl(gy, wb2, tl, n, K, cout)
                return ops.bn_backward_multi(x, y if (relu and use_y) else None, [gin] + list(extra), mean, var, gamma, 1e-5, relu, True,
                                             False, beta=beta if (relu and not use_y) else None)

            def bwd_fused():
                gm, pt = ops.spconv_fwd_tl_bnbwd(gy, wb2, tl, n, K, cout, x, mean, var, 1e-5, relu, y=y if use_y else None, gamma=gamma,
                                                 beta=beta, extra=extra)
                return ops.bn_backward_partials(x, gm, pt, mean, var, gamma, 1e-5, True) + (gm,)
            gxa, _, gga, gba = bwd_plain()
            gxb, ggb, gbb, gm = bwd_fused()
            row.update({tag + "_gx_rel": rel(gxb, gxa), tag + "_gx_bitwise": bool(torch.equal(gxa, gxb)), tag + "_ggamma_rel": rel(ggb, gga),
                        tag + "_gbeta_rel": rel(gbb, gba)})
            row[tag + "_plain_us"] = timed(bwd_plain, reps)
            row[tag + "_fused_us"] = timed(bwd_fused, reps)
        row["dgrad_conv_us"] = timed(lambda: ops.spconv_fwd_tl(gy, wb2, tl, n, K, cout), reps)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
