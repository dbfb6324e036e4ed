#!/usr/bin/env python
"""Round 6: how much does the weight-gradient stream stretch the main stream's kernels, and does that depend on the weight gradient's
L2 MISS traffic?  A side stream replays the weight-gradient probe (tools/probes/wgrad_w1.hip) back to back -- with the product's pair
order and items (386 MB of misses per launch) or with Z-ordered pair arrays and XCD-pinned strided items (100 MB) -- while the main
stream times a level-0 batch-norm backward, a batch-norm forward and the dominant tile-list convolution.   REPS=n"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402
import micro_w1 as W  # noqa: E402


def main():
    reps = int(os.environ.get("REPS", "40"))
    side_n = int(os.environ.get("SIDE", "120"))
    dev = torch.device("cuda", 0)
    _lib.load()
    lib = ctypes.CDLL(W.build_probe())
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.osn_dbg_wgrad_w1.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp]
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    K, c = 27, 96
    n = cm.size(1)
    x = torch.randn(n, c, device=dev)
    g = torch.randn(n, c, device=dev)
    xs, gs = torch.randn(n, c, device=dev), torch.randn(n, c, device=dev)       # the side stream's own operands
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    tiles = cm.kmap_tiles(1, 1, 3)[0]
    tl_conv = ops.tile_lists(tiles[1], out_rows=tiles[0])
    w = torch.randn(K, c, c, device=dev) * 0.05
    wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
    side = torch.cuda.Stream(device=dev)
    ws = torch.empty(512 * c * c, dtype=torch.float32, device=dev)
    gw = torch.zeros(K, c, c, device=dev)

    def side_setup(order, plan):
        if order == "morton":
            perm = W.morton_order(cm.coords(1), 1)
            tl = ops.tile_lists(cm.kmap(1, 1, 3)[0][:, perm.long()].contiguous(), out_rows=perm.int().contiguous(), bm=64)
        else:
            tl = ops.tile_lists(tiles[1], out_rows=tiles[0])
        pl = ops.pair_lists(tl)
        pl_cpu = pl.cpu().numpy()
        if plan == -2:
            # the PRODUCT kernel with the probe's region plan: 8 row regions pinned to the 8 XCDs, strided items (item.w = j | n << 16),
            # written over the pair-list buffer's own item table (the reduction's per-offset ranges no longer apply: timing only)
            items = W.plan_items(pl_cpu, K, n, tl.bm, 8, True)[0]
            pl[2048:2048 + 512 * 16].view(torch.int32).copy_(torch.from_numpy(items.reshape(-1)).to(dev))
            return tl, pl, "partial", None, None
        if plan < 0:
            return tl, pl, None, None, None
        if plan == 0:
            it = pl_cpu.view(np.int32)[512:512 + 2048].reshape(512, 4).copy()
            it[:, 3] = 1 << 16
            lists = [np.nonzero(it[:, 0] == k)[0] for k in range(K)]
            first = np.zeros(K + 1, dtype=np.int32)
            first[1:] = np.cumsum([len(l) for l in lists])
            p = (it, first, np.concatenate(lists).astype(np.int32))
        else:
            p = W.plan_items(pl_cpu, K, n, tl.bm, 8, True)[:3]
        return (tl, pl) + tuple(torch.from_numpy(a).to(dev) for a in p)

    workloads = {
        "bn_bwd_L0_96": lambda: ops.bn_backward(x, None, g, mean, var, gamma, 1e-5, True, True, False, beta=beta),
        "bn_fwd_L0_96": lambda: ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1),
        "tl_conv_L0_96": lambda: ops.spconv_fwd_tl(x, wf, tl_conv, n, K, c),
    }
    # a deep-level chain: 3^3 128 -> 128 on the 3.3 k-row level (weight-stationary kernel + its reduction) and its one-launch batch norm
    n3 = cm.size(8)
    x3 = torch.randn(n3, 128, device=dev)
    w3 = torch.randn(K, 128, 128, device=dev) * 0.05
    wf3, _ = ops.weight_prep_tl(w3, want_dgrad=False)
    nbr3 = cm.kmap(8, 8, 3)[0]
    tl3 = ops.tile_lists(nbr3)
    ops.pair_lists(tl3)
    g3, b3 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev)
    rm3, rv3 = torch.zeros(128, device=dev), torch.ones(128, device=dev)
    workloads["ws_conv_L3_128"] = lambda: ops.spconv_fwd_ws(x3, wf3, tl3, nbr3, n3, K, 128)
    workloads["bn_small_L3_128"] = lambda: ops.bn_forward_train(x3, g3, b3, 1e-5, None, True, rm3, rv3, 0.1)
    res = {}
    cases = [("alone", None, None), ("beside_PRODUCT_kernel", "tile", -1), ("beside_PRODUCT_kernel_zorder_xcd_strided", "morton", -2)]
    if os.environ.get("PROBE", "0") == "1":
        cases += [("beside_probe_product_order", "tile", 0), ("beside_probe_zorder_xcd", "morton", 1)]
    lib2 = _lib.load()
    job = ctypes.create_string_buffer(64)
    for tag, order, plan in cases:
        cfg = side_setup(order, plan) if order else None
        for name, fn in workloads.items():
            fn()
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if cfg:
                tl, pl, it_d, first_d, ids_d = cfg
                s0.record(side)
                if isinstance(it_d, str):             # the product kernel alone (no reduction) on the strided region items
                    for _ in range(side_n):
                        rc = lib2.osn_spconv_wgrad_tl_partial(xs.data_ptr(), gs.data_ptr(), pl.data_ptr(), 0, gw.data_ptr(), n, n, K, c, c,
                                                              ws.data_ptr(), ws.numel() * 4, ctypes.addressof(job), side.cuda_stream)
                        assert rc == 0
                elif it_d is None:                    # the product's own kernel (158 VGPRs, 2 workgroups per CU) + its reduction
                    with torch.cuda.stream(side):
                        for _ in range(side_n):
                            ops.spconv_wgrad_tl(xs, gs, tl, K)
                for _ in range(side_n if (it_d is not None and not isinstance(it_d, str)) else 0):
                    rc = lib.osn_dbg_wgrad_w1(xs.data_ptr(), gs.data_ptr(), pl.data_ptr(), gw.data_ptr(), n, n, K, c, c, ws.data_ptr(),
                                              2 | (1 << 4), 0, it_d.data_ptr(), first_d.data_ptr(), ids_d.data_ptr(), side.cuda_stream)
                    assert rc == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if cfg:
                s1.record(side)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res["%s/%s_us" % (name, tag)] = round(e0.elapsed_time(e1) / reps * 1e3, 2)
            if cfg:
                res["%s/%s_side_us_per_launch" % (name, tag)] = round(s0.elapsed_time(s1) * 1e3 / side_n, 1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
