#!/usr/bin/env python
"""Micro-benchmark of the one-wave-per-block weight gradient (tools/probes/wgrad_w1.hip) against the product's pair-array kernel
(osn_spconv_wgrad_tl) on the S100k scene.  HIP-event times over back-to-back launches; results against an fp64 reference through the same
pair arrays.   REPS=n  VARIANTS=NW:SCHED:DBG,...  SHAPES=hot|all"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def build_probe():
    src = os.path.join(ROOT, "tools", "probes", "wgrad_w1.hip")
    out = os.path.join(ROOT, "tools", "probes", "bin", "libprobe_w1.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNDEBUG",
                               "-I", os.path.join(ROOT, "openscene_amd", "csrc"), src, "-o", out])
    return out


def plan_items(pl_cpu, K, n_out, bm, regions, strided, slots=512):
    """Work items for the probe: the map's rows cut into `regions` tile ranges of equal pair counts (region x -> item indices
    x, x + regions, ...: block b runs on XCD b % 8), every (region, offset) pair range cut into n items -- strided (item j takes the
    32-pair steps j, j + n, ...: all items of a region march through its rows together) or contiguous.  Returns (items int32
    [slots, 4], first int32 [K + 1], ids int32 [n_items], stats)."""
    i32 = pl_cpu.view(np.int32)
    total = i32[256:256 + K].astype(np.int64)
    cap = K * max(n_out, 1)
    nt = (n_out + bm - 1) // bm
    pref = i32[4096 + 2 * cap:4096 + 2 * cap + K * nt].reshape(K, nt).astype(np.int64)
    prefe = np.concatenate([pref, total[:, None]], axis=1)              # [K][nt + 1]
    cum = prefe.sum(axis=0)                                             # pairs in the tiles before t
    P = int(cum[-1])
    tb = [int(np.searchsorted(cum, P * x / regions, side="left")) for x in range(regions)] + [nt]
    tb[0] = 0
    per = slots // regions
    items = np.full((slots, 4), -1, dtype=np.int32)
    lists = [[] for _ in range(K)]
    worst, mean = 0, 0.0
    for x in range(regions):
        rng = [(int(prefe[k][tb[x]]), int(prefe[k][tb[x + 1]])) for k in range(K)]
        steps = [(b - a + 31) // 32 for a, b in rng]
        T = max(1, sum(steps) // per)
        while sum((st + T - 1) // T for st in steps if st) > per:
            T += 1
        slot = 0
        for k in range(K):
            if not steps[k]:
                continue
            n = (steps[k] + T - 1) // T
            a, b = rng[k]
            for j in range(n):
                idx = x + regions * slot
                if strided:
                    items[idx] = (k, a, b, j | (n << 16))
                    mine = (steps[k] - j + n - 1) // n
                else:
                    c = (steps[k] + n - 1) // n
                    items[idx] = (k, a + 32 * c * j, min(b, a + 32 * c * (j + 1)), 1 << 16)
                    mine = min(c, steps[k] - c * j)
                worst = max(worst, mine)
                lists[k].append(idx)
                slot += 1
        mean += sum(steps) / per / regions
    first = np.zeros(K + 1, dtype=np.int32)
    for k in range(K):
        first[k + 1] = first[k] + len(lists[k])
    ids = np.array([i for l in lists for i in l], dtype=np.int32)
    return items, first, ids, {"max_steps": worst, "mean_steps": round(mean, 1), "items": int(first[K])}


def morton_order(coords4, stride):
    """argsort of the rows by (batch, 3-D Morton code of the voxel coordinates divided by the level's tensor stride)."""
    c = coords4.long()
    xyz = (c[:, 1:] - c[:, 1:].min(0)[0]) // int(stride)
    code = torch.zeros_like(xyz[:, 0])
    for b in range(16):
        for a in range(3):
            code |= ((xyz[:, a] >> b) & 1) << (3 * b + a)
    code |= c[:, 0] << 48
    return torch.argsort(code, stable=True)


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
    reps = int(os.environ.get("REPS", "20"))
    # NW : SCHED : DBG : plan  (plan 0 = the pair lists' own items, 1 = 8 regions strided, 2 = 8 regions contiguous, 3 = 1 region strided)
    variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "2:1:0:0,2:1:0:1,2:1:0:2,2:1:0:3").split(",")]
    dev = torch.device("cuda", 0)
    _lib.load()
    lib = ctypes.CDLL(build_probe())
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.osn_dbg_wgrad_w1.argtypes = [vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, i32, i32, vp, vp, vp, vp]
    lib.osn_dbg_xcc_ids.argtypes = [vp, i32, vp]
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    shapes = [(1, 1, 3, 96, 96), (2, 2, 3, 96, 96), (2, 2, 3, 32, 32), (4, 4, 3, 64, 64)]
    if os.environ.get("SHAPES", "all") == "hot":
        shapes = shapes[:1]
    st = torch.cuda.current_stream(dev).cuda_stream
    xcc = torch.zeros(512, dtype=torch.int32, device=dev)
    lib.osn_dbg_xcc_ids(xcc.data_ptr(), 512, st)
    torch.cuda.synchronize()
    xc = xcc.cpu().numpy()
    print("XCC_ID of block b == b % 8 for", int((xc == np.arange(512) % 8).sum()), "of 512 blocks;  first 16:", xc[:16].tolist(), flush=True)
    for si, so, ks, cin, cout in shapes:
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        x = torch.randn(n_in, cin, device=dev)
        g = torch.randn(n_out, cout, device=dev)
        tiles = cm.kmap_tiles(si, so, ks)[0]
        if os.environ.get("ORDER", "tile") == "morton":
            # round 6: the table's rows in Z-order of the OUTPUT coordinates (the weight gradient sums over all pairs: its order is
            # free) -- a window of consecutive rows is then a spatial block whose 27 offsets gather the same few rows
            perm = morton_order(cm.coords(so), so)
            tl = ops.tile_lists(cm.kmap(si, so, ks)[0][:, perm.long()].contiguous(), out_rows=perm.int().contiguous(),
                                bm=int(os.environ.get("BM", "64")))
        else:
            tl = ops.tile_lists(tiles[1], out_rows=tiles[0]) if tiles is not None else ops.tile_lists(cm.kmap(si, so, ks)[0])
        pl = ops.pair_lists(tl)
        poff, pin, pout = ops.pair_arrays(tl)
        pairs = int(poff[K])
        fl = 2.0 * pairs * cin * cout
        row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out, "pairs": pairs}
        t = timed(lambda: ops.spconv_wgrad_tl(x, g, tl, K), reps)
        prod = ops.spconv_wgrad_tl(x, g, tl, K)
        row.update({"product_us": t, "product_TF": fl / t / 1e6})
        ref = torch.zeros(K, cin, cout, dtype=torch.float64, device=dev)
        po = poff.tolist()
        for k in range(K):
            a, b = po[k], po[k + 1]
            if b > a:
                ref[k] = x[pin[a:b].long()].double().t() @ g[pout[a:b].long()].double()
        scale = ref.abs().max().item()
        row["product_err"] = (prod.double() - ref).abs().max().item() / scale
        ws = torch.empty(512 * cin * cout, dtype=torch.float32, device=dev)
        pl_cpu = pl.cpu().numpy()
        plans = {}
        for NW, SC, DB, PLAN in variants:
            if (cin, cout) != (96, 96) and (NW, SC, DB) != (2, 0, 0):
                continue
            if PLAN not in plans:
                if PLAN == 0:
                    it = pl_cpu.view(np.int32)[512:512 + 2048].reshape(512, 4).copy()
                    it[:, 3] = 1 << 16
                    ks = it[:, 0]
                    lists = [np.nonzero(ks == k)[0] for k in range(K)]
                    first = np.zeros(K + 1, dtype=np.int32)
                    first[1:] = np.cumsum([len(l) for l in lists])
                    plans[PLAN] = (it, first, np.concatenate(lists).astype(np.int32), {"items": int(first[K])})
                elif PLAN == 4:
                    # the product's (balanced) items, re-indexed by the output row of their middle pair: the 1/8 of the items lowest in
                    # the map on XCD 0 (indices 0, 8, 16, ...), the next eighth on XCD 1, ...
                    it0 = pl_cpu.view(np.int32)[512:512 + 2048].reshape(512, 4)
                    valid = [t for t in range(512) if it0[t, 0] >= 0]
                    po_h, pout_h = poff.cpu().numpy(), pout.cpu().numpy()
                    centre = [int(pout_h[po_h[it0[t, 0]] + (it0[t, 1] + it0[t, 2]) // 2]) for t in valid]
                    order = [valid[i] for i in np.argsort(centre, kind="stable")]
                    per_x = (len(order) + 7) // 8
                    it = np.full((512, 4), -1, dtype=np.int32)
                    lists = [[] for _ in range(K)]
                    for r, t in enumerate(order):
                        idx = (r // per_x) + 8 * (r % per_x)
                        it[idx] = (it0[t, 0], it0[t, 1], it0[t, 2], 1 << 16)
                    for idx in range(512):
                        if it[idx, 0] >= 0:
                            lists[it[idx, 0]].append(idx)
                    for k in range(K):
                        lists[k].sort(key=lambda i: it[i, 1])
                    first = np.zeros(K + 1, dtype=np.int32)
                    first[1:] = np.cumsum([len(l) for l in lists])
                    plans[PLAN] = (it, first, np.array([i for l in lists for i in l], dtype=np.int32), {"items": len(order)})
                else:
                    plans[PLAN] = plan_items(pl_cpu, K, n_out, tl.bm, 1 if PLAN == 3 else 8, PLAN != 2)
                row["plan%d" % PLAN] = plans[PLAN][3]
            it_d, first_d, ids_d = (torch.from_numpy(a).to(dev) for a in plans[PLAN][:3])
            gw = torch.zeros(K, cin, cout, device=dev)
            var = NW | (SC << 4) | (DB << 8)

            def run(reduce=1):
                rc = lib.osn_dbg_wgrad_w1(x.data_ptr(), g.data_ptr(), pl.data_ptr(), gw.data_ptr(), n_in, n_out, K, cin, cout, ws.data_ptr(),
                                          var, reduce, it_d.data_ptr(), first_d.data_ptr(), ids_d.data_ptr(), st)
                if rc:
                    raise RuntimeError("osn_dbg_wgrad_w1 failed (%d)" % rc)
            key = "w1_NW%d_S%d_D%d_P%d" % (NW, SC, DB, PLAN)
            row[key + "_us"] = timed(run, reps)
            row[key + "_kernel_only_us"] = timed(lambda: run(0), reps)
            run()
            torch.cuda.synchronize()
            if DB == 0:
                row[key + "_err"] = (gw.double() - ref).abs().max().item() / scale
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
