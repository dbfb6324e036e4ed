#!/usr/bin/env python
"""Micro-benchmark of round 5's structural experiment (VERDICT r4 "next" #1): the tile-list convolution with the gathered
operand read from PRE-SPLIT bf16 planes straight into MFMA fragments (csrc/spconv_pl.hip) against the product's tile-list kernel,
on the S100k scene's dominant convolutions.  HIP-event times over back-to-back launches; outputs compared bit for bit.
REPS=n  SHAPES=hot|all  BMS=64,96,128  OCC=2|3"""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def build_probe():
    """tools/probes/spconv_pl.hip -> tools/probes/bin/libprobe_pl.so (hipcc cross-compiles here; the .so travels to the GPU box)."""
    import subprocess
    src = os.path.join(ROOT, "tools", "probes", "spconv_pl.hip")
    out = os.path.join(ROOT, "tools", "probes", "bin", "libprobe_pl.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNDEBUG",
                               "-I", os.path.join(ROOT, "openscene_amd", "csrc"), src, "-o", out])
    return out


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


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d)" % (what, rc))


def main():
    reps = int(os.environ.get("REPS", "20"))
    occ = int(os.environ.get("OCC", "2"))
    bms = [int(b) for b in os.environ.get("BMS", "64,96,128").split(",")]
    dev = torch.device("cuda", 0)
    _lib.load()
    lib = ctypes.CDLL(build_probe())
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.osn_dbg_split_planes.argtypes = [vp, i64, i32, vp, vp]
    lib.osn_dbg_spconv_fwd_pl.argtypes = [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, vp]
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    shapes = [(1, 1, 3, 96, 96), (1, 1, 3, 128, 96), (2, 2, 3, 96, 96), (2, 2, 3, 128, 96), (2, 2, 3, 32, 32), (1, 1, 3, 96, 128)]
    if os.environ.get("SHAPES", "hot") == "hot":
        shapes = shapes[:2]
    st = torch.cuda.current_stream(dev).cuda_stream
    counters = torch.zeros(256, dtype=torch.int32, device=dev)
    for si, so, ks, cin, cout in shapes:
        K = ks ** 3
        n_in, n_out = cm.size(si), cm.size(so)
        x = torch.randn(n_in, cin, device=dev)
        w = torch.randn(K, cin, cout, device=dev) * 0.05
        tiles = cm.kmap_tiles(si, so, ks)[0]
        tbl, rows = tiles[1], tiles[0]
        pairs = int(ops.kmap_count(cm.kmap(si, so, ks)[0]).sum())
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        planes = torch.empty(n_in * cin * 6, dtype=torch.uint8, device=dev)

        def split():
            check(lib.osn_dbg_split_planes(x.data_ptr(), n_in, cin, planes.data_ptr(), st), "split_planes")
        t_split = timed(split, reps)
        fl = 2.0 * pairs * cin * cout
        for bm in bms:
            tl = ops.tile_lists(tbl, out_rows=rows, bm=bm)
            row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out, "pairs": pairs, "bm": bm, "occ": occ,
                   "split_planes_us": t_split}
            ref = None
            if bm <= 88:
                t_tl = timed(lambda: ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout), reps)
                ref = ops.spconv_fwd_tl(x, wf, tl, n_out, K, cout)
                row.update({"tl_us": t_tl, "tl_TF": fl / t_tl / 1e6})
            out = torch.empty(n_out, cout, device=dev)

            def run():
                check(lib.osn_dbg_spconv_fwd_pl(planes.data_ptr(), n_in, wf.data_ptr(), tl.buf.data_ptr(),
                                                      rows.data_ptr() if rows is not None else None, out.data_ptr(), None, n_out, K, cin, cout,
                                                      bm, counters.data_ptr(), occ, st), "spconv_fwd_pl")
            t_pl = timed(run, reps)
            out.zero_()
            run()
            torch.cuda.synchronize()
            if ref is None:
                tl64 = ops.tile_lists(tbl, out_rows=rows, bm=64)
                ref = ops.spconv_fwd_tl(x, wf, tl64, n_out, K, cout)
            if bm == 64 and cin == 96 and cout == 96 and os.environ.get("DBG", "1") == "1":
                for d, nm in ((1, "no_A_loads"), (2, "no_B_reloads"), (3, "no_loads")):
                    def run_d():
                        check(lib.osn_dbg_spconv_fwd_pl(planes.data_ptr(), n_in, wf.data_ptr(), tl.buf.data_ptr(), rows.data_ptr(), out.data_ptr(), None,
                                                              n_out, K, cin, cout, bm, counters.data_ptr(), occ | (d << 8), st), "spconv_fwd_pl dbg")
                    row["pl_%s_us" % nm] = timed(run_d, reps)
            row.update({"pl_us": t_pl, "pl_TF": fl / t_pl / 1e6, "bitwise_equal": bool(torch.equal(out, ref)),
                        "max_rel_diff": float((out - ref).abs().max() / ref.abs().max()), "counters_zero": int(counters.abs().sum()) == 0})
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
