#!/usr/bin/env python
"""Micro-benchmark of round 5's second structural experiment: the tile-list convolution with a producer wave feeding the MFMA waves
through a ring of LDS slots and flags instead of two barriers per step (tools/probes/spconv_tr.hip) against the product kernel.
HIP-event times over back-to-back launches; outputs compared bit for bit.   REPS=n  VARIANTS=R:GD:WGS,...  BMS=64,..."""
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402


def build_probe():
    src = os.path.join(ROOT, "tools", "probes", "spconv_tr.hip")
    out = os.path.join(ROOT, "tools", "probes", "bin", "libprobe_tr.so")
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


def main():
    reps = int(os.environ.get("REPS", "20"))
    variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("VARIANTS", "2:1:2:1").split(",")]
    bms = [int(b) for b in os.environ.get("BMS", "64").split(",")]
    dev = torch.device("cuda", 0)
    _lib.load()
    lib = ctypes.CDLL(build_probe())
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.osn_dbg_spconv_fwd_tr.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, i32, vp]
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    shapes = [(1, 1, 3, 96, 96), (2, 2, 3, 96, 96), (1, 1, 3, 128, 96)]
    if os.environ.get("SHAPES", "all") == "hot":
        shapes = shapes[:1]
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
        fl = 2.0 * pairs * cin * cout
        for bm in bms:
            bm_eff = bm if n_out > 60000 else min(bm, 32) if bm == 64 else bm
            tl = ops.tile_lists(tbl, out_rows=rows, bm=bm_eff)
            row = {"shape": "s%d->s%d k%d %d->%d" % (si, so, ks, cin, cout), "n_out": n_out, "pairs": pairs, "bm": bm_eff}
            tl_ref = tl if bm_eff <= 88 else ops.tile_lists(tbl, out_rows=rows, bm=64)
            t_tl = timed(lambda: ops.spconv_fwd_tl(x, wf, tl_ref, n_out, K, cout), reps)
            ref = ops.spconv_fwd_tl(x, wf, tl_ref, n_out, K, cout)
            row.update({"tl_us": t_tl, "tl_TF": fl / t_tl / 1e6})
            for R, GD, wgs, npw in variants:
                if cin == 128 and (R, GD, npw) != (2, 3, 2):
                    continue
                out = torch.zeros(n_out, cout, device=dev)

                def run():
                    rc = lib.osn_dbg_spconv_fwd_tr(x.data_ptr(), n_in, wf.data_ptr(), tl.buf.data_ptr(), rows.data_ptr(), out.data_ptr(), n_out, K,
                                                   cin, cout, bm_eff, counters.data_ptr(), R | (GD << 4) | (npw << 8), wgs, st)
                    if rc:
                        raise RuntimeError("spconv_fwd_tr failed (%d)" % rc)
                t = timed(run, reps)
                torch.cuda.synchronize()
                key = "tr_R%d_GD%d_W%d_P%d" % (R, GD, wgs, npw)
                row[key + "_us"] = t
                row[key + "_bitwise"] = bool(torch.equal(out, ref))
                if (R, GD, npw) == (2, 1, 1) and cin == 96 and os.environ.get("ABL", "1") == "1":
                    for dbg, nm in ((1, "no_gathers"), (2, "no_B_reloads"), (3, "no_loads"), (7, "no_loads_no_tile_rmw"), (11, "no_loads_no_mfma"), (15, "no_loads_no_rmw_no_mfma"), (31, "skeleton_no_frag_reads"), (47, "skeleton_no_staging"), (63, "skeleton_control_only"), (127, "control_no_epilogue_stores"), (191, "control_static_tiles"), (255, "control_static_no_epilogue"), (639, "control_no_steps_no_epilogue")):
                        def run_d():
                            lib.osn_dbg_spconv_fwd_tr(x.data_ptr(), n_in, wf.data_ptr(), tl.buf.data_ptr(), rows.data_ptr(), out.data_ptr(), n_out, K,
                                                      cin, cout, bm_eff, counters.data_ptr(), R | (GD << 4) | (npw << 8) | (dbg << 12), wgs, st)
                        row[key + "_" + nm + "_us"] = timed(run_d, reps)
            print(json.dumps(row), flush=True)
    print("counters zero:", int(counters.abs().sum()) == 0)


if __name__ == "__main__":
    main()
