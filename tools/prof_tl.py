#!/usr/bin/env python
"""Phase timers of the tile-list convolution kernel (wave 0 of every tile, s_memtime ticks) through the tools-only
entry point osn_dbg_spconv_fwd_tl_prof.  Prints mean ticks per tile and per 32-pair step for each phase."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import _lib, ops, synthetic as syn  # noqa: E402
from openscene_amd.sparse import CoordinateManager  # noqa: E402

PHASES = ["prologue", "list load", "gather wait+split", "barrier A", "barrier B",
          "B wait+frags+MFMA", "tile RMW", "epilogue", "B issue", "stage write+gather issue"]
PHASES2 = ["prologue", "list load", "counted wait (DMA landed)", "barrier", "tail-fragment issue", "acc/frag reads + k-step 0",
           "DMA + prefetch issue", "k-steps 1.. + write-back", "epilogue", "-"]


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    if not hasattr(lib, "osn_dbg_spconv_fwd_tl_prof"):
        raise SystemExit("the product library has no phase-timer entry point: rebuild with  OSN_BUILD_TOOLS=1 python -m openscene_amd.build --force")
    fn = lib.osn_dbg_spconv_fwd_tl_prof
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    fn.restype = i32
    fn.argtypes = [vp, i64, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, ctypes.c_size_t, vp, vp]
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([vox])).to(dev))
    variant = 0
    for stride, cin, cout in ((1, 96, 96), (1, 128, 96), (2, 96, 96), (2, 32, 32)):
        n = cm.size(stride)
        tiles = cm.kmap_tiles(stride, stride, 3)[0]
        tl = ops.tile_lists(tiles[1], out_rows=tiles[0])
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        out = torch.empty(n, cout, device=dev)
        prof = torch.zeros(512, 10, dtype=torch.int64, device=dev)
        ws = torch.zeros(256 + 16 * n * cout * 4, dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            rc = fn(x.data_ptr(), n, wf.data_ptr(), tl.buf.data_ptr(), tl.out_rows.data_ptr(), out.data_ptr(), n, 27,
                    cin, cout, tl.bm, ws.data_ptr(), ws.numel(), prof.data_ptr(), stream)
            assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn(x.data_ptr(), n, wf.data_ptr(), tl.buf.data_ptr(), tl.out_rows.data_ptr(), out.data_ptr(), n, 27,
           cin, cout, tl.bm, ws.data_ptr(), ws.numel(), prof.data_ptr(), stream)
        e1.record()
        torch.cuda.synchronize()
        print("kernel with timers: %.1f us" % (e0.elapsed_time(e1) * 1e3))
        p = prof.double().cpu()
        cnt = tl.counts().cpu()
        nchunk = 1 if variant else -(-cin // 128)
        steps = ((cnt + 31) // 32).sum().item() * nchunk
        offs = (cnt > 0).sum().item()
        tot = p.sum(1)
        print("stride %d %d->%d: %d tiles, %.1f offsets and %.1f steps per tile; %.0f ticks per workgroup (min %.0f max %.0f), %.0f per tile" % (
            stride, cin, cout, tl.n_tiles, offs / tl.n_tiles, steps / tl.n_tiles, tot.mean(), tot.min(), tot.max(), tot.sum() / tl.n_tiles))
        # p[w, i] = ticks workgroup w spent in phase i over ALL its tiles: per tile = sum / tiles, per step = sum / steps
        for i, name in enumerate(PHASES2 if variant else PHASES):
            print("   %-24s %9.0f ticks/tile  %5.1f %%   %8.0f per step" % (
                name, p[:, i].sum() / tl.n_tiles, 100 * p[:, i].sum() / tot.sum(), p[:, i].sum() / steps))


if __name__ == "__main__":
    main()
