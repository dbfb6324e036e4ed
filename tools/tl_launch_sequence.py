#!/usr/bin/env python
"""The order in which ONE training step launches a tile-list kernel instance, from the executor's stage program (no GPU):
forward stages in program order, then the input gradients in reverse order.  Prints a JSON list of labels for
tools/rocpd_stats.py --by-position, e.g.

    python tools/tl_launch_sequence.py MinkUNet18A 768 3 3 > labels.json     # spconv_tl_kernel<3, 3, ...>

(instance <NW, KS>: NW = 32-column groups of the written side, KS = k-steps per channel chunk of the gathered side)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S100K = [100999, 47618, 12868, 3052, 700]


def instance(c_gather, c_write):
    nw, pad_best = 1, 1 << 30
    for w in (4, 3, 2, 1):
        pd = -(-c_write // (32 * w)) * 32 * w - c_write
        if pd < pad_best:
            nw, pad_best = w, pd
    ns = (c_gather + 31) // 32
    ks = ns if ns <= 4 else (4 if ns % 4 == 0 else (3 if ns % 3 == 0 else 4))
    return nw, ks


def main():
    arch, out_dim, nw, ks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from openscene_amd import executor as E
    from openscene_amd.mink_unet import mink_unet
    ex = E.for_model(mink_unet(3, out_dim, 3, arch))
    kern = ex.kernels(S100K, training=True)
    ops = ex.program.ops
    seq = []
    for (i, kf, kd, kw), o in zip(kern, ops):
        if kf == "tl" and instance(o["cin"], o["cout"]) == (nw, ks):
            seq.append("op %d fwd   K=%d %d->%d rows %d" % (i, o["K"], o["cin"], o["cout"], S100K[o["lvl_out"]]))
    for (i, kf, kd, kw), o in reversed(list(zip(kern, ops))):
        if kd == "tl" and instance(o["cout"], o["cin"]) == (nw, ks):
            seq.append("op %d dgrad K=%d %d->%d rows %d" % (i, o["K"], o["cout"], o["cin"], S100K[o["lvl_in"]]))
    print(json.dumps(seq))


if __name__ == "__main__":
    main()
