#!/usr/bin/env python
"""1x1 convolution (osn_dense_fwd) stand-alone: HIP-event time, TF and error against an fp64 product, per shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import ops  # noqa: E402


def timed(fn, reps=20):
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
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    for n, cin, cout in ((100999, 96, 768), (100999, 96, 512), (100999, 768, 96), (100999, 128, 96), (100999, 96, 160), (100999, 64, 256),
                         (100999, 32, 256), (63, 96, 768), (100999, 96, 20)):
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(cin, cout, device=dev) * 0.05
        wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
        a = ops.dense_fwd(x, wf, cout)
        ref = x.double() @ w.double()
        t = timed(lambda: ops.dense_fwd(x, wf, cout))
        print("%6d rows %3d -> %3d: %.1f us (%.0f TF)  rel err %.1e  checksum %.9e" % (
            n, cin, cout, t, 2.0 * n * cin * cout / t / 1e6, ((a.double() - ref).abs().max() / ref.abs().max()).item(),
            a.double().sum().item()), flush=True)


if __name__ == "__main__":
    main()
