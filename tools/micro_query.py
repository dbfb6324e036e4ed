#!/usr/bin/env python
"""Query micro-benchmark (SURVEY.md 8(d) Q shapes): ms and fraction of the 8 TB/s HBM roof."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd.query import query_distill  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    shapes = ((150000, 768, 20, False), (150000, 512, 20, False), (500000, 768, 160, True), (500000, 768, 160, False),
              (1000000, 768, 160, False), (500000, 768, 40, False), (500000, 768, 80, False))
    if os.environ.get("SHAPES") == "wide":                               # only the shapes query_wide_kernel takes
        shapes = tuple(s for s in shapes if s[2] > 64)
    for n_pts, d, c, scores in shapes:
        n_vox = n_pts // 2
        x = torch.randn(n_vox, d, generator=g).to(dev)
        idx = torch.randint(0, n_vox, (n_pts,), generator=g).to(dev)
        t = torch.nn.functional.normalize(torch.randn(c, d, generator=g), dim=1).half().to(dev)
        for variant in ("-",):
            for _ in range(2):
                query_distill(x, t, idx, return_scores=scores)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                query_distill(x, t, idx, return_scores=scores)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            byts = 4.0 * n_pts * d + 2.0 * c * d + 16.0 * n_pts + (2.0 * n_pts * c if scores else 0.0)
            print("n=%d d=%d c=%d scores=%s variant=%s: %.3f ms  %.1f %% of 8 TB/s" % (
                n_pts, d, c, scores, variant, ms, 100 * byts / (ms * 1e-3) / 8e12), flush=True)
        del x, idx


if __name__ == "__main__":
    main()
