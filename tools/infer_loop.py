#!/usr/bin/env python
"""N inference passes (coordinate pyramid + maps + eval-mode forward) of one S100k scene through MinkUNet18A / 768-d -- the workload of
bench.py's `inference_fwd` -- for rocprofv3 (tools/gpu_prof_infer.sh).  PASSES=n, ARCH=..."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import synthetic as syn  # noqa: E402
from openscene_amd.disnet import DisNet  # noqa: E402
from openscene_amd.sparse import SparseTensor  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    passes = int(os.environ.get("PASSES", "20"))

    class Cfg:
        arch_3d = os.environ.get("ARCH", "MinkUNet18A")
        feature_2d_extractor = "openseg"

    torch.manual_seed(1463)
    model = DisNet(Cfg()).to(dev).eval()
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = torch.from_numpy(syn.batch_coords([vox])).to(dev)
    feats = torch.ones(coords.shape[0], 3, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(SparseTensor(feats, coords))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            model(SparseTensor(feats, coords))
        torch.cuda.synchronize()
    print("inference_fwd %.3f ms per pass (%d passes, %d voxels)" % ((time.perf_counter() - t0) * 1e3 / passes, passes, coords.shape[0]), flush=True)


if __name__ == "__main__":
    main()
