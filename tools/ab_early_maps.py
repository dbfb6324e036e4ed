#!/usr/bin/env python
"""A/B of executor.EARLY_MAPS (inference: the tables the first stages read in front of the pass, every other map product on a second
stream beside those stages): outputs must be BITWISE equal; wall clock of `maps + eval-mode forward` (bench.py phase inference_fwd),
alternating the two settings.  ARCH=MinkUNet18A|MinkUNet34C, REPS=n."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import executor as ex  # noqa: E402
from openscene_amd import synthetic as syn  # noqa: E402
from openscene_amd.disnet import DisNet  # noqa: E402
from openscene_amd.sparse import SparseTensor  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    arch = os.environ.get("ARCH", "MinkUNet18A")
    reps = int(os.environ.get("REPS", "20"))

    class Cfg:
        arch_3d = arch
        feature_2d_extractor = "openseg"

    torch.manual_seed(1463)
    model = DisNet(Cfg()).to(dev).eval()
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = torch.from_numpy(syn.batch_coords([vox])).to(dev)
    feats = torch.ones(coords.shape[0], 3, device=dev)

    def infer():
        with torch.no_grad():
            return model(SparseTensor(feats, coords))

    outs = {}
    for on in (False, True):
        ex.EARLY_MAPS = on
        for _ in range(3):
            outs[on] = infer()
    torch.cuda.synchronize()
    equal = bool(torch.equal(outs[False], outs[True]))
    ms = {False: [], True: []}
    for _ in range(4):
        for on in (False, True):
            ex.EARLY_MAPS = on
            infer()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                infer()
            torch.cuda.synchronize()
            ms[on].append((time.perf_counter() - t0) * 1e3 / reps)
    print(json.dumps({"arch": arch, "voxels": int(coords.shape[0]), "bitwise_equal": equal,
                      "inference_fwd_ms_plain": [round(v, 3) for v in ms[False]],
                      "inference_fwd_ms_early_maps": [round(v, 3) for v in ms[True]]}), flush=True)
    if not equal:
        sys.exit(1)


if __name__ == "__main__":
    main()
