#!/usr/bin/env python
"""Where the wall clock of `maps_only` goes (bench.py phase; the map phase of an inference pass): host time of the SparseTensor
constructor (pyramid: one C call + the read-back of the level sizes = the one synchronisation), host time of prebuild() (planning in
Python + one C call that queues every map; returns without waiting) and the device time left after it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import synthetic as syn  # noqa: E402
from openscene_amd.sparse import SparseTensor  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = torch.from_numpy(syn.batch_coords([vox])).to(dev)
    feats = torch.ones(coords.shape[0], 3, device=dev)
    reps = int(os.environ.get("REPS", "50"))
    for pairs in (False, "ws", True):
        acc = [0.0, 0.0, 0.0]
        for it in range(reps + 5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x = SparseTensor(feats, coords)
            t1 = time.perf_counter()
            x.coordinate_manager.prebuild(pairs=pairs)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            if it >= 5:
                acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2
        print("pairs=%-5s constructor (pyramid + size read-back) %.3f ms | prebuild host (plan + queue) %.3f ms | device tail %.3f ms | sum %.3f ms"
              % (pairs, acc[0] / reps * 1e3, acc[1] / reps * 1e3, acc[2] / reps * 1e3, sum(acc) / reps * 1e3), flush=True)


if __name__ == "__main__":
    main()
