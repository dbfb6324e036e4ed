#!/usr/bin/env python
"""Where the HOST time of an inference pass goes on the GPU box (coordinate pyramid + maps + eval-mode forward of one S100k scene):
issue time vs issue + drain, the three phases of the map build (tools/maps_host_time.py's split), then cProfile over N passes."""
import cProfile
import os
import pstats
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

    class Cfg:
        arch_3d = os.environ.get("ARCH", "MinkUNet18A")
        feature_2d_extractor = "openseg"

    torch.manual_seed(1463)
    model = DisNet(Cfg()).to(dev).eval()
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = torch.from_numpy(syn.batch_coords([vox])).to(dev)
    feats = torch.ones(coords.shape[0], 3, device=dev)
    n = int(os.environ.get("PASSES", "30"))

    def one():
        with torch.no_grad():
            return model(SparseTensor(feats, coords))

    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("issue %.3f ms/pass, issue + drain %.3f ms/pass" % (t_issue * 1e3 / n, t_all * 1e3 / n))
    # per-pass: host time from the constructor's return (pyramid sizes are on the host) to the return of the forward call
    acc = [0.0, 0.0]
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = SparseTensor(feats, coords)
        t1 = time.perf_counter()
        with torch.no_grad():
            model(x)
        t2 = time.perf_counter()
        acc[0] += t1 - t0
        acc[1] += t2 - t1
    torch.cuda.synchronize()
    print("constructor (pyramid + size read-back) %.3f ms, model(x) host time (maps + forward queued) %.3f ms" % (acc[0] * 1e3 / n, acc[1] * 1e3 / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        one()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(40)
    st.sort_stats("cumulative").print_stats(30)


if __name__ == "__main__":
    main()
