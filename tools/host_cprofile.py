#!/usr/bin/env python
"""Where the HOST time of a training step goes on the GPU box: cProfile over N steps of the bench's step (kernels are
asynchronous, so with a GPU that keeps up this is the launch path: wrappers, torch.empty, autograd, ctypes)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from openscene_amd.disnet import DisNet  # noqa: E402
from openscene_amd.sparse import SparseTensor  # noqa: E402


def main():
    dev = torch.device("cuda", 0)

    class Cfg:
        arch_3d = "MinkUNet18A"
        feature_2d_extractor = "openseg"

    torch.manual_seed(1463)
    model = DisNet(Cfg()).to(dev)
    coords = bench.build_scene(1463, dev)
    n = coords.shape[0]
    feats = torch.ones(n, 3, device=dev)
    g = torch.Generator().manual_seed(7)
    sel = torch.randperm(n, generator=g)[:20000].sort()[0].to(dev)
    target = torch.nn.functional.normalize(torch.randn(20000, 768, generator=g), dim=1).to(dev)
    cos = torch.nn.CosineSimilarity()
    optim = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)

    def step():
        out = model(SparseTensor(feats, coords))
        loss = (1 - cos(out.index_select(0, sel), target)).mean()
        optim.zero_grad(set_to_none=True)
        loss.backward()
        optim.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    steps = int(os.environ.get("STEPS", "20"))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("issue %.2f ms/step, issue+drain %.2f ms/step" % (t_issue * 1e3 / steps, t_all * 1e3 / steps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
