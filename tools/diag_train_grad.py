#!/usr/bin/env python
"""GPU-box diagnostic: run module compositions once on the HIP ops (fp32, cuda:0) and once on
the CPU stand-in (tests/cpu_backend.py, float64), print the relative error of every gradient.
Not a test; used to localise a composition-level discrepancy."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cpu_backend  # noqa: E402
import openscene_amd.ops as ops  # noqa: E402
from openscene_amd import synthetic as syn  # noqa: E402
from openscene_amd.sparse import SparseTensor  # noqa: E402
import openscene_amd.minkowski as ME  # noqa: E402
from openscene_amd.mink_unet import mink_unet  # noqa: E402

REAL = {n: getattr(ops, n) for n in cpu_backend._NAMES}


def use(cpu):
    for n in cpu_backend._NAMES:
        setattr(ops, n, getattr(cpu_backend, n) if cpu else REAL[n])


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def run(name, factory, coords, cin, train=True, seed=0):
    torch.manual_seed(seed)
    mod = factory()
    for m in mod.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    mod.train(train)
    g = torch.Generator().manual_seed(seed + 1)
    feats = torch.rand(coords.shape[0], cin, generator=g)
    state = {k: v.clone() for k, v in mod.state_dict().items()}
    res = {}
    for cpu in (True, False):
        use(cpu)
        m2 = factory()
        m2.load_state_dict(state)
        m2.train(train)
        dev = torch.device("cpu") if cpu else torch.device("cuda", 0)
        m2 = m2.double() if cpu else m2.to(dev)
        x = (feats.double() if cpu else feats.to(dev)).requires_grad_(True)
        out = m2(SparseTensor(x, torch.from_numpy(coords).to(dev)))
        F = out.F if isinstance(out, SparseTensor) else out
        tgt = torch.randn(F.shape, generator=torch.Generator().manual_seed(seed + 2), dtype=torch.float64)
        (F * tgt.to(F.dtype).to(dev)).sum().backward()
        res[cpu] = (F.detach(), x.grad, {n: p.grad for n, p in m2.named_parameters()})
    use(False)
    print("== %s (train=%s, N=%d)" % (name, train, coords.shape[0]))
    print("   out %.2e   d/dx %.2e" % (rel(res[False][0], res[True][0]), rel(res[False][1], res[True][1])))
    worst = 0
    for n in res[True][2]:
        e = rel(res[False][2][n], res[True][2][n])
        worst = max(worst, e)
        if e > 2e-5 or os.environ.get("DIAG_ALL"):
            print("   %-40s %.2e" % (n, e))
    print("   worst param grad %.2e" % worst)


class Seq(torch.nn.Module):
    def __init__(self, *mods):
        super().__init__()
        self.m = torch.nn.ModuleList(mods)

    def forward(self, x):
        for m in self.m:
            x = m(x)
        return x


class DownUp(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c0 = ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3)
        self.b0 = ME.MinkowskiBatchNorm(32)
        self.down = ME.MinkowskiConvolution(32, 32, kernel_size=2, stride=2, dimension=3)
        self.b1 = ME.MinkowskiBatchNorm(32)
        self.up = ME.MinkowskiConvolutionTranspose(32, 32, kernel_size=2, stride=2, dimension=3)
        self.b2 = ME.MinkowskiBatchNorm(32)
        self.relu = ME.MinkowskiReLU()
        self.fin = ME.MinkowskiConvolution(64, 16, kernel_size=1, dimension=3)

    def forward(self, x):
        a = self.relu(self.b0(self.c0(x)))
        d = self.relu(self.b1(self.down(a)))
        u = self.relu(self.b2(self.up(d)))
        return self.fin(ME.cat(u, a))


def main():
    coords = syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(1, n_pts=4000), 0.05), 1)])
    big = syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(2, n_pts=40000), 0.03), 2)])
    for c, tag in ((coords, "small"), (big, "big")):
        run("conv3(8->32) " + tag, lambda: Seq(ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3)), c, 8)
        run("conv3+BN " + tag, lambda: Seq(ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3),
                                            ME.MinkowskiBatchNorm(32)), c, 8)
        run("conv3+BN eval " + tag, lambda: Seq(ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3),
                                                 ME.MinkowskiBatchNorm(32)), c, 8, train=False)
        run("conv3+BN+ReLU+conv3+BN " + tag, lambda: Seq(ME.MinkowskiConvolution(8, 32, kernel_size=3, dimension=3),
                                                          ME.MinkowskiBatchNorm(32), ME.MinkowskiReLU(),
                                                          ME.MinkowskiConvolution(32, 32, kernel_size=3, dimension=3),
                                                          ME.MinkowskiBatchNorm(32)), c, 8)
        run("BasicBlock(32) " + tag, lambda: Seq(ME.BasicBlock(32, 32, dimension=3)), c, 32)
        run("BasicBlock x2 " + tag, lambda: Seq(ME.BasicBlock(32, 32, dimension=3), ME.BasicBlock(32, 32, dimension=3)), c, 32)
        run("down/up/cat " + tag, DownUp, c, 8)
    os.environ["DIAG_ALL"] = "1"
    run("MinkUNet14A train", lambda: mink_unet(3, 32, 3, "MinkUNet14A"), coords, 3)
    run("MinkUNet14A eval", lambda: mink_unet(3, 32, 3, "MinkUNet14A"), coords, 3, train=False)


if __name__ == "__main__":
    main()
