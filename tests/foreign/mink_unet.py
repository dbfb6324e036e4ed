"""A U-Net written by "someone else" against the MinkowskiEngine API only (`import MinkowskiEngine as ME`), with the attribute
names and the module-by-module, UN-fused dataflow of the reference's models/mink_unet.py:116-174 (conv, then BN, then ReLU as
three module calls; `ME.cat`; `.F` at the end).  It stands in for the reference's file on the GPU box, where /root/reference
does not exist: tests and bench.py's `drop_in_step` phase run it through `openscene_amd.install_minkowski_alias()` to measure
what a maintainer gets with the call sites of run/distill.py unchanged.  Nothing in here imports openscene_amd."""
import MinkowskiEngine as ME
import torch.nn as nn
from MinkowskiEngine.modules.resnet_block import BasicBlock

ENC = ("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")
DEC = ("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")


class MinkUNetBase(nn.Module):
    BLOCK, PLANES, LAYERS, INIT_DIM = None, None, (2,) * 8, 32

    def __init__(self, in_channels, out_channels, D=3):
        nn.Module.__init__(self)
        self.D = D
        w = self.INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, w, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(w)
        skips = [w]
        for i, name in enumerate(ENC):
            self.add_module(name, ME.MinkowskiConvolution(w, w, kernel_size=2, stride=2, dimension=D))
            self.add_module("bn%d" % (i + 1), ME.MinkowskiBatchNorm(w))
            w = self._layer("block%d" % (i + 1), w, self.PLANES[i], self.LAYERS[i])
            skips.append(w)
        skips.pop()
        for i, name in enumerate(DEC):
            p = self.PLANES[4 + i]
            self.add_module(name, ME.MinkowskiConvolutionTranspose(w, p, kernel_size=2, stride=2, dimension=D))
            self.add_module("bntr%d" % (4 + i), ME.MinkowskiBatchNorm(p))
            w = self._layer("block%d" % (5 + i), p + skips.pop(), p, self.LAYERS[4 + i])
        self.final = ME.MinkowskiConvolution(w, out_channels, kernel_size=1, dimension=D)
        self.relu = ME.MinkowskiReLU(inplace=True)
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.utils.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")

    def _layer(self, name, w_in, planes, n):
        w_out = planes * self.BLOCK.expansion
        short = None
        if w_in != w_out:
            short = nn.Sequential(ME.MinkowskiConvolution(w_in, w_out, kernel_size=1, stride=1, dimension=self.D),
                                  ME.MinkowskiBatchNorm(w_out))
        seq = [self.BLOCK(w_in, planes, stride=1, dilation=1, downsample=short, dimension=self.D)]
        seq += [self.BLOCK(w_out, planes, stride=1, dilation=1, dimension=self.D) for _ in range(n - 1)]
        self.add_module(name, nn.Sequential(*seq))
        return w_out

    def forward(self, x):
        y = self.relu(self.bn0(self.conv0p1s1(x)))
        kept = [y]
        for i, name in enumerate(ENC):
            y = getattr(self, name)(y)
            y = getattr(self, "bn%d" % (i + 1))(y)
            y = self.relu(y)
            y = getattr(self, "block%d" % (i + 1))(y)
            kept.append(y)
        kept.pop()
        for i, name in enumerate(DEC):
            y = getattr(self, name)(y)
            y = getattr(self, "bntr%d" % (4 + i))(y)
            y = self.relu(y)
            y = ME.cat(y, kept.pop())
            y = getattr(self, "block%d" % (5 + i))(y)
        return self.final(y).F


class MinkUNet14A(MinkUNetBase):
    BLOCK, LAYERS, PLANES = BasicBlock, (1,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet18A(MinkUNetBase):
    BLOCK, LAYERS, PLANES = BasicBlock, (2,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)


class MinkUNet34C(MinkUNetBase):
    BLOCK, LAYERS, PLANES = BasicBlock, (2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 96, 96)
