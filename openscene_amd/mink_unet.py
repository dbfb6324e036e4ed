"""MinkUNet family on the HIP kernels -- host-side mirror of the reference's
``models/mink_unet.py`` (+ ``models/resnet_base.py::_make_layer`` /
``weight_initialization``): same class names, constructor signature, module tree,
parameter names/shapes/registration order (=> identical ``state_dict`` keys and
optimizer parameter order), same ``forward(x) -> Tensor[N_0, out_channels]`` in
input row order (models/mink_unet.py:116-174).

The network is described by two tables (LAYERS, PLANES; models/mink_unet.py:176-238)
and built by one loop; conv -> BN -> ReLU triples run as conv + ONE fused
BN/ReLU kernel pass, residual blocks fuse BN + add + ReLU.
"""
import torch.nn as nn

from . import executor
from . import functional as F_
from . import minkowski as ME
from .minkowski import BasicBlock

_DOWN = ("conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2")
_UP = ("convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2")


class MinkUNetBase(nn.Module):
    BLOCK = None
    PLANES = None
    DILATIONS = (1,) * 8
    LAYERS = (2,) * 8
    INIT_DIM = 32
    OUT_TENSOR_STRIDE = 1

    def __init__(self, in_channels, out_channels, D=3):
        super().__init__()
        if self.BLOCK is None or self.PLANES is None:
            raise TypeError("instantiate a concrete MinkUNet variant (e.g. MinkUNet18A)")
        if D != 3:
            raise NotImplementedError("only D=3")
        self.D = D
        self.network_initialization(in_channels, out_channels, D)
        self.weight_initialization()

    # -- construction ------------------------------------------------------------
    def _stage(self, planes, n_blocks, dilation=1):
        """models/resnet_base.py:82-118 with stride 1: a 1x1-conv + BN shortcut iff the width changes."""
        width = planes * self.BLOCK.expansion
        shortcut = None
        if self.inplanes != width:
            shortcut = nn.Sequential(ME.MinkowskiConvolution(self.inplanes, width, kernel_size=1, stride=1, dimension=self.D),
                                     ME.MinkowskiBatchNorm(width))
        blocks = [self.BLOCK(self.inplanes, planes, stride=1, dilation=dilation, downsample=shortcut, dimension=self.D)]
        self.inplanes = width
        blocks += [self.BLOCK(width, planes, stride=1, dilation=dilation, dimension=self.D) for _ in range(n_blocks - 1)]
        return nn.Sequential(*blocks)

    def network_initialization(self, in_channels, out_channels, D):
        P, L, E = self.PLANES, self.LAYERS, self.BLOCK.expansion
        self.inplanes = self.INIT_DIM
        self.conv0p1s1 = ME.MinkowskiConvolution(in_channels, self.inplanes, kernel_size=5, dimension=D)
        self.bn0 = ME.MinkowskiBatchNorm(self.inplanes)
        for i in range(4):                                   # encoder: k2s2 conv, BN, residual stage
            setattr(self, _DOWN[i], ME.MinkowskiConvolution(self.inplanes, self.inplanes, kernel_size=2, stride=2, dimension=D))
            setattr(self, "bn%d" % (i + 1), ME.MinkowskiBatchNorm(self.inplanes))
            setattr(self, "block%d" % (i + 1), self._stage(P[i], L[i]))
        skip_width = (P[2] * E, P[1] * E, P[0] * E, self.INIT_DIM)
        for i in range(4):                                   # decoder: k2s2 transposed conv, BN, cat skip, stage
            setattr(self, _UP[i], ME.MinkowskiConvolutionTranspose(self.inplanes, P[4 + i], kernel_size=2, stride=2, dimension=D))
            setattr(self, "bntr%d" % (4 + i), ME.MinkowskiBatchNorm(P[4 + i]))
            self.inplanes = P[4 + i] + skip_width[i]
            setattr(self, "block%d" % (5 + i), self._stage(P[4 + i], L[4 + i]))
        self.final = ME.MinkowskiConvolution(P[7] * E, out_channels, kernel_size=1, dimension=D)   # no bias (mink_unet.py:108-113)
        self.relu = ME.MinkowskiReLU(inplace=True)

    def weight_initialization(self):
        # models/resnet_base.py:73-80: kaiming fan_out on MinkowskiConvolution kernels only
        # (transposed convs keep ME's default uniform init), BN gamma 1 / beta 0
        for m in self.modules():
            if isinstance(m, ME.MinkowskiConvolution):
                ME.kaiming_normal_(m.kernel, mode="fan_out", nonlinearity="relu")
            if isinstance(m, ME.MinkowskiBatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)

    # -- dataflow ----------------------------------------------------------------
    @staticmethod
    def _conv_bn_relu(x, conv, norm):
        y = conv(x)
        return y._like(F_.batch_norm_act(y.F, norm.bn, relu=True))

    def forward(self, x, rows=None):
        """rows (optional, int64 row indices): return only output[rows] -- what run/distill.py:321-322 keeps of the output
        (`output_3d = model(sinput); output_3d = output_3d[mask]`).  With the executor the final 1x1 convolution then runs on those
        rows only (forward and both gradients); without it the full output is computed and indexed."""
        # one C call per pass (openscene_amd/executor.py) when the configuration allows it; otherwise -- and for the
        # reference's own models/mink_unet.py running through the MinkowskiEngine alias -- module by module
        ex = executor.for_model(self)
        if ex is not None and ex.usable(x, self):
            return ex.forward(self, x, rows=rows)
        with F_.deferred_bn_counters():
            out = self.final(self._forward(x)).F
        return out if rows is None else out.index_select(0, rows)

    def forward_features(self, x):
        """The input of the final 1x1 convolution (float32 [N_0, PLANES[7]], input row order): what
        ``openscene_amd.query.query_distill_fused`` folds the head into."""
        ex = executor.for_model(self)
        if ex is not None and ex.usable(x, self):
            out = ex.forward(self, x, features_only=True)
            if out is not None:
                return out
        with F_.deferred_bn_counters():
            return self._forward(x).F

    def _forward(self, x):
        x.coordinate_manager.prebuild()       # all level syncs first, then the host runs ahead of the GPU
        out = self._conv_bn_relu(x, self.conv0p1s1, self.bn0)
        skips = [out]
        for i in range(4):
            out = self._conv_bn_relu(out, getattr(self, _DOWN[i]), getattr(self, "bn%d" % (i + 1)))
            out = getattr(self, "block%d" % (i + 1))(out)
            skips.append(out)
        skips.pop()                                          # the bottleneck output is not a skip
        for i in range(4):
            out = self._conv_bn_relu(out, getattr(self, _UP[i]), getattr(self, "bntr%d" % (4 + i)))
            out = ME.cat(out, skips.pop())
            out = getattr(self, "block%d" % (5 + i))(out)
        return out


def _variant(name, block, layers, planes):
    return type(name, (MinkUNetBase,), {"BLOCK": block, "LAYERS": layers, "PLANES": planes, "__doc__":
                "%s: LAYERS=%s PLANES=%s (models/mink_unet.py:176-238)" % (name, layers, planes)})


_L14, _L18, _L34 = (1,) * 8, (2,) * 8, (2, 3, 4, 6, 2, 2, 2, 2)
MinkUNet14A = _variant("MinkUNet14A", BasicBlock, _L14, (32, 64, 128, 256, 128, 128, 96, 96))
MinkUNet14B = _variant("MinkUNet14B", BasicBlock, _L14, (32, 64, 128, 256, 128, 128, 128, 128))
MinkUNet14C = _variant("MinkUNet14C", BasicBlock, _L14, (32, 64, 128, 256, 192, 192, 128, 128))
MinkUNet14D = _variant("MinkUNet14D", BasicBlock, _L14, (32, 64, 128, 256, 384, 384, 384, 384))
MinkUNet18A = _variant("MinkUNet18A", BasicBlock, _L18, (32, 64, 128, 256, 128, 128, 96, 96))
MinkUNet18B = _variant("MinkUNet18B", BasicBlock, _L18, (32, 64, 128, 256, 128, 128, 128, 128))
MinkUNet18D = _variant("MinkUNet18D", BasicBlock, _L18, (32, 64, 128, 256, 384, 384, 384, 384))
MinkUNet34A = _variant("MinkUNet34A", BasicBlock, _L34, (32, 64, 128, 256, 256, 128, 64, 64))
MinkUNet34B = _variant("MinkUNet34B", BasicBlock, _L34, (32, 64, 128, 256, 256, 128, 64, 32))
MinkUNet34C = _variant("MinkUNet34C", BasicBlock, _L34, (32, 64, 128, 256, 256, 128, 96, 96))

_ARCHS = {c.__name__: c for c in (MinkUNet14A, MinkUNet14B, MinkUNet14C, MinkUNet14D, MinkUNet18A, MinkUNet18B,
                                  MinkUNet18D, MinkUNet34A, MinkUNet34B, MinkUNet34C)}


def mink_unet(in_channels=3, out_channels=20, D=3, arch="MinkUNet18A"):
    """Factory with the reference's signature (models/mink_unet.py:241-263)."""
    try:
        return _ARCHS[arch](in_channels, out_channels, D)
    except KeyError:
        raise Exception("architecture not supported yet".format(arch))
