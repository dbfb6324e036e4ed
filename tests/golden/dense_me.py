"""A stand-in `MinkowskiEngine` made of torch's DENSE 3-D operators -- golden-vector tooling, not product code.

MinkowskiEngine is not part of /root/reference (an un-vendored dependency), so the reference's own model file
(models/mink_unet.py) cannot run on the real engine in the authoring container.  This module lets it run on an
INDEPENDENT implementation of the same operators: every sparse tensor is scattered onto a zero-filled grid, the
convolution is torch.nn.functional.conv3d / conv_transpose3d, and the result is sampled back at the active sites.
tests/golden/make_golden_unet.py imports the reference's models/mink_unet.py on top of it and stores inputs and
outputs as a fixture; oracle/sparse_ops.py (CPU) and the HIP path (GPU) are then checked against that fixture.

What this pins and what it cannot: the layer plan, skip order, residual / BatchNorm placement and parameter names
come from the reference's file itself; the arithmetic of every operator comes from torch's dense kernels (nothing
from oracle/ or openscene_amd/ is imported here).  The engine's CONVENTIONS are restated from its documentation
(SURVEY.md appendix C): kernel offset index = ix + k iy + k^2 iz, odd kernels centred, even kernels spanning [0, k),
a stride-2 convolution writes the sites floor(c / 2s) 2s, a transposed one writes the cached finer map, a
kernel_size=1 convolution stores its kernel as [Cin, Cout].  A real MinkowskiEngine build is still the only thing
that could pin those.

Surface provided: SparseTensor(.F, .C, .tensor_stride, +, +=), MinkowskiConvolution, MinkowskiConvolutionTranspose,
MinkowskiBatchNorm(.bn), MinkowskiReLU, cat, utils.kaiming_normal_, modules.resnet_block.BasicBlock / Bottleneck
(BasicBlock follows ME's modules/resnet_block.py: conv1-norm1-relu-conv2-norm2-(+downsample(x))-relu).
Coordinates must be non-negative and unique.
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Grid:
    """The coordinate maps of one input: level s holds the sites floor(c / s) s of the level below, sorted."""

    def __init__(self, coords):
        c = torch.as_tensor(coords).long()
        assert c.dim() == 2 and c.shape[1] == 4 and int(c.min()) >= 0
        assert torch.unique(c, dim=0).shape[0] == c.shape[0], "duplicate coordinates"
        self.batch = int(c[:, 0].max()) + 1
        self.extent = (int(c[:, 1:].max()) // 16 + 1) * 16          # four stride-2 levels
        self.levels = {1: c}

    def level(self, s):
        if s not in self.levels:
            c = self.level(s // 2).clone()
            c[:, 1:] = torch.div(c[:, 1:], s, rounding_mode="floor") * s
            self.levels[s] = torch.unique(c, dim=0)
        return self.levels[s]

    def scatter(self, rows, s):
        c = self.level(s)
        d = self.extent // s
        dense = rows.new_zeros((self.batch, rows.shape[1], d, d, d))
        return _index_put(dense, c[:, 0], c[:, 1:] // s, rows)

    def sample(self, dense, s):
        c = self.level(s)
        i = c[:, 1:] // s
        return dense[c[:, 0], :, i[:, 2], i[:, 1], i[:, 0]]


def _index_put(dense, b, i, rows):
    # dense[b, :, z, y, x] = rows  (x is the innermost, fastest axis), differentiable in `rows`
    bsz, ch, d = dense.shape[0], dense.shape[1], dense.shape[2]
    flat = dense.permute(0, 2, 3, 4, 1).reshape(-1, ch)
    lin = ((b * d + i[:, 2]) * d + i[:, 1]) * d + i[:, 0]
    flat = flat.index_put((lin,), rows)
    return flat.reshape(bsz, d, d, d, ch).permute(0, 4, 1, 2, 3)


class SparseTensor:
    def __init__(self, features, coordinates=None, tensor_stride=1, _grid=None):
        self.F = features
        self.tensor_stride = tensor_stride
        self._grid = _grid if _grid is not None else _Grid(coordinates)

    @property
    def C(self):
        return self._grid.level(self.tensor_stride)

    def _like(self, rows, stride=None):
        return SparseTensor(rows, tensor_stride=self.tensor_stride if stride is None else stride, _grid=self._grid)

    def _same(self, other):
        assert self._grid is other._grid and self.tensor_stride == other.tensor_stride

    def __add__(self, other):
        self._same(other)
        return self._like(self.F + other.F)

    def __iadd__(self, other):
        self._same(other)
        self.F = self.F + other.F
        return self


def cat(*tensors):
    for t in tensors[1:]:
        tensors[0]._same(t)
    return tensors[0]._like(torch.cat([t.F for t in tensors], 1))


class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=-1, **_kw):
        super().__init__()
        assert dimension == 3 and dilation == 1 and not bias
        self.k, self.stride = int(kernel_size), int(stride)
        vol = self.k ** 3
        shape = (in_channels, out_channels) if vol == 1 else (vol, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape))
        nn.init.normal_(self.kernel, 0.0, 0.05)

    def _w(self):
        k = self.k
        return self.kernel.reshape(k, k, k, self.kernel.shape[-2], self.kernel.shape[-1])      # [kz, ky, kx, cin, cout]


class MinkowskiConvolution(_ConvBase):
    def forward(self, x):
        g, s = x._grid, x.tensor_stride
        if self.k == 1:
            assert self.stride == 1
            return x._like(x.F @ self.kernel)
        w = self._w().permute(4, 3, 0, 1, 2)                                                   # [cout, cin, kz, ky, kx]
        dense = g.scatter(x.F, s)
        if self.stride == 1:
            assert self.k % 2 == 1
            return x._like(g.sample(F.conv3d(dense, w, padding=self.k // 2), s))
        assert self.stride == 2 and self.k == 2
        return x._like(g.sample(F.conv3d(dense, w, stride=2), 2 * s), 2 * s)


class MinkowskiConvolutionTranspose(_ConvBase):
    def forward(self, x):
        g, s = x._grid, x.tensor_stride
        assert self.stride == 2 and self.k == 2 and s >= 2
        w = self._w().permute(3, 4, 0, 1, 2)                                                   # [cin, cout, kz, ky, kx]
        up = F.conv_transpose3d(g.scatter(x.F, s), w, stride=2)
        return x._like(g.sample(up, s // 2), s // 2)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(self.bn(x.F))


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return x._like(F.relu(x.F))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation, dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        residual = x
        out = self.relu(self.norm1(self.conv1(x)))
        out = self.norm2(self.conv2(out))
        if self.downsample is not None:
            residual = self.downsample(x)
        out += residual
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, *a, **k):
        raise NotImplementedError("Bottleneck blocks are not used by the MinkUNet A-D variants of the hot path")


def install():
    """Register this module as `MinkowskiEngine` (+ .utils, .modules.resnet_block) in sys.modules."""
    me = types.ModuleType("MinkowskiEngine")
    for name in ("SparseTensor", "cat", "MinkowskiConvolution", "MinkowskiConvolutionTranspose", "MinkowskiBatchNorm",
                 "MinkowskiReLU"):
        setattr(me, name, globals()[name])
    me.__dense_emulation__ = True
    utils = types.ModuleType("MinkowskiEngine.utils")
    utils.kaiming_normal_ = lambda t, mode="fan_out", nonlinearity="relu": nn.init.normal_(t, 0.0, 0.05)
    modules = types.ModuleType("MinkowskiEngine.modules")
    rb = types.ModuleType("MinkowskiEngine.modules.resnet_block")
    rb.BasicBlock, rb.Bottleneck = BasicBlock, Bottleneck
    me.utils, me.modules, modules.resnet_block = utils, modules, rb
    sys.modules.update({"MinkowskiEngine": me, "MinkowskiEngine.utils": utils, "MinkowskiEngine.modules": modules,
                        "MinkowskiEngine.modules.resnet_block": rb})
    return me
