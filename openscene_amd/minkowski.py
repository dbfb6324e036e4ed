"""The slice of the MinkowskiEngine Python surface that the reference touches,
re-implemented on the HIP kernels (SURVEY.md section 8(b) lists every symbol):

    import MinkowskiEngine as ME                          models/mink_unet.py:25
    from MinkowskiEngine.modules.resnet_block import BasicBlock, Bottleneck
    from MinkowskiEngine import SparseTensor              run/distill.py:18

``openscene_amd.install_minkowski_alias()`` registers this module under the
name ``MinkowskiEngine`` so those imports resolve unchanged.  Modules are plain
``nn.Module``s with ordinary ``nn.Parameter``s carrying ME's names and shapes
(``.kernel`` [K, Cin, Cout] -- 2-D [Cin, Cout] when K == 1 --, ``.bn`` =
``nn.BatchNorm1d``), so Adam, DDP, ``state_dict`` and the released checkpoints
work untouched.
"""
import math

import torch
import torch.nn as nn

from . import functional as F_
from .sparse import CoordinateManager, SparseTensor, cat  # noqa: F401  (re-exported ME names)

__version__ = "0.5.4+openscene_amd"


def _triple_ok(v, name):
    if isinstance(v, (list, tuple)):
        if len(set(v)) != 1:
            raise NotImplementedError("anisotropic %s %r is not used by the reference and not supported" % (name, v))
        v = v[0]
    return int(v)


class MinkowskiConvolutionBase(nn.Module):
    TRANSPOSED = False

    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, convolution_mode=None, dimension=None):
        super().__init__()
        if dimension is not None and dimension != 3:
            raise NotImplementedError("only 3-D sparse tensors are supported (dimension=%r)" % (dimension,))
        if kernel_generator is not None or expand_coordinates:
            raise NotImplementedError("custom kernel generators / coordinate expansion are outside the reference path")
        self.in_channels, self.out_channels = int(in_channels), int(out_channels)
        self.kernel_size = _triple_ok(kernel_size, "kernel_size")
        self.stride = _triple_ok(stride, "stride")
        self.dilation = _triple_ok(dilation, "dilation")
        if self.kernel_size < 1:
            raise ValueError("kernel_size must be >= 1")
        if self.stride not in (1, 2):
            raise NotImplementedError("stride %d: the reference only uses strides 1 and 2" % self.stride)
        self.dimension = 3
        self.kernel_volume = self.kernel_size ** 3
        shape = (self.in_channels, self.out_channels) if self.kernel_volume == 1 else \
            (self.kernel_volume, self.in_channels, self.out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(1, self.out_channels, dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME default: U(-s, s), s = 1/sqrt(C * K), C = Cin (conv) or Cout (transposed)
        with torch.no_grad():
            n = (self.out_channels if self.TRANSPOSED else self.in_channels) * self.kernel_volume
            s = 1.0 / math.sqrt(n)
            self.kernel.uniform_(-s, s)
            if self.bias is not None:
                self.bias.uniform_(-s, s)

    def forward(self, x):
        cm, s_in = x.coordinate_manager, x.tensor_stride
        if self.TRANSPOSED:
            if self.stride == 1:
                # ME swaps the in/out maps of EVERY transposed conv (odd kernels: mirrored offsets); the reference
                # only builds stride-2 transposed convs (models/mink_unet.py:76-99), so refuse rather than return
                # the un-mirrored convolution silently
                raise NotImplementedError("MinkowskiConvolutionTranspose with stride 1 is outside the reference path")
            else:
                if s_in % self.stride:
                    raise ValueError("cannot up-sample a tensor of stride %d by %d" % (s_in, self.stride))
                s_out = s_in // self.stride
                if not cm.has(s_out):
                    raise RuntimeError("transposed convolution needs the cached stride-%d map of the encoder" % s_out)
        else:
            s_out = s_in * self.stride
        maps = cm.kmap(s_in, s_out, self.kernel_size, self.dilation)
        tiles = cm.kmap_tiles(s_in, s_out, self.kernel_size, self.dilation)
        counts = cm.kmap_counts(s_in, s_out, self.kernel_size, self.dilation) if self.kernel.requires_grad else None
        lists = None
        if F_.CONV_MODE == "tl" and self.kernel_volume > 1 and F_.ops.tl_eligible(self.kernel_volume, self.in_channels,
                                                                                  self.out_channels):
            lists = cm.kmap_lists(s_in, s_out, self.kernel_size, self.dilation)
        out = F_.sparse_conv(x.F, self.kernel, maps, cm.size(s_out), tiles, counts, lists, transposed=self.TRANSPOSED,
                             fine_unique=self.kernel_size == 2 and self.stride == 2 and self.dilation == 1)
        if self.bias is not None:
            out = out + self.bias
        return SparseTensor(out, tensor_stride=s_out, coordinate_manager=cm)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d, dilation=%d" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.dilation)


class MinkowskiConvolution(MinkowskiConvolutionBase):
    TRANSPOSED = False


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    TRANSPOSED = True


class MinkowskiBatchNorm(nn.Module):
    """Holds ``self.bn = nn.BatchNorm1d`` exactly like ME (models/resnet_base.py:79-80
    initialises ``m.bn.weight``); the arithmetic runs in the HIP kernels."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return x._like(F_.batch_norm_act(x.F, self.bn))


class MinkowskiReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        # stand-alone ReLU of the unfused module chain (fused paths: BasicBlock, openscene_amd.mink_unet)
        return x._like(F_.relu(x.F))


def _outside_path(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **kw):
            super().__init__()
            raise NotImplementedError(
                "%s is only referenced by dead code of the reference (models/resnet_base.py:45-71, never "
                "constructed by MinkUNetBase) and is outside the accelerated path" % name)
    _Stub.__name__ = name
    return _Stub


MinkowskiAvgPooling = _outside_path("MinkowskiAvgPooling")
MinkowskiGlobalMaxPooling = _outside_path("MinkowskiGlobalMaxPooling")
MinkowskiLinear = _outside_path("MinkowskiLinear")


# ---- ME.modules.resnet_block ------------------------------------------------
class BasicBlock(nn.Module):
    """conv3-bn-relu-conv3-bn-(+residual)-relu, expansion 1 ([ME] modules/resnet_block.py;
    member names conv1/norm1/conv2/norm2/downsample are the checkpoint keys).  BN, the
    residual add and the ReLU run as ONE fused kernel pass per BN."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=1, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.conv1(x)
        out = out._like(F_.batch_norm_act(out.F, self.norm1.bn, relu=True))
        out = self.conv2(out)
        if self.downsample is not None:
            ds = self.downsample
            if isinstance(ds, nn.Sequential) and len(ds) == 2 and isinstance(ds[1], MinkowskiBatchNorm):
                res = ds[0](x)
                res = F_.batch_norm_act(res.F, ds[1].bn)
            else:
                res = ds(x).F
        else:
            res = x.F
        return out._like(F_.batch_norm_act(out.F, self.norm2.bn, residual=res, relu=True))


class Bottleneck(nn.Module):
    """1x1 - 3x3x3 - 1x1 (expansion 4); imported by the reference, used by no shipped config."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, bn_momentum=0.1, dimension=-1):
        super().__init__()
        self.conv1 = MinkowskiConvolution(inplanes, planes, kernel_size=1, dimension=dimension)
        self.norm1 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv2 = MinkowskiConvolution(planes, planes, kernel_size=3, stride=stride, dilation=dilation,
                                          dimension=dimension)
        self.norm2 = MinkowskiBatchNorm(planes, momentum=bn_momentum)
        self.conv3 = MinkowskiConvolution(planes, planes * self.expansion, kernel_size=1, dimension=dimension)
        self.norm3 = MinkowskiBatchNorm(planes * self.expansion, momentum=bn_momentum)
        self.relu = MinkowskiReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.conv1(x)
        out = out._like(F_.batch_norm_act(out.F, self.norm1.bn, relu=True))
        out = self.conv2(out)
        out = out._like(F_.batch_norm_act(out.F, self.norm2.bn, relu=True))
        out = self.conv3(out)
        res = self.downsample(x).F if self.downsample is not None else x.F
        return out._like(F_.batch_norm_act(out.F, self.norm3.bn, residual=res, relu=True))


# ---- ME.utils ------------------------------------------------------------------
def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
    """ME.utils.kaiming_normal_ (models/resnet_base.py:76): fan_in = Cin*K, fan_out = Cout*K for a
    [K, Cin, Cout] kernel (plain [Cin, Cout] when K == 1)."""
    if tensor.dim() == 3:
        k, cin, cout = tensor.shape
    elif tensor.dim() == 2:
        k, (cin, cout) = 1, tensor.shape
    else:
        raise ValueError("kernel tensors are [K, Cin, Cout] or [Cin, Cout]")
    fan = cin * k if mode == "fan_in" else cout * k
    gain = nn.init.calculate_gain(nonlinearity, a)
    std = gain / math.sqrt(fan)
    with torch.no_grad():
        return tensor.normal_(0, std)
