"""torch-CPU restatement of the sparse ops and the MinkUNet dataflow (test
oracle; float64-capable; PARITY UNPINNED vs MinkowskiEngine, see
oracle/__init__.py).

Follows:
  * ME convolution semantics: per kernel offset
    gather -> mm -> scatter-add, W laid out [K, Cin, Cout] ([Cin, Cout] if K=1);
  * ``models/mink_unet.py:44-114`` (layer plan) and ``:116-174`` (dataflow);
  * ``models/resnet_base.py:82-118`` (_make_layer: 1x1 conv + BN downsample
    when channels change);
  * ME ``modules/resnet_block.py`` BasicBlock: conv-bn-relu-conv-bn-(+res)-relu.
It is also the ``cpu_baseline`` of bench.py (kind "port"): this is the same
per-offset gather -> BLAS mm -> index_add loop ME's CPU backend runs.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .coords import CoordinateManager

ARCH = {
    # name: (LAYERS, PLANES)   models/mink_unet.py:176-238
    "MinkUNet14A": ((1,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet14B": ((1,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet14C": ((1,) * 8, (32, 64, 128, 256, 192, 192, 128, 128)),
    "MinkUNet14D": ((1,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet18A": ((2,) * 8, (32, 64, 128, 256, 128, 128, 96, 96)),
    "MinkUNet18B": ((2,) * 8, (32, 64, 128, 256, 128, 128, 128, 128)),
    "MinkUNet18D": ((2,) * 8, (32, 64, 128, 256, 384, 384, 384, 384)),
    "MinkUNet34A": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 64)),
    "MinkUNet34B": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 64, 32)),
    "MinkUNet34C": ((2, 3, 4, 6, 2, 2, 2, 2), (32, 64, 128, 256, 256, 128, 96, 96)),
}
INIT_DIM = 32


def sparse_conv(feats, W, nbr):
    """out[o] = sum_k feats[nbr[k,o]] @ W[k];  nbr: [K, N_out] (numpy or tensor)."""
    nbr = torch.as_tensor(np.asarray(nbr)).long()
    if W.dim() == 2:
        W = W.unsqueeze(0)
    K, n_out = nbr.shape
    out = feats.new_zeros((n_out, W.shape[2]))
    for k in range(K):
        o = torch.nonzero(nbr[k] >= 0).reshape(-1)
        if o.numel() == 0:
            continue
        out = out.index_add(0, o, feats.index_select(0, nbr[k][o]) @ W[k])
    return out


def batch_norm(x, p, prefix, train, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], train, momentum, eps)


def layer_plan(arch):
    """[(block_name, n_blocks, inplanes, planes)] for block1..block8."""
    layers, planes = ARCH[arch]
    plan, inpl = [], INIT_DIM
    skips = [None, None, None, None, planes[2], planes[1], planes[0], INIT_DIM]
    for b in range(8):
        if b >= 4:
            inpl = planes[b] + skips[b]
        plan.append(("block%d" % (b + 1), layers[b], inpl, planes[b]))
        inpl = planes[b]
    return plan


def init_params(arch="MinkUNet18A", in_channels=3, out_channels=20, seed=0, dtype=torch.float32):
    """state-dict with the reference's parameter names and init scheme
    (``models/resnet_base.py:73-80``: kaiming-normal fan_out on
    MinkowskiConvolution kernels; transposed convs keep ME's default uniform
    init; BN weight 1 / bias 0).  Keys carry no ``net3d.`` prefix."""
    g = torch.Generator().manual_seed(seed)
    layers, planes = ARCH[arch]
    p = {}

    def conv(name, K, cin, cout, transposed=False):
        shape = (cin, cout) if K == 1 else (K, cin, cout)
        if transposed:
            s = 1.0 / np.sqrt(cout * K)
            w = (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * s
        else:
            w = torch.randn(shape, generator=g, dtype=torch.float64) * np.sqrt(2.0 / (cout * K))
        p[name + ".kernel"] = w.to(dtype)

    def bn(name, c):
        p[name + ".bn.weight"] = torch.ones(c, dtype=dtype)
        p[name + ".bn.bias"] = torch.zeros(c, dtype=dtype)
        p[name + ".bn.running_mean"] = torch.zeros(c, dtype=dtype)
        p[name + ".bn.running_var"] = torch.ones(c, dtype=dtype)

    conv("conv0p1s1", 125, in_channels, INIT_DIM); bn("bn0", INIT_DIM)
    plan = layer_plan(arch)
    down = ["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]
    up = ["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]
    prev = INIT_DIM
    for b, (bname, nblk, inpl, pl) in enumerate(plan):
        if b < 4:
            conv(down[b], 8, prev, prev); bn("bn%d" % (b + 1), prev)
        else:
            conv(up[b - 4], 8, prev, planes[b], transposed=True); bn("bntr%d" % b, planes[b])
        for j in range(nblk):
            cin = inpl if j == 0 else pl
            conv("%s.%d.conv1" % (bname, j), 27, cin, pl); bn("%s.%d.norm1" % (bname, j), pl)
            conv("%s.%d.conv2" % (bname, j), 27, pl, pl); bn("%s.%d.norm2" % (bname, j), pl)
            if j == 0 and cin != pl:
                conv("%s.0.downsample.0" % bname, 1, cin, pl); bn("%s.0.downsample.1" % bname, pl)
        prev = pl
    conv("final", 1, planes[7], out_channels)
    return p


def unet_forward(p, feats, coords4, arch="MinkUNet18A", train=False, cm=None, relu_masks=None, record_masks=None):
    """models/mink_unet.py:116-174 on (feats [N,Cin], coords4 int32 [N,4]) -> [N, out].

    relu_masks (optional): list of bool tensors, one per ReLU in call order.  When given, ReLU i
    is evaluated as ``y * relu_masks[i]`` -- i.e. with a PRESCRIBED activation pattern.  Tests use
    it to compare gradients of an fp32 run against this float64 oracle on the same pattern: a
    pre-activation within fp32 rounding of zero legitimately flips its ReLU between precisions,
    and one flipped element already moves a gradient's relative L2 error by ~1/sqrt(#elements).

    record_masks (optional): a list that receives this evaluation's OWN activation pattern
    (bool tensor ``y > 0`` per ReLU, call order) -- tests count how many elements of the fp32
    run's pattern differ from the float64 one."""
    cm = cm or CoordinateManager(np.asarray(coords4))
    plan = layer_plan(arch)
    masks = list(relu_masks) if relu_masks is not None else None

    def act(y):
        if record_masks is not None:
            record_masks.append(y.detach() > 0)
        if masks is None:
            return F.relu(y)
        m = masks.pop(0)
        return y * m.to(y.dtype)

    def bnrelu(x, name, relu=True):
        y = batch_norm(x, p, name + ".bn", train)
        return act(y) if relu else y

    def block(x, bname, nblk, stride):
        t = cm.kmap(stride, stride, 3)
        for j in range(nblk):
            pre = "%s.%d" % (bname, j)
            res = x
            y = bnrelu(sparse_conv(x, p[pre + ".conv1.kernel"], t), pre + ".norm1")
            y = bnrelu(sparse_conv(y, p[pre + ".conv2.kernel"], t), pre + ".norm2", relu=False)
            if (pre + ".downsample.0.kernel") in p:
                res = sparse_conv(x, p[pre + ".downsample.0.kernel"],
                                  np.arange(x.shape[0], dtype=np.int32)[None])
                res = bnrelu(res, pre + ".downsample.1", relu=False)
            x = act(y + res)
        return x

    x = bnrelu(sparse_conv(feats, p["conv0p1s1.kernel"], cm.kmap(1, 1, 5)), "bn0")
    skips = [x]
    down = ["conv1p1s2", "conv2p2s2", "conv3p4s2", "conv4p8s2"]
    up = ["convtr4p16s2", "convtr5p8s2", "convtr6p4s2", "convtr7p2s2"]
    s = 1
    for b in range(4):
        x = bnrelu(sparse_conv(x, p[down[b] + ".kernel"], cm.kmap(s, s * 2, 2)), "bn%d" % (b + 1))
        s *= 2
        x = block(x, plan[b][0], plan[b][1], s)
        skips.append(x)
    for b in range(4, 8):
        x = bnrelu(sparse_conv(x, p[up[b - 4] + ".kernel"], cm.kmap(s, s // 2, 2)), "bntr%d" % b)
        s //= 2
        x = torch.cat([x, skips[7 - b]], 1)
        x = block(x, plan[b][0], plan[b][1], s)
    return sparse_conv(x, p["final.kernel"], np.arange(x.shape[0], dtype=np.int32)[None])
