"""Hand-derivable known-answer tests for the (ME-unpinned) part of the oracle:
coordinate maps, kernel maps and sparse convolution."""
import numpy as np
import torch

from oracle import coords as oc
from oracle import sparse_ops as so


def c4(xyz, b=0):
    xyz = np.asarray(xyz, dtype=np.int32).reshape(-1, 3)
    return np.concatenate([np.full((xyz.shape[0], 1), b, np.int32), xyz], 1)


def test_offsets_enumeration():
    o3 = oc.kernel_offsets(3)
    assert o3.shape == (27, 3) and tuple(o3[13]) == (0, 0, 0)
    assert tuple(o3[0]) == (-1, -1, -1) and tuple(o3[1]) == (0, -1, -1)   # x fastest
    assert tuple(oc.kernel_offsets(5)[62]) == (0, 0, 0)
    o2 = oc.kernel_offsets(2, tensor_stride=4)
    assert tuple(o2[1]) == (4, 0, 0) and tuple(o2[7]) == (4, 4, 4) and tuple(o2[0]) == (0, 0, 0)


def test_single_voxel_centre_only():
    cm = oc.CoordinateManager(c4([[5, 6, 7]]))
    t = cm.kmap(1, 1, 3)
    assert (t >= 0).sum() == 1 and t[13, 0] == 0
    W = torch.stack([torch.eye(4) * k for k in range(27)])
    x = torch.arange(4.0).reshape(1, 4) + 1
    assert torch.equal(so.sparse_conv(x, W, t), 13 * x)


def test_two_voxel_line():
    cm = oc.CoordinateManager(c4([[3, 3, 3], [4, 3, 3]]))
    t = cm.kmap(1, 1, 3)
    ks = sorted(set(np.nonzero(t >= 0)[0].tolist()))
    assert ks == [12, 13, 14]
    assert (t[13] >= 0).sum() == 2 and (t[12] >= 0).sum() == 1 and (t[14] >= 0).sum() == 1
    # offset 14 = (+1,0,0): out row 0 (x=3) reads in row 1 (x=4)
    assert t[14, 0] == 1 and t[12, 1] == 0


def test_dense_cube_pair_count_and_symmetry():
    g = np.stack(np.meshgrid(*[np.arange(3)] * 3, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(0)
    g = g[rng.permutation(27)]
    cm = oc.CoordinateManager(c4(g))
    t = cm.kmap(1, 1, 3)
    assert (t >= 0).sum() == 343            # 7**3
    # (i,o) in map_k  <=>  (o,i) in map_{K-1-k}
    tt = oc.transpose_table(t, 27)
    assert np.array_equal(tt, t[::-1])


def test_stride2_block_and_transpose():
    g = np.stack(np.meshgrid(*[np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 3) + 4
    cm = oc.CoordinateManager(c4(g))
    assert cm.level(2).shape[0] == 1 and tuple(cm.level(2)[0]) == (0, 4, 4, 4)
    down = cm.kmap(1, 2, 2)
    assert down.shape == (8, 1) and sorted(down[:, 0].tolist()) == list(range(8))
    up = cm.kmap(2, 1, 2)
    assert up.shape == (8, 8) and (up >= 0).sum() == 8 and np.all((up >= 0).sum(0) == 1)
    # transposed conv returns each child in * W[octant]
    W = torch.arange(8.0).reshape(8, 1, 1) + 1
    out = so.sparse_conv(torch.tensor([[2.0]]), W, up)
    for r in range(8):
        k = (g[r, 0] - 4) + 2 * (g[r, 1] - 4) + 4 * (g[r, 2] - 4)
        assert out[r, 0] == 2.0 * (k + 1)


def test_negative_coords_floor():
    cm = oc.CoordinateManager(c4([[-1, -2, -3], [0, 0, 0], [-4, 1, 5]]))
    lv = cm.level(2)
    assert sorted(map(tuple, lv.tolist())) == sorted([(0, -2, -2, -4), (0, 0, 0, 0), (0, -4, 0, 4)])


def test_unet_shapes_and_param_count():
    p = so.init_params("MinkUNet18A", 3, 768)
    n_conv = sum(v.numel() for k, v in p.items() if k.endswith(".kernel"))
    assert n_conv == 15_554_272
    p = so.init_params("MinkUNet34C", 3, 768)
    assert sum(v.numel() for k, v in p.items() if k.endswith(".kernel")) == 37_910_240
    rng = np.random.default_rng(1)
    g = np.unique(rng.integers(0, 12, (300, 3)), axis=0)
    p = so.init_params("MinkUNet14A", 3, 8, dtype=torch.float64)
    out = so.unet_forward(p, torch.ones(g.shape[0], 3, dtype=torch.float64), c4(g), "MinkUNet14A")
    assert out.shape == (g.shape[0], 8) and torch.isfinite(out).all()
