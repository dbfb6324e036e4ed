"""Independent cross-check of the "parity unpinned" part of the oracle (oracle/coords.py +
oracle/sparse_ops.py): under the stated kernel-offset convention (first spatial axis fastest; odd kernels
centred, even kernels spanning [0, k)) the sparse convolutions must equal torch's DENSE conv3d /
conv_transpose3d on a zero-filled grid, sampled at the active sites.  This ties the kernel-map
enumeration to the weight index (a swapped axis order or a mirrored offset would show up here); it
does not replace a MinkowskiEngine build for the convention itself."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import coords as oc
from oracle import sparse_ops as so


def cloud(seed, n=400, extent=12):
    rng = np.random.default_rng(seed)
    g = np.unique(rng.integers(0, extent, (n, 3)), axis=0)
    g = g[rng.permutation(g.shape[0])]
    return np.concatenate([np.zeros((g.shape[0], 1), np.int64), g], 1).astype(np.int32), extent


def dense_of(coords, feats, extent):
    """[N, C] rows at (b, x, y, z) -> dense [1, C, Z, Y, X] (x is the innermost = fastest axis)."""
    d = torch.zeros(1, feats.shape[1], extent, extent, extent, dtype=feats.dtype)
    c = torch.from_numpy(coords[:, 1:].astype(np.int64))
    d[0, :, c[:, 2], c[:, 1], c[:, 0]] = feats.T
    return d


def sample(dense, coords, scale=1):
    c = torch.from_numpy((coords[:, 1:] // scale).astype(np.int64))
    return dense[0, :, c[:, 2], c[:, 1], c[:, 0]].T


def test_stride1_k3_and_k5_match_dense_conv3d():
    coords, extent = cloud(1)
    cm = oc.CoordinateManager(coords)
    g = torch.Generator().manual_seed(0)
    for k in (3, 5):
        cin, cout = 5, 7
        feats = torch.randn(coords.shape[0], cin, generator=g, dtype=torch.float64)
        W = torch.randn(k ** 3, cin, cout, generator=g, dtype=torch.float64)
        got = so.sparse_conv(feats, W, cm.kmap(1, 1, k))
        # W[kidx] with kidx = ix + k * iy + k^2 * iz  ->  conv3d weight [cout, cin, kz, ky, kx]
        w = W.reshape(k, k, k, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
        ref = F.conv3d(dense_of(coords, feats, extent), w, padding=k // 2)
        assert torch.allclose(got, sample(ref, coords), rtol=1e-12, atol=1e-12)


def test_stride2_k2_and_its_transpose_match_dense():
    coords, extent = cloud(2)
    cm = oc.CoordinateManager(coords)
    g = torch.Generator().manual_seed(1)
    cin, cout = 4, 6
    feats = torch.randn(coords.shape[0], cin, generator=g, dtype=torch.float64)
    W = torch.randn(8, cin, cout, generator=g, dtype=torch.float64)
    coarse = cm.level(2)
    got = so.sparse_conv(feats, W, cm.kmap(1, 2, 2))
    w = W.reshape(2, 2, 2, cin, cout).permute(4, 3, 0, 1, 2).contiguous()
    ref = F.conv3d(dense_of(coords, feats, extent), w, stride=2)
    assert got.shape[0] == coarse.shape[0]
    assert torch.allclose(got, sample(ref, coarse, scale=2), rtol=1e-12, atol=1e-12)
    # every non-zero cell of the dense result is an active coarse voxel (no output site is missing)
    nz = (ref[0].abs().sum(0) > 0).nonzero()
    assert nz.shape[0] <= coarse.shape[0]
    # transposed conv back to the fine map: out[o] = in[parent(o)] @ Wt[octant(o)]
    Wt = torch.randn(8, cout, cin, generator=g, dtype=torch.float64)
    up = so.sparse_conv(got, Wt, cm.kmap(2, 1, 2))
    dense_coarse = torch.zeros(1, cout, extent // 2, extent // 2, extent // 2, dtype=torch.float64)
    cc = torch.from_numpy((coarse[:, 1:] // 2).astype(np.int64))
    dense_coarse[0, :, cc[:, 2], cc[:, 1], cc[:, 0]] = got.T
    wt = Wt.reshape(2, 2, 2, cout, cin).permute(3, 4, 0, 1, 2).contiguous()      # conv_transpose3d: [cin_t, cout_t, kz, ky, kx]
    ref_up = F.conv_transpose3d(dense_coarse, wt, stride=2)
    assert torch.allclose(up, sample(ref_up, coords), rtol=1e-12, atol=1e-12)
