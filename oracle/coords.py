"""numpy restatement of the coordinate-map / kernel-map semantics the reference
relies on (test oracle, CPU only).  PARITY UNPINNED vs MinkowskiEngine -- see
oracle/__init__.py.  Restates ME v0.5.4 semantics recalled in SURVEY.md
appendix C items 1-4, exercised by ``models/mink_unet.py:47-113``:

  * coordinates are int32 rows (batch, x, y, z); unique rows keep caller order;
  * a stride-2 conv creates the coarse map floor(c / 2s) * 2s (unique);
  * kernel offsets enumerate with the FIRST spatial axis fastest; odd kernels
    are centred, even kernels span [0, k); all scaled by the tensor stride;
  * kernel map entry: in = find(coord[out] + offset_k);
  * a transposed conv uses the swapped map of the matching strided conv and
    lands on the cached finer map.

Layout choice of this repo (both oracle and HIP): the kernel map is a dense
k-major neighbour table ``nbr[K, N_out]`` int32 with -1 for "no input voxel".
Coarse rows are ordered by their first (lowest-index) child row; ME leaves
that order unspecified, and every consumer is order-independent.
"""
import numpy as np

_B = 1 << 15  # bias so that negative coordinates pack monotonically


def pack(coords4):
    c = np.asarray(coords4).astype(np.int64)
    return (c[:, 0] << 48) | ((c[:, 1] + _B) << 32) | ((c[:, 2] + _B) << 16) | (c[:, 3] + _B)


def floor_to_stride(coords4, stride):
    c = np.asarray(coords4).astype(np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], stride) * stride
    return c.astype(np.int32)


def unique_first(coords4):
    """(unique rows in first-occurrence order, inverse map, first index)."""
    key = pack(coords4)
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")          # rank unique keys by first occurrence
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    first_sorted = first[order]
    return (np.asarray(coords4)[first_sorted].astype(np.int32),
            rank[inv.reshape(-1)].astype(np.int32), first_sorted.astype(np.int64))


def stride_map(coords4, out_stride):
    """fine rows -> (coarse coords [M,4], parent [N] int32)."""
    coarse_all = floor_to_stride(coords4, out_stride)
    coarse, parent, _ = unique_first(coarse_all)
    return coarse, parent


def kernel_offsets(ksize, tensor_stride=1, dilation=1):
    """[K,3] int offsets, x fastest (ME HYPER_CUBE region)."""
    r = np.arange(ksize)
    if ksize % 2 == 1:
        r = r - ksize // 2
    r = r * dilation * tensor_stride
    iz, iy, ix = np.meshgrid(r, r, r, indexing="ij")
    return np.stack([ix.reshape(-1), iy.reshape(-1), iz.reshape(-1)], 1).astype(np.int32)


def kernel_map(in_coords4, out_coords4, offsets):
    """nbr[K, N_out]: row of in_coords4 equal to out + offset_k, else -1."""
    in_key = pack(in_coords4)
    order = np.argsort(in_key, kind="stable")
    sk = in_key[order]
    out = np.asarray(out_coords4).astype(np.int64)
    K = offsets.shape[0]
    nbr = np.full((K, out.shape[0]), -1, dtype=np.int32)
    for k in range(K):
        q = out.copy()
        q[:, 1:] += offsets[k].astype(np.int64)
        qk = pack(q)
        pos = np.searchsorted(sk, qk)
        pos_c = np.minimum(pos, sk.size - 1)
        hit = (sk.size > 0) & (sk[pos_c] == qk) if sk.size else np.zeros(qk.shape, bool)
        nbr[k, hit] = order[pos_c[hit]]
    return nbr


def transpose_table(nbr, n_in):
    """tbl[k, i] = o  <=>  nbr[k, o] = i  (the map of the transposed operator)."""
    K = nbr.shape[0]
    t = np.full((K, n_in), -1, dtype=np.int32)
    for k in range(K):
        o = np.nonzero(nbr[k] >= 0)[0]
        t[k, nbr[k, o]] = o
    return t


class CoordinateManager:
    """Per-tensor cache of maps, mirroring ME's per-SparseTensor manager."""

    def __init__(self, coords4):
        coords4 = np.asarray(coords4, dtype=np.int32)
        uniq, inv, first = unique_first(coords4)
        assert uniq.shape[0] == coords4.shape[0], "oracle expects unique coordinates"
        self.coords = {1: coords4}
        self.parent = {}
        self._kmaps = {}

    def level(self, stride):
        if stride not in self.coords:
            fine = self.level(stride // 2)
            self.coords[stride], self.parent[stride] = stride_map(fine, stride)
        return self.coords[stride]

    def kmap(self, in_stride, out_stride, ksize):
        """neighbour table for a conv reading stride `in_stride`, writing `out_stride`."""
        key = (in_stride, out_stride, ksize)
        if key in self._kmaps:
            return self._kmaps[key]
        if out_stride >= in_stride:                       # ordinary (possibly strided) conv
            off = kernel_offsets(ksize, in_stride)
            t = kernel_map(self.level(in_stride), self.level(out_stride), off)
        else:                                             # transposed: swap the fine->coarse map
            fwd = self.kmap(out_stride, in_stride, ksize)
            t = transpose_table(fwd, self.level(out_stride).shape[0])
        self._kmaps[key] = t
        return t
