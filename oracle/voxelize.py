"""numpy restatement of the reference voxeliser (test oracle, CPU only).

Follows, line by line in behaviour (not in text):
  * ``dataset/voxelization_utils.py:9-22``   fnv_hash_vec  (multiply-then-xor)
  * ``dataset/voxelization_utils.py:25-41``  ravel_hash_vec
  * ``dataset/voxelization_utils.py:112-132`` sparse_quantize(return_index=True)
  * ``dataset/voxelizer.py:46-76``           get_transformation_matrix (RNG order)
  * ``dataset/voxelizer.py:97-140``          voxelize (clip_bound=None path)
Pinned against the reference's own code by tests/golden/make_golden.py.
"""
import numpy as np
from scipy.linalg import expm, norm

FNV_OFFSET = np.uint64(14695981039346656037)
FNV_PRIME = np.uint64(1099511628211)


def fnv_keys(int_coords):
    """[N, D] non-negative integral values (any float/int dtype) -> [N] uint64."""
    c = np.asarray(int_coords).astype(np.uint64)
    h = np.full(c.shape[0], FNV_OFFSET, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for j in range(c.shape[1]):
            h = h * FNV_PRIME          # wraps mod 2**64
            h = h ^ c[:, j]
    return h


def ravel_keys(int_coords):
    c = np.asarray(int_coords).copy()
    c -= c.min(0)
    c = c.astype(np.uint64)
    ext = c.max(0).astype(np.uint64) + np.uint64(1)
    k = np.zeros(c.shape[0], dtype=np.uint64)
    for j in range(c.shape[1] - 1):
        k += c[:, j]
        k *= ext[j + 1]
    k += c[:, -1]
    return k


def quantize_first_occurrence(coords):
    """sparse_quantize(coords, return_index=True): (inds, inverse).

    inds[v]    = index of the first point whose key is the v-th smallest key
    inverse[p] = rank of key(p) among the distinct keys
    """
    key = fnv_keys(np.floor(np.asarray(coords, dtype=np.float64)))
    _, inds, inverse = np.unique(key, return_index=True, return_inverse=True)
    return inds.astype(np.int64), inverse.astype(np.int64).reshape(-1)


def axis_rotation(axis, theta):
    return expm(np.cross(np.eye(3), axis / norm(axis) * theta))


def draw_transform(voxel_size, use_augmentation=True,
                   scale_bound=(0.9, 1.1),
                   rot_bound=((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi)),
                   rng=np.random):
    """Consumes the global numpy RNG exactly like voxelizer.py:46-76.

    Returns the 4x4 float64 matrix R @ V applied by voxelize()."""
    V, R = np.eye(4), np.eye(4)
    rot = np.eye(3)
    if use_augmentation and rot_bound is not None:
        mats = []
        for ax, bound in enumerate(rot_bound):
            theta = 0
            axis = np.zeros(3)
            axis[ax] = 1
            if bound is not None:
                theta = rng.uniform(*bound)
            mats.append(axis_rotation(axis, theta))
        rng.shuffle(mats)
        rot = mats[0] @ mats[1] @ mats[2]
    R[:3, :3] = rot
    scale = 1 / voxel_size
    if use_augmentation and scale_bound is not None:
        scale *= rng.uniform(*scale_bound)
    np.fill_diagonal(V[:3, :3], scale)
    return (R @ V) if use_augmentation else V


def voxelize_with_matrix(xyz, T):
    """voxelizer.py:117-129 for a given 4x4 transform T.

    Returns (voxel_coords float64 [Nv,3], inds int64 [Nv], inverse int64 [Np])."""
    xyz = np.asarray(xyz, dtype=np.float64)
    homo = np.hstack((xyz, np.ones((xyz.shape[0], 1), dtype=xyz.dtype)))
    grid = np.floor(homo @ T.T[:, :3])
    grid = np.floor(grid - grid.min(0))
    inds, inverse = quantize_first_occurrence(grid)
    return grid[inds], inds, inverse


def voxelize(xyz, voxel_size, **kw):
    T = draw_transform(voxel_size, **kw)
    return voxelize_with_matrix(xyz, T) + (T,)
