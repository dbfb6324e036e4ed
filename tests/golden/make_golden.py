"""Generate golden vectors by running the REFERENCE's own code.

Run in the authoring container only (needs /root/reference):
    python tests/golden/make_golden.py
Imports ``dataset/voxelizer.py`` and ``dataset/voxelization_utils.py`` from
/root/reference behind the two-line ``collections`` alias shim they need on
Python >= 3.10 (they use ``collections.Sequence`` / ``collections.Iterable``),
runs them on seeded inputs and stores inputs + outputs as small .npz fixtures.
The GPU box has no /root/reference; tests only read the fixtures.
"""
import collections
import collections.abc
import os
import sys

import numpy as np

collections.Sequence = collections.abc.Sequence
collections.Iterable = collections.abc.Iterable
sys.path.insert(0, "/root/reference")
from dataset.voxelization_utils import fnv_hash_vec, ravel_hash_vec, sparse_quantize  # noqa: E402
from dataset.voxelizer import Voxelizer  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROT = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))  # point_loader.py:58-60


def hash_kat():
    rng = np.random.default_rng(7)
    c = np.concatenate([np.array([[0, 0, 0], [1, 2, 3], [241, 181, 121]], dtype=np.float64),
                        rng.integers(0, 4096, (4093, 3)).astype(np.float64)])
    np.savez_compressed(os.path.join(HERE, "hash_kat.npz"), coords=c,
                        fnv=fnv_hash_vec(c), ravel=ravel_hash_vec(c))


def quantize_case(name, coords):
    inds, inv = sparse_quantize(coords, return_index=True)
    np.savez_compressed(os.path.join(HERE, name), coords=coords,
                        inds=np.asarray(inds, np.int64), inverse=np.asarray(inv, np.int64).reshape(-1))


def voxelize_case(name, seed, n, voxel_size, extent):
    rng = np.random.default_rng(seed)
    # clustered surface-ish cloud with many duplicates per voxel
    xyz = rng.random((n, 3)) * np.asarray(extent)
    xyz[:, 2] = np.round(xyz[:, 2] * 4) / 4 + rng.normal(0, 0.003, n)
    feats = rng.random((n, 3)) * 255
    labels = rng.integers(0, 20, n).astype(np.uint8)
    vox = Voxelizer(voxel_size=voxel_size, clip_bound=None, use_augmentation=True,
                    scale_augmentation_bound=(0.9, 1.1), rotation_augmentation_bound=ROT,
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)))
    np.random.seed(seed)
    c, f, l, inv, inds = vox.voxelize(xyz, feats, labels, return_ind=True)
    # the transform the voxelizer drew (re-draw with the same seed)
    np.random.seed(seed)
    M_v, M_r = vox.get_transformation_matrix()
    np.savez_compressed(os.path.join(HERE, name), xyz=xyz, np_seed=seed, voxel_size=voxel_size,
                        T=M_r @ M_v, coords=c, inds=np.asarray(inds, np.int64),
                        inverse=np.asarray(inv, np.int64).reshape(-1), feats_sel=f, labels_sel=l)


if __name__ == "__main__":
    hash_kat()
    rng = np.random.default_rng(3)
    quantize_case("quantize_small.npz", rng.integers(0, 12, (500, 3)).astype(np.float64))
    quantize_case("quantize_frac.npz", rng.random((3000, 3)) * 9.0)
    voxelize_case("voxelize_a.npz", 11, 6000, 0.02, (1.2, 0.9, 0.6))
    voxelize_case("voxelize_b.npz", 12, 20000, 0.05, (8.0, 6.0, 2.5))
    print("golden vectors written to", HERE)
