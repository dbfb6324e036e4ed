"""Hash voxeliser on the GPU behind the reference's ``Voxelizer`` interface
(``dataset/voxelizer.py:14-140``; called from ``dataset/point_loader.py:157-158`` and
``dataset/feature_loader.py:126-127,147-148,168-169``).

Host side (this file): constructor arguments, the random rotation / scale draw --
which consumes ``numpy.random``'s global state in exactly the reference's order
(``voxelizer.py:46-76``) so seeded runs agree --, the optional clip, and the final
row gathers.  Device side (``osn_voxelize_fnv``): float64 transform + floor, shift
to the origin, FNV-1 64-bit keys, stable radix sort, first-occurrence selection and
the inverse map -- bit-identical to ``sparse_quantize(..., return_index=True)``
(``dataset/voxelization_utils.py:112-132``).

``voxelize`` takes / returns numpy arrays like the reference; ``voxelize_tensors``
keeps everything in HBM for a GPU-resident loader.
"""
import collections.abc

import numpy as np
import torch
from scipy.linalg import expm, norm

from . import ops


def _axis_rotation(axis, theta):
    # voxelizer.py:11-12
    return expm(np.cross(np.eye(3), axis / norm(axis) * theta))


class Voxelizer:

    def __init__(self, voxel_size=1, clip_bound=None, use_augmentation=False, scale_augmentation_bound=None,
                 rotation_augmentation_bound=None, translation_augmentation_ratio_bound=None, ignore_label=255,
                 device=None):
        self.voxel_size = voxel_size
        self.clip_bound = clip_bound
        self.ignore_label = ignore_label
        self.use_augmentation = use_augmentation
        self.scale_augmentation_bound = scale_augmentation_bound
        self.rotation_augmentation_bound = rotation_augmentation_bound
        self.translation_augmentation_ratio_bound = translation_augmentation_ratio_bound
        self.device = torch.device(device) if device is not None else None

    # -- host: RNG-order-preserving matrix draw (voxelizer.py:46-76) ---------------
    def get_transformation_matrix(self):
        voxelization_matrix, rotation_matrix = np.eye(4), np.eye(4)
        rot = np.eye(3)
        if self.use_augmentation and self.rotation_augmentation_bound is not None:
            if not isinstance(self.rotation_augmentation_bound, collections.abc.Iterable):
                raise ValueError()
            mats = []
            for axis_ind, bound in enumerate(self.rotation_augmentation_bound):
                axis = np.zeros(3)
                axis[axis_ind] = 1
                theta = np.random.uniform(*bound) if bound is not None else 0
                mats.append(_axis_rotation(axis, theta))
            np.random.shuffle(mats)
            rot = mats[0] @ mats[1] @ mats[2]
        rotation_matrix[:3, :3] = rot
        scale = 1 / self.voxel_size
        if self.use_augmentation and self.scale_augmentation_bound is not None:
            scale *= np.random.uniform(*self.scale_augmentation_bound)
        np.fill_diagonal(voxelization_matrix[:3, :3], scale)
        return voxelization_matrix, rotation_matrix

    def clip(self, coords, center=None, trans_aug_ratio=None):
        """Boolean mask of the points inside ``clip_bound`` (per axis a half-open interval [lo, hi) relative to ``center``, by
        default the middle of the cloud's bounding box, optionally shifted by a fraction of the box) -- the result of
        dataset/voxelizer.py:78-95.  Unused by every shipped config (``clip_bound=None``, point_loader.py:95)."""
        lo, hi = np.min(coords, 0).astype(float), np.max(coords, 0).astype(float)
        mid = lo + (hi - lo) * 0.5 if center is None else np.asarray(center, dtype=float)
        if trans_aug_ratio is not None:
            mid = mid + np.multiply(trans_aug_ratio, hi - lo)
        bound = np.asarray(self.clip_bound, dtype=float)              # [3, 2]
        xyz = coords[:, :3]
        return np.all((xyz >= bound[:, 0] + mid) & (xyz < bound[:, 1] + mid), axis=1)

    def _device(self):
        if self.device is not None:
            return self.device
        if not torch.cuda.is_available():
            raise RuntimeError("openscene_amd.Voxelizer needs a HIP device (there is no CPU path)")
        return torch.device("cuda", torch.cuda.current_device())

    # -- device ----------------------------------------------------------------------
    def voxelize_tensors(self, xyz, transform):
        """xyz float64 [N,3] device tensor, transform 4x4 (host) ->
        (grid float64 [N,3] integral & origin-aligned, inds int64 [Nv], inverse int64 [N])."""
        return ops.voxelize_fnv(xyz, transform)

    def voxelize(self, coords, feats, labels, center=None, link=None, return_ind=False):
        assert coords.shape[1] == 3 and coords.shape[0] == feats.shape[0] and coords.shape[0]
        if self.clip_bound is not None:
            trans_aug_ratio = np.zeros(3)
            if self.use_augmentation and self.translation_augmentation_ratio_bound is not None:
                for axis_ind, bound in enumerate(self.translation_augmentation_ratio_bound):
                    trans_aug_ratio[axis_ind] = np.random.uniform(*bound)
            keep = self.clip(coords, center, trans_aug_ratio)
            if keep.sum():
                coords, feats = coords[keep], feats[keep]
                if labels is not None:
                    labels = labels[keep]

        M_v, M_r = self.get_transformation_matrix()
        T = M_r @ M_v if self.use_augmentation else M_v

        dev = self._device()
        xyz = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float64)).to(dev)
        grid, inds_t, inverse_t = self.voxelize_tensors(xyz, T)
        inds = inds_t.cpu().numpy()
        inds_reconstruct = inverse_t.cpu().numpy()
        coords_aug = grid[inds_t].cpu().numpy()
        feats, labels = feats[inds], labels[inds]

        if feats.shape[1] > 6:                      # normals ride along in columns 3:6
            feats[:, 3:6] = feats[:, 3:6] @ (M_r[:3, :3].T)

        if return_ind:
            return coords_aug, feats, labels, np.array(inds_reconstruct), inds
        if link is not None:
            return coords_aug, feats, labels, np.array(inds_reconstruct), link[inds]
        return coords_aug, feats, labels, np.array(inds_reconstruct)
