"""numpy restatement of the reference's fused-feature re-indexing and batch assembly
(test oracle, CPU only -- never imported by the product).

Follows, in behaviour (not in text):
  * ``dataset/feature_loader.py:103-113``  merged-mask files; val/test: zero-filled feature
    matrix ``feat_new[mask] = feat`` and ``mask_chunk = ones``
  * ``dataset/feature_loader.py:124-143``  train: ``mask = mask_chunk[vox_ind]`` and the
    ``index1 / chunk_ind / cumsum`` chain that finds, for every kept voxel, the row of the
    compact feature matrix that belongs to its first point
  * ``dataset/feature_loader.py:166-171``  val/test: ``feat[vox_ind]``, ``mask[vox_ind]``
  * ``dataset/feature_loader.py:177-191``  coords ``[1 | xyz]`` int32, constant-one or
    colour features, int64 labels
  * ``dataset/feature_loader.py:193-209`` / ``dataset/point_loader.py:36-52``  collation:
    batch column ``*= i``, concatenation, ``inds_reconstruct`` offset by the voxels before it
Pinned against the reference's own FusedFeatureLoader by tests/golden/make_golden.py
(fixtures tests/golden/loader_*.npz).
"""
import numpy as np


def remap(mask_chunk, vox_ind):
    """-> (mask_vox bool [V], src_row int64 [V], indices int64 [n_sel]).

    mask_chunk[p]  : point p has a fused 2-D feature; the compact feature matrix has one row per
                     True entry, in point order  =>  row of point p = (#True before p).
    vox_ind[v]     : the point that represents voxel v (first occurrence, voxeliser output).
    mask_vox[v]    = mask_chunk[vox_ind[v]]
    src_row[v]     = feature row of voxel v's point, or -1 when it has none
    indices        = src_row[mask_vox]   (what feature_loader.py:124-140 calls `indices`)
    """
    mask_chunk = np.asarray(mask_chunk, dtype=bool)
    vox_ind = np.asarray(vox_ind, dtype=np.int64)
    rank = np.cumsum(mask_chunk.astype(np.int64)) - 1            # index3 - 1
    mask_vox = mask_chunk[vox_ind]
    src_row = np.where(mask_vox, rank[vox_ind], -1).astype(np.int64)
    return mask_vox, src_row, src_row[mask_vox]


def item(split, coords3, vox_ind, labels_in, colors_in, feat, mask_chunk, eval_all=False, input_color=False):
    """One scene after voxelisation -> (coords [V,4] int32, feats [V,3] f32, labels int64, feat_3d, mask).

    coords3 int [V,3] voxel coordinates, vox_ind int64 [V] (both from the voxeliser);
    labels_in [N] uint8, colors_in [N,3] in 0..255, feat [M,D] compact features, mask_chunk bool [N]."""
    mask_chunk = np.asarray(mask_chunk, dtype=bool)
    mask_vox, src_row, indices = remap(mask_chunk, vox_ind)
    if split == "train":
        feat_3d = feat[indices]
        mask = mask_vox
    else:
        # zero rows for points without a feature, every point evaluated
        full = np.zeros((mask_chunk.shape[0], feat.shape[1]), dtype=feat.dtype)
        full[mask_chunk] = feat
        feat_3d = full[vox_ind]
        mask = mask_vox
    coords = np.concatenate([np.ones((coords3.shape[0], 1), np.int32), np.asarray(coords3).astype(np.int32)], 1)
    if input_color:
        # torch.from_numpy(feats).float() / 127.5 - 1.  : cast to fp32 FIRST, then fp32 arithmetic (:181)
        feats = np.asarray(colors_in)[vox_ind].astype(np.float32) / np.float32(127.5) - np.float32(1.0)
    else:
        feats = np.ones((coords.shape[0], 3), np.float32)
    labels = np.asarray(labels_in if eval_all else np.asarray(labels_in)[vox_ind]).astype(np.int64)
    return coords, feats, labels, feat_3d, mask


def collate(items, inds_reconstruct=None):
    """collation_fn / collation_fn_eval_all: batch column = scene index (the loader wrote 1, `*= i`)."""
    coords = []
    for i, it in enumerate(items):
        c = it[0].copy()
        c[:, 0] *= i
        coords.append(c)
    out = [np.concatenate(coords)] + [np.concatenate([it[j] for it in items]) for j in range(1, len(items[0]))]
    if inds_reconstruct is not None:
        acc, rec = 0, []
        for it, r in zip(items, inds_reconstruct):
            rec.append(np.asarray(r, np.int64) + acc)
            acc += it[0].shape[0]
        out.append(np.concatenate(rec))
    return tuple(out)
