"""GPU-resident scene -> batch assembly for distillation / evaluation.

The reference does this per scene on DataLoader worker CPUs
(``dataset/feature_loader.py:60-191``, collation ``:193-235``): voxelise, find the fused 2-D
feature row of every kept voxel (the ``index1 / chunk_ind / cumsum`` chain, ``:124-143``),
build ``[1 | xyz]`` coordinates and constant-one (or colour) features, then concatenate the
scenes with the batch index in column 0.  Here the same steps run on tensors that already
live in HBM: ``osn_voxelize_fnv`` (through :class:`Voxelizer`), ``osn_feature_remap`` and
``osn_batch_coords``; the row gathers are ``torch`` indexing (HBM copies).

Same semantics and RNG consumption order as the reference with ``aug=False`` (the
elastic / chromatic augmentations of ``aug=True`` are CPU numpy code outside the hot path):
the voxeliser draws its rotation / scale from ``numpy.random`` per scene, in scene order.

    scene = FusedScene(xyz, colors, labels, feat, mask_full)     # tensors on the GPU
    item  = fused_feature_item(voxelizer, scene, split="train")
    coords, feats, labels, feat_3d, mask = collate([item0, item1, ...])
"""
import torch

from . import ops


class FusedScene:
    """One scene in the reference's on-disk terms, resident on a GPU.

    xyz float64 [N,3]; colors float [N,3] in 0..255 (the loader's ``(c + 1) * 127.5``,
    ``feature_loader.py:79``) or None; labels uint8/int [N] (255 = ignore);
    feat [M, D] compact fused features (one row per True of mask_full, point order, usually
    fp16 as stored by ``scripts/feature_fusion``); mask_full bool [N]."""

    def __init__(self, xyz, colors, labels, feat, mask_full):
        if xyz.dtype != torch.float64:
            raise TypeError("xyz must be float64 (the reference voxelises in float64)")
        n = xyz.shape[0]
        if labels.shape[0] != n or mask_full.shape[0] != n or (colors is not None and colors.shape[0] != n):
            raise ValueError("per-point arrays disagree on the number of points")
        self.xyz, self.colors, self.labels, self.feat, self.mask_full = xyz, colors, labels, feat, mask_full


def fused_feature_item(voxelizer, scene, split="train", eval_all=False, input_color=False):
    """FusedFeatureLoader.__getitem__ for merged-mask feature files (``feature_loader.py:103-191``).

    -> (coords3 int32 [V,3], feats f32 [V,3], labels int64, feat_3d [*, D], mask bool [V][, inds_reconstruct int64 [N]])
    train: feat_3d has one row per voxel whose point carries a feature (``mask``), in voxel order;
    val/test: one row per voxel, zeros where there is no feature (``:107-113,166-171``)."""
    M_v, M_r = voxelizer.get_transformation_matrix()            # consumes numpy.random like the reference
    T = M_r @ M_v if voxelizer.use_augmentation else M_v
    grid, vox_ind, inverse = voxelizer.voxelize_tensors(scene.xyz, T)
    coords3 = grid[vox_ind].to(torch.int32)
    mask_vox, src_row, indices = ops.feature_remap(scene.mask_full, vox_ind)
    if split == "train":
        feat_3d = scene.feat[indices]
    else:
        feat_3d = scene.feat[src_row.clamp(min=0)]
        feat_3d = torch.where(mask_vox.unsqueeze(1), feat_3d, torch.zeros_like(feat_3d))
    n_vox = coords3.shape[0]
    if input_color:
        if scene.colors is None:
            raise ValueError("input_color=True needs per-point colours")
        # fp32 arithmetic after the cast (:181).  Division by a TENSOR: torch's GPU kernels turn division by a
        # Python scalar into a multiplication by its reciprocal, which is not the reference's IEEE quotient.
        div = torch.full((1, 1), 127.5, dtype=torch.float32, device=coords3.device)
        feats = scene.colors[vox_ind].float() / div - 1.0
    else:
        feats = torch.ones(n_vox, 3, device=coords3.device)      # the reference's constant-one input (:183-184)
    labels = (scene.labels if eval_all else scene.labels[vox_ind]).long()
    item = (coords3, feats, labels, feat_3d, mask_vox)
    return item + (inverse,) if eval_all else item


def collate(items):
    """``collation_fn`` / ``collation_fn_eval_all`` (``feature_loader.py:193-235``): column 0 of the
    coordinates is the scene's index in the batch; with a sixth entry per item (``inds_reconstruct``)
    it is offset by the voxels of the scenes before it."""
    if not items:
        raise ValueError("empty batch")
    dev = items[0][0].device
    sizes = [it[0].shape[0] for it in items]
    coords = torch.empty((sum(sizes), 4), dtype=torch.int32, device=dev)
    off = 0
    for b, (it, n) in enumerate(zip(items, sizes)):
        ops.batch_coords(it[0], b, coords[off:off + n])
        off += n
    out = [coords] + [torch.cat([it[j] for it in items]) for j in range(1, 5)]
    if len(items[0]) > 5:
        acc, rec = 0, []
        for it, n in zip(items, sizes):
            rec.append(it[5] + acc)
            acc += n
        out.append(torch.cat(rec))
    return tuple(out)
