"""GPU-resident scene -> batch assembly for distillation / evaluation.

The reference does this per scene on DataLoader worker CPUs
(``dataset/feature_loader.py:60-191``, collation ``:193-235``): voxelise, find the fused 2-D
feature row of every kept voxel (the ``index1 / chunk_ind / cumsum`` chain, ``:124-143``),
build ``[1 | xyz]`` coordinates and constant-one (or colour) features, then concatenate the
scenes with the batch index in column 0.  Here the same steps run on tensors that already
live in HBM: ``osn_voxelize_fnv`` (through :class:`Voxelizer`), ``osn_feature_remap`` and
``osn_batch_coords``; the row gathers are ``torch`` indexing (HBM copies).

Same semantics and RNG consumption order as the reference: the voxeliser draws its rotation /
scale from ``numpy.random`` per scene, in scene order; with ``aug`` (the training configuration,
``dataset/point_loader.py:101-113``) :class:`TrainAugmentation` draws from ``random`` /
``numpy.random`` exactly what the reference's transforms draw -- elastic distortion before the
voxeliser, horizontal flip and the chromatic transforms after it -- and applies the arithmetic
to the tensors in HBM (float64, numpy's operation order: bit-identical results).

    scene = FusedScene(xyz, colors, labels, feat, mask_full)     # tensors on the GPU
    item  = fused_feature_item(voxelizer, scene, split="train")
    coords, feats, labels, feat_3d, mask = collate([item0, item1, ...])
"""
import random

import numpy as np
import torch

from . import ops


class TrainAugmentation:
    """The reference's training-time transforms (``dataset/point_loader.py:101-113``, ``dataset/augmentation.py``):
    ``prevoxel`` = ElasticDistortion, ``after_voxelizer`` = RandomHorizontalFlip, ChromaticAutoContrast,
    ChromaticTranslation, ChromaticJitter, HueSaturationTranslation -- same draws from ``random`` and ``numpy.random`` in
    the same order (seed parity with the reference loader), arithmetic on device tensors.

    ElasticDistortion is drawn but, like in the reference, its result does not reach the voxeliser for fused-feature
    training items (``feature_loader.py:122`` computes ``locs``, ``:126`` voxelises ``locs_in``).  Its random field's
    grid size depends on the cloud's extent AFTER the previous distortion, so the first field is evaluated (trilinear
    interpolation on the device) to size the second one; only its bounding box comes back to the host."""

    def __init__(self, elastic_params=((0.2, 0.4), (0.8, 1.6)), color_trans_ratio=0.1, color_jitter_std=0.05, hue_max=0.5,
                 saturation_max=0.2):
        self.elastic_params = elastic_params
        self.color_trans_ratio, self.color_jitter_std = color_trans_ratio, color_jitter_std
        self.hue_max, self.saturation_max = hue_max, saturation_max

    # ---- ElasticDistortion.__call__ (augmentation.py:160-205)
    def prevoxel(self, xyz):
        if self.elastic_params is None:
            return
        if random.random() < 0.95:
            pts = xyz
            for gi, (granularity, magnitude) in enumerate(self.elastic_params):
                last = gi == len(self.elastic_params) - 1
                pts = self._elastic(pts, granularity, magnitude, evaluate=not last)

    @staticmethod
    def _elastic(pts, granularity, magnitude, evaluate):
        from scipy import ndimage
        lo = pts.min(0)[0]
        ext = (pts - lo).max(0)[0]
        lo_h, ext_h = lo.cpu().numpy(), ext.cpu().numpy()                    # the one read-back of a distortion
        noise_dim = (ext_h // granularity).astype(int) + 3
        noise = np.random.randn(*noise_dim, 3).astype(np.float32)
        if not evaluate:
            return None                                                       # the field is drawn; nobody reads the result
        blurx = np.ones((3, 1, 1, 1)).astype("float32") / 3
        blury = np.ones((1, 3, 1, 1)).astype("float32") / 3
        blurz = np.ones((1, 1, 3, 1)).astype("float32") / 3
        for _ in range(2):
            noise = ndimage.convolve(noise, blurx, mode="constant", cval=0)
            noise = ndimage.convolve(noise, blury, mode="constant", cval=0)
            noise = ndimage.convolve(noise, blurz, mode="constant", cval=0)
        # trilinear interpolation on the grid  lo - g + g * i,  i = 0 .. noise_dim - 1  (zero outside)
        field = torch.from_numpy(noise.astype(np.float64)).to(pts.device)
        g0 = torch.from_numpy(lo_h - granularity).to(pts.device)
        t = (pts - g0) / granularity
        nd = torch.tensor(noise_dim, device=pts.device)
        inside = ((t >= 0) & (t <= (nd - 1))).all(1, keepdim=True)
        i0 = torch.minimum(t.floor().clamp(min=0).long(), nd - 2)
        f = t - i0
        out = torch.zeros_like(pts)
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    out += w[:, None] * field[i0[:, 0] + dx, i0[:, 1] + dy, i0[:, 2] + dz]
        return pts + torch.where(inside, out, torch.zeros_like(out)) * magnitude

    # ---- the input transforms, after the voxeliser (augmentation.py:19-139, point_loader.py:105-113)
    def after_voxelizer(self, coords, colors):
        """coords: float64 / int [V, 3] voxel coordinates (device); colors: float64 [V, 3] in 0..255 or None (then only the
        random draws of the chromatic transforms happen).  -> (coords, colors)."""
        n = coords.shape[0]
        # RandomHorizontalFlip('z'): the two horizontal axes, each mirrored about its maximum with probability 1/2
        if random.random() < 0.95:
            for ax in (0, 1):
                if random.random() < 0.5:
                    coords = coords.clone()
                    coords[:, ax] = coords[:, ax].max() - coords[:, ax]
        f = colors
        # ChromaticAutoContrast
        if random.random() < 0.2:
            blend = None
            if f is not None:
                lo, hi = f.min(0, keepdim=True)[0], f.max(0, keepdim=True)[0]
                contrast = (f - lo) * (255 / (hi - lo))
            blend = random.random()
            if f is not None:
                f = (1 - blend) * f + blend * contrast
        # ChromaticTranslation
        if random.random() < 0.95:
            tr = (np.random.rand(1, 3) - 0.5) * 255 * 2 * self.color_trans_ratio
            if f is not None:
                f = (torch.from_numpy(tr).to(f.device) + f).clamp(0, 255)
        # ChromaticJitter
        if random.random() < 0.95:
            noise = np.random.randn(n, 3)
            noise *= self.color_jitter_std * 255
            if f is not None:
                f = (torch.from_numpy(noise).to(f.device) + f).clamp(0, 255)
        # HueSaturationTranslation
        hue_val = (random.random() - 0.5) * 2 * self.hue_max
        sat_ratio = 1 + (random.random() - 0.5) * 2 * self.saturation_max
        if f is not None:
            hsv = self._rgb_to_hsv(f)
            h = torch.remainder(hue_val + hsv[:, 0] + 1, 1)
            sat = (sat_ratio * hsv[:, 1]).clamp(0, 1)
            f = self._hsv_to_rgb(h, sat, hsv[:, 2]).clamp(0, 255)
        return coords, f

    @staticmethod
    def _rgb_to_hsv(rgb):
        r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        maxc, minc = rgb.max(1)[0], rgb.min(1)[0]
        mask = maxc != minc
        one = torch.ones_like(maxc)
        span = torch.where(mask, maxc - minc, one)
        s = torch.where(mask, (maxc - minc) / torch.where(mask, maxc, one), torch.zeros_like(maxc))
        zero = torch.zeros_like(maxc)
        rc = torch.where(mask, (maxc - r) / span, zero)
        gc = torch.where(mask, (maxc - g) / span, zero)
        bc = torch.where(mask, (maxc - b) / span, zero)
        h = torch.where(r == maxc, bc - gc, torch.where(g == maxc, 2.0 + rc - bc, 4.0 + gc - rc))
        h = torch.remainder(h / 6.0, 1.0)
        return torch.stack([h, s, maxc], 1)

    @staticmethod
    def _hsv_to_rgb(h, s, v):
        i = (h * 6.0).to(torch.uint8)                       # numpy's astype('uint8'): truncation
        f = (h * 6.0) - i
        p = v * (1.0 - s)
        q = v * (1.0 - s * f)
        t = v * (1.0 - s * (1.0 - f))
        i = i % 6
        grey = s == 0.0

        def pick(grey_v, c1, c2, c3, c4, c5, default):
            out = default
            for cond, val in ((i == 5, c5), (i == 4, c4), (i == 3, c3), (i == 2, c2), (i == 1, c1), (grey, grey_v)):
                out = torch.where(cond, val, out)       # applied last = highest precedence: numpy.select takes the FIRST true
            return out
        r = pick(v, q, p, p, t, v, v)
        g = pick(v, v, v, q, p, p, t)
        b = pick(v, p, t, v, v, q, p)
        return torch.stack([r, g, b], 1).to(torch.uint8).to(torch.float64)     # .astype('uint8') of the reference


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


def fused_feature_item(voxelizer, scene, split="train", eval_all=False, input_color=False, aug=None):
    """FusedFeatureLoader.__getitem__ for merged-mask feature files (``feature_loader.py:103-191``).
    aug: a :class:`TrainAugmentation` (the reference's ``aug=True``, training split) or None.

    -> (coords3 int32 [V,3], feats f32 [V,3], labels int64, feat_3d [*, D], mask bool [V][, inds_reconstruct int64 [N]])
    train: feat_3d has one row per voxel whose point carries a feature (``mask``), in voxel order;
    val/test: one row per voxel, zeros where there is no feature (``:107-113,166-171``)."""
    if aug is not None:
        if split != "train":
            raise NotImplementedError("the reference augments the training split only (run/distill.py:163-170)")
        aug.prevoxel(scene.xyz)                                  # drawn before the voxeliser's own draws (:122)
    M_v, M_r = voxelizer.get_transformation_matrix()            # consumes numpy.random like the reference
    T = M_r @ M_v if voxelizer.use_augmentation else M_v
    grid, vox_ind, inverse = voxelizer.voxelize_tensors(scene.xyz, T)
    locs = grid[vox_ind]
    colors = None
    if aug is not None:                                          # input transforms on the voxelised cloud (:176-177)
        colors = scene.colors[vox_ind].double() if (input_color and scene.colors is not None) else None
        locs, colors = aug.after_voxelizer(locs, colors)
    coords3 = locs.to(torch.int32)
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
        feats = (colors if colors is not None else scene.colors[vox_ind]).float() / div - 1.0
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
