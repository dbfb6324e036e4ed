"""Shared body of the loader tests: run openscene_amd.loader on the scenes of tests/golden/loader_fused.npz
(inputs + outputs of the reference's REAL FusedFeatureLoader) and compare bit for bit."""
import os

import numpy as np
import torch


def load(golden_dir):
    return np.load(os.path.join(golden_dir, "loader_fused.npz"))


def run(d, device, split, eval_all, input_color):
    from openscene_amd.loader import FusedScene, collate, fused_feature_item
    from openscene_amd.voxelizer import Voxelizer
    rot = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))       # point_loader.py:58-60
    vox = Voxelizer(voxel_size=0.05, clip_bound=None, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=rot,
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)), device=device)
    np.random.seed(int(d["%s_seed" % split]))
    items = []
    for k in range(2):
        labels = d["s%d_labels" % k].copy()
        labels[labels == -100] = 255                                # feature_loader.py:72-73
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        scene = FusedScene(t(d["s%d_xyz" % k]), t((d["s%d_colors" % k] + 1.0) * 127.5), t(labels.astype(np.uint8)),
                           t(d["s%d_feat" % k]), t(d["s%d_mask_full" % k]))
        items.append(fused_feature_item(vox, scene, split=split, eval_all=eval_all, input_color=input_color))
    return collate(items)


def check(d, got, split, eval_all):
    names = ["coords", "feats", "labels", "feat_3d", "mask"] + (["inds_recons"] if eval_all else [])
    assert len(got) == len(names)
    for nm, g in zip(names, got):
        want = d["%s_%s" % (split, nm)]
        g = g.cpu().numpy()
        assert g.shape == want.shape and g.dtype == want.dtype, (nm, g.shape, g.dtype, want.shape, want.dtype)
        assert np.array_equal(g, want), nm


def load_aug(golden_dir):
    return np.load(os.path.join(golden_dir, "loader_fused_aug.npz"))


def run_aug(d, device, tag, input_color):
    """The training configuration (aug=True) on the golden scenes, seeded like the reference run."""
    import random
    from openscene_amd.loader import FusedScene, TrainAugmentation, collate, fused_feature_item
    from openscene_amd.voxelizer import Voxelizer
    rot = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))
    vox = Voxelizer(voxel_size=0.05, clip_bound=None, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=rot,
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)), device=device)
    seed = int(d["%s_seed" % tag])
    np.random.seed(seed)
    random.seed(seed)
    aug = TrainAugmentation()
    items = []
    for k in range(2):
        labels = d["s%d_labels" % k].copy()
        labels[labels == -100] = 255
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        scene = FusedScene(t(d["s%d_xyz" % k]), t((d["s%d_colors" % k] + 1.0) * 127.5), t(labels.astype(np.uint8)),
                           t(d["s%d_feat" % k]), t(d["s%d_mask_full" % k]))
        items.append(fused_feature_item(vox, scene, split="train", input_color=input_color, aug=aug))
    return collate(items)


def check_aug(d, got, tag):
    for nm, g in zip(["coords", "feats", "labels", "feat_3d", "mask"], got):
        want = d["%s_%s" % (tag, nm)]
        g = g.cpu().numpy()
        assert g.shape == want.shape and g.dtype == want.dtype, (tag, nm, g.shape, g.dtype, want.shape, want.dtype)
        assert np.array_equal(g, want), (tag, nm, int((g != want).sum()))
