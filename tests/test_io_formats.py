"""openscene_amd.io reads / writes the reference's on-disk formats.  The scene and
feature arrays of tests/golden/loader_fused.npz are the ones tests/golden/make_golden.py wrote to disk for
the reference's REAL loader, so feeding files written here through openscene_amd.io + openscene_amd.loader
must reproduce that loader's outputs."""
import os

import numpy as np
import pytest
import torch

import cpu_backend
import loader_cases
from openscene_amd import io as oio


def test_scene_and_feature_files_round_trip_into_the_loader(tmp_path, golden_dir, monkeypatch):
    cpu_backend.install(monkeypatch)
    d = loader_cases.load(golden_dir)
    from openscene_amd.loader import collate, fused_feature_item
    from openscene_amd.voxelizer import Voxelizer
    rot = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))
    vox = Voxelizer(voxel_size=0.05, clip_bound=None, use_augmentation=True, scale_augmentation_bound=(0.9, 1.1),
                    rotation_augmentation_bound=rot, device="cpu")
    scenes = []
    for k in range(2):
        sp, fp = tmp_path / ("scene%04d_00_vh_clean_2.pth" % k), tmp_path / ("scene%04d_00_0.pt" % k)
        oio.save_scene(sp, d["s%d_xyz" % k], d["s%d_colors" % k], d["s%d_labels" % k])
        oio.save_fused_features(fp, torch.from_numpy(d["s%d_feat" % k]), torch.from_numpy(d["s%d_mask_full" % k]))
        xyz, colors, labels = oio.load_scene(sp)
        assert labels.dtype == np.uint8 and (labels == 255).sum() == (d["s%d_labels" % k] == -100).sum()
        assert np.array_equal(colors, (d["s%d_colors" % k] + 1.0) * 127.5)
        scenes.append(oio.scene_to_device(sp, fp, torch.device("cpu")))
    np.random.seed(int(d["train_seed"]))
    got = collate([fused_feature_item(vox, s, split="train") for s in scenes])
    loader_cases.check(d, got, "train", False)


def test_legacy_three_key_feature_file_and_lidar_scene(tmp_path):
    n = 50
    mask_full = torch.zeros(n, dtype=torch.bool)
    mask_full[::2] = True                                   # 25 candidate points
    feat_all = torch.arange(25 * 4, dtype=torch.float32).reshape(25, 4, 1)
    visible = torch.tensor([0, 3, 7, 24])
    torch.save({"feat": feat_all, "mask": visible, "mask_full": mask_full.numpy()}, tmp_path / "legacy.pt")
    feat, m = oio.load_fused_features(tmp_path / "legacy.pt")
    assert feat.shape == (4, 4) and torch.equal(feat, feat_all[visible, :, 0])
    assert int(m.sum()) == 4 and torch.equal(m.nonzero().squeeze(1), torch.tensor([0, 6, 14, 48]))
    oio.save_scene(tmp_path / "lidar.pth", np.zeros((5, 3)), 0, np.array([1, -100, 3, 4, -100]))
    xyz, colors, labels = oio.load_scene(tmp_path / "lidar.pth")
    assert colors.shape == (5, 3) and not colors.any() and labels.tolist() == [1, 255, 3, 4, 255]


def test_checkpoints_interchange_with_the_reference_naming(tmp_path, monkeypatch):
    cpu_backend.install(monkeypatch)
    from openscene_amd.mink_unet import mink_unet
    torch.manual_seed(0)
    a, b = mink_unet(3, 8, 3, "MinkUNet14A"), mink_unet(3, 8, 3, "MinkUNet14A")
    opt = torch.optim.Adam(a.parameters(), lr=1e-3)
    with pytest.raises(KeyError):
        oio.save_checkpoint({"epoch": 1, "state_dict": a.state_dict()}, False, tmp_path)
    # a checkpoint written from a DistributedDataParallel-wrapped model carries the `module.` prefix
    sd = {"module." + k: v for k, v in a.state_dict().items()}
    p = oio.save_checkpoint({"epoch": 7, "state_dict": sd, "optimizer": opt.state_dict(), "best_iou": 0.5}, True,
                            tmp_path / "model")
    assert os.path.exists(tmp_path / "model" / "model_best.pth.tar")
    epoch, best = oio.load_checkpoint(p, b, torch.optim.Adam(b.parameters(), lr=1e-3))
    assert (epoch, best) == (7, 0.5)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    assert "conv0p1s1.kernel" in a.state_dict() and "block2.0.downsample.1.bn.running_var" in a.state_dict()
