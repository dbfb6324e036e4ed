"""The reference's on-disk formats -- pure I/O, no kernels.

* scene file ``<scene>_vh_clean_2.pth`` / ``<scene>.pth``: ``torch.save((coords f64 [N,3], colors [N,3]
  in [-1, 1] or the scalar 0, labels [N] with -100 = ignore))``
  (written by ``scripts/preprocess/preprocess_3d_scannet.py:24-25``; read at
  ``dataset/feature_loader.py:70-79``, ``dataset/point_loader.py:132-139``);
* fused-feature file ``<scene>_<k>.pt``: ``{"feat": fp16 [M, D], "mask_full": bool [N]}`` (one row per True,
  point order), or the legacy three-key form ``{"feat", "mask", "mask_full"}``
  (``scripts/feature_fusion/fusion_util.py:87``; read at ``dataset/feature_loader.py:100-118``);
* checkpoint ``model_last.pth.tar`` / ``model_best.pth.tar``:
  ``{"epoch", "state_dict", "optimizer", "best_iou"}`` (``run/distill.py:235-242``, ``util/util.py:18-22``;
  read at ``run/distill.py:158-163``, ``run/evaluate.py:174-190`` incl. the ``module.`` prefix fallback).

Our modules keep MinkowskiEngine's parameter names, so these files are interchangeable with the
reference's in both directions.
"""
import os
import shutil

import numpy as np
import torch


def _load(path, map_location="cpu"):
    # the reference's files hold numpy arrays / plain dicts: they need the full unpickler
    return torch.load(path, map_location=map_location, weights_only=False)


def load_scene(path):
    """-> (xyz float64 [N,3], colors float64 [N,3] in 0..255, labels uint8 [N]) exactly as the reference's
    loaders hold them after reading (``feature_loader.py:70-79``): -100 -> 255, colours -> (c + 1) * 127.5,
    a scalar-0 colour entry (lidar clouds) -> zeros."""
    locs, feats, labels = _load(path)
    locs = np.asarray(locs)
    labels = np.asarray(labels).copy()
    labels[labels == -100] = 255
    labels = labels.astype(np.uint8)
    if np.isscalar(feats) and feats == 0:
        feats = np.zeros_like(locs)
    else:
        feats = (np.asarray(feats) + 1.0) * 127.5
    return locs, feats, labels


def save_scene(path, xyz, colors_pm1, labels):
    """Write a scene file in the reference's format (colours in [-1, 1]; labels with -100 = ignore)."""
    torch.save((np.asarray(xyz), colors_pm1 if np.isscalar(colors_pm1) else np.asarray(colors_pm1),
                np.asarray(labels)), path)


def load_fused_features(path):
    """-> (feat tensor [M, D], mask_full bool tensor [N]) with one feature row per True of mask_full.
    The legacy three-key files carry features for every point of ``mask_full`` plus a visibility index
    ``mask``; they are reduced to the same compact form (``feature_loader.py:114-118,144-150``)."""
    d = _load(path)
    feat, mask_full = d["feat"], d["mask_full"]
    if isinstance(mask_full, np.ndarray):
        mask_full = torch.from_numpy(mask_full)
    if isinstance(feat, np.ndarray):
        feat = torch.from_numpy(feat)
    mask_full = mask_full.bool()
    if len(d.keys()) > 2:                                   # legacy: feat rows for all of mask_full, `mask` = visible ones
        visible = torch.zeros(feat.shape[0], dtype=torch.bool)
        visible[d["mask"]] = True
        feat = feat[visible]
        mask_full = mask_full.clone()
        mask_full[mask_full.clone()] = visible
    if feat.dim() > 2:
        feat = feat[..., 0]
    return feat, mask_full


def save_fused_features(path, feat, mask_full):
    torch.save({"feat": feat, "mask_full": mask_full}, path)


def save_checkpoint(state, is_best, save_dir, filename="model_last.pth.tar"):
    """``util/util.py:18-22``: ``state = {"epoch", "state_dict", "optimizer", "best_iou"}``."""
    missing = {"epoch", "state_dict", "optimizer", "best_iou"} - set(state)
    if missing:
        raise KeyError("checkpoint state lacks %s" % sorted(missing))
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, filename)
    torch.save(state, path)
    if is_best:
        shutil.copyfile(path, os.path.join(save_dir, "model_best.pth.tar"))
    return path


def load_checkpoint(path, model, optimizer=None, map_location="cpu"):
    """``run/evaluate.py:174-190`` / ``run/distill.py:158-163``: strict load; if the names disagree by the
    ``module.`` prefix of DistributedDataParallel, add / strip it and load strictly again.
    -> (epoch, best_iou or None)."""
    ck = _load(path, map_location)
    sd = ck["state_dict"]
    try:
        model.load_state_dict(sd, strict=True)
    except RuntimeError:
        sd = {(k[7:] if k.startswith("module.") else "module." + k): v for k, v in sd.items()}
        model.load_state_dict(sd, strict=True)
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return ck.get("epoch"), ck.get("best_iou")


def scene_to_device(scene_path, feature_path, device):
    """Both files of one training scene as a loader.FusedScene resident on `device`."""
    from .loader import FusedScene
    xyz, colors, labels = load_scene(scene_path)
    feat, mask_full = load_fused_features(feature_path)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return FusedScene(t(xyz.astype(np.float64)), t(colors), t(labels), feat.to(device), mask_full.to(device))
