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


def loader_cases():
    """Drive the reference's REAL FusedFeatureLoader (dataset/feature_loader.py) on two synthetic scenes
    written in its on-disk formats, train split and val split (eval_all), and store every input and
    output.  SharedArray (shared-memory cache, unused with memcache_init=False) is stubbed."""
    import shutil
    import tempfile
    import types
    import torch
    sys.modules.setdefault("SharedArray", types.ModuleType("SharedArray"))
    from dataset import feature_loader as fl
    # the reference targets torch 1.x, where torch.load unpickled numpy tuples by default
    real_load = torch.load
    torch.load = lambda *a, **k: real_load(*a, **dict(k, weights_only=False))

    root = tempfile.mkdtemp(prefix="osn_golden_")
    try:
        data = os.path.join(root, "scannet_3d")
        featdir = os.path.join(root, "feat")
        rng = np.random.default_rng(21)
        scenes = {}
        for split in ("train", "val"):
            os.makedirs(os.path.join(data, split))
        os.makedirs(featdir)
        D = 16
        for si, n in enumerate((5000, 3500)):
            xyz = rng.random((n, 3)) * np.asarray((2.0, 1.5, 0.8))
            xyz[:, 2] = np.round(xyz[:, 2] * 3) / 3 + rng.normal(0, 0.004, n)
            colors = rng.random((n, 3)).astype(np.float64) * 2 - 1          # [-1, 1] on disk
            labels = rng.integers(0, 20, n).astype(np.float64)
            labels[rng.random(n) < 0.05] = -100
            mask_full = rng.random(n) < 0.6
            feat = rng.standard_normal((int(mask_full.sum()), D)).astype(np.float16)
            name = "scene%04d_00" % si
            for split in ("train", "val"):
                torch.save((xyz, colors, labels.copy()), os.path.join(data, split, name + "_vh_clean_2.pth"))
            torch.save({"feat": torch.from_numpy(feat), "mask_full": torch.from_numpy(mask_full)},
                       os.path.join(featdir, name + "_0.pt"))
            scenes[name] = dict(xyz=xyz, colors=colors, labels=labels, mask_full=mask_full, feat=feat)
        out = {}
        for k, (name, sc) in enumerate(sorted(scenes.items())):
            for key, v in sc.items():
                out["s%d_%s" % (k, key)] = v
        for split, eval_all, input_color, seed in (("train", False, False, 5), ("val", True, True, 6)):
            ds = fl.FusedFeatureLoader(datapath_prefix=data, datapath_prefix_feat=featdir, voxel_size=0.05,
                                       split=split, aug=False, memcache_init=False, eval_all=eval_all,
                                       input_color=input_color)
            np.random.seed(seed)
            items = [ds[i] for i in range(len(ds))]
            batch = (fl.collation_fn_eval_all if eval_all else fl.collation_fn)(items)
            names = ("coords", "feats", "labels", "feat_3d", "mask", "inds_recons")
            for nm, t in zip(names, batch):
                out["%s_%s" % (split, nm)] = t.numpy()
            out["%s_seed" % split] = seed
        np.savez_compressed(os.path.join(HERE, "loader_fused.npz"), **out)
        # ---- the TRAINING configuration: aug=True (config/scannet/ours_openseg.yaml `aug: True`): elastic distortion is
        # drawn (and, for merged-mask files, not used: feature_loader.py:122 vs :126), then after the voxeliser the
        # horizontal flip and -- with colour input -- the chromatic transforms (dataset/point_loader.py:101-113).
        import random
        aug = {k: v for k, v in out.items() if k.startswith("s")}
        for tag, input_color, seed in (("ones", False, 7), ("color", True, 8), ("color2", True, 9)):
            ds = fl.FusedFeatureLoader(datapath_prefix=data, datapath_prefix_feat=featdir, voxel_size=0.05,
                                       split="train", aug=True, memcache_init=False, eval_all=False, input_color=input_color)
            np.random.seed(seed)
            random.seed(seed)
            items = [ds[i] for i in range(len(ds))]
            batch = fl.collation_fn(items)
            for nm, t in zip(("coords", "feats", "labels", "feat_3d", "mask"), batch):
                aug["%s_%s" % (tag, nm)] = t.numpy()
            aug["%s_seed" % tag] = seed
        np.savez_compressed(os.path.join(HERE, "loader_fused_aug.npz"), **aug)
    finally:
        torch.load = real_load
        shutil.rmtree(root, ignore_errors=True)


def fusion_cases():
    """PointCloudToImageMapper / make_intrinsic / adjust_intrinsic of scripts/feature_fusion/fusion_util.py, executed
    from the reference file itself (the module imports TensorFlow at the top, so only these three definitions are
    compiled out of its syntax tree -- nothing is copied).  Three views of a room-shaped cloud: ScanNet intrinsics
    (scannet_openseg.py:119-125) resized 640x480 -> 320x240, cut_bound 10, visibility 0.25, a z-buffer depth image
    with noise and holes (so that both outcomes of the occlusion test occur) and one view without depth."""
    import ast
    import math
    src = open("/root/reference/scripts/feature_fusion/fusion_util.py").read()
    want = ("PointCloudToImageMapper", "make_intrinsic", "adjust_intrinsic")
    body = [n for n in ast.parse(src).body if getattr(n, "name", None) in want]
    ns = {"np": np, "math": math}
    exec(compile(ast.Module(body=body, type_ignores=[]), "fusion_util.py (subset)", "exec"), ns)
    rng = np.random.default_rng(21)
    n = 24000
    # surface-ish room cloud 6 x 4.5 x 2.6 m: points on the floor, the walls and a few boxes, float32 like the scene files
    face = rng.integers(0, 6, n)
    pts = rng.random((n, 3)) * np.array([6.0, 4.5, 2.6])
    pts[face == 0, 2] = 0.0
    pts[face == 1, 0] = 0.0
    pts[face == 2, 0] = 6.0
    pts[face == 3, 1] = 0.0
    pts[face == 4, 1] = 4.5
    box = face == 5
    pts[box] = np.array([2.0, 1.5, 0.0]) + rng.random((int(box.sum()), 3)) * np.array([1.2, 0.8, 0.9])
    coords = pts.astype(np.float32).astype(np.float64)
    intr = ns["make_intrinsic"](577.870605, 577.870605, 319.5, 239.5)
    intr = ns["adjust_intrinsic"](intr, [640, 480], (320, 240))
    mapper = ns["PointCloudToImageMapper"](image_dim=(320, 240), visibility_threshold=0.25, cut_bound=10, intrinsics=intr)

    def pose(eye, yaw, pitch):
        cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
        fwd = np.array([cy * cp, sy * cp, sp])
        right = np.cross(fwd, [0.0, 0.0, 1.0])
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, fwd, eye      # camera axes x right, y down, z forward
        return m

    out = {"coords": coords, "intrinsic": intr, "image_dim": np.array([320, 240]), "cut_bound": 10, "vis_thres": 0.25}
    views = [((3.0, 2.2, 1.4), 0.4, -0.2), ((1.0, 1.0, 1.6), 2.3, -0.35), ((5.0, 3.5, 1.2), -2.6, 0.05)]
    for v, (eye, yaw, pitch) in enumerate(views):
        c2w = pose(np.array(eye), yaw, pitch)
        # z-buffer of the cloud itself, then noise and holes
        w2c = np.linalg.inv(c2w)
        pc = w2c @ np.concatenate([coords, np.ones((n, 1))], 1).T
        u = np.round(pc[0] * intr[0][0] / pc[2] + intr[0][2]).astype(int)
        w = np.round(pc[1] * intr[1][1] / pc[2] + intr[1][2]).astype(int)
        ok = (pc[2] > 0.1) & (u >= 0) & (u < 320) & (w >= 0) & (w < 240)
        depth = np.full((240, 320), np.inf)
        np.minimum.at(depth, (w[ok], u[ok]), pc[2][ok])
        depth[np.isinf(depth)] = 0.0
        depth *= 1.0 + rng.normal(0, 0.08, depth.shape)                       # pushes some points across the threshold
        depth[rng.random(depth.shape) < 0.05] = 0.0                           # invalid depth pixels
        depth = np.round(depth * 1000.0) / 1000.0                             # uint16 millimetres / depth_scale
        out["pose%d" % v] = c2w
        out["depth%d" % v] = depth
        out["mapping%d" % v] = mapper.compute_mapping(c2w, coords, depth)
    out["mapping_nodepth"] = mapper.compute_mapping(out["pose0"], coords, None)
    # the un-adjusted intrinsics too (make_intrinsic alone) and an adjust that returns its input
    out["intrinsic_raw"] = ns["make_intrinsic"](577.870605, 577.870605, 319.5, 239.5)
    np.savez_compressed(os.path.join(HERE, "fusion_mapping.npz"), **out)
    print("fusion: visible per view", [int(out["mapping%d" % v][:, 2].sum()) for v in range(3)],
          "no depth", int(out["mapping_nodepth"][:, 2].sum()))


if __name__ == "__main__":
    hash_kat()
    rng = np.random.default_rng(3)
    quantize_case("quantize_small.npz", rng.integers(0, 12, (500, 3)).astype(np.float64))
    quantize_case("quantize_frac.npz", rng.random((3000, 3)) * 9.0)
    voxelize_case("voxelize_a.npz", 11, 6000, 0.02, (1.2, 0.9, 0.6))
    voxelize_case("voxelize_b.npz", 12, 20000, 0.05, (8.0, 6.0, 2.5))
    loader_cases()
    fusion_cases()
    print("golden vectors written to", HERE)
