"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's multi-view feature fusion.

* ``compute_mapping``: scripts/feature_fusion/fusion_util.py:93-139 (``PointCloudToImageMapper.compute_mapping``):
  world -> camera, pinhole projection, rounding to the pixel, image-boundary test, depth-occlusion test.
  PINNED: tests/golden/fusion_mapping.npz holds the outputs of the reference's own class (executed from
  /root/reference by tests/golden/make_golden.py) on seeded views; tests/test_oracle_fusion.py compares bit for bit.
* ``make_intrinsic`` / ``adjust_intrinsic``: fusion_util.py:17-39 (pinned by the same fixture).
* ``accumulate`` / ``finish``: the running mean of scripts/feature_fusion/scannet_openseg.py:75-111
  (``counter[mask != 0] += 1; sum_features[mask != 0] += feat_2d[:, y, x].T`` per view, then
  ``counter[counter == 0] = 1e-5; feat_bank = sum_features / counter``).  Parity unpinned: the reference function needs
  the TensorFlow OpenSeg model to run; these five lines are restated with the same torch operations.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import math

import numpy as np
import torch


def make_intrinsic(fx, fy, mx, my):
    intrinsic = np.eye(4)
    intrinsic[0][0] = fx
    intrinsic[1][1] = fy
    intrinsic[0][2] = mx
    intrinsic[1][2] = my
    return intrinsic


def adjust_intrinsic(intrinsic, intrinsic_image_dim, image_dim):
    if tuple(intrinsic_image_dim) == tuple(image_dim):
        return intrinsic
    resize_width = int(math.floor(image_dim[1] * float(intrinsic_image_dim[0]) / float(intrinsic_image_dim[1])))
    intrinsic = intrinsic.copy()
    intrinsic[0, 0] *= float(resize_width) / float(intrinsic_image_dim[0])
    intrinsic[1, 1] *= float(image_dim[1]) / float(intrinsic_image_dim[1])
    intrinsic[0, 2] *= float(image_dim[0] - 1) / float(intrinsic_image_dim[0] - 1)
    intrinsic[1, 2] *= float(image_dim[1] - 1) / float(intrinsic_image_dim[1] - 1)
    return intrinsic


def compute_mapping(camera_to_world, coords, depth, intrinsic, image_dim, vis_thres=0.25, cut_bound=0, world_to_camera=None):
    """-> int64 [N, 3] rows (pixel row, pixel column, visible) exactly as fusion_util.py:103-139.
    world_to_camera: the already inverted pose (tests of the product's host code pass the matrix the product computed)."""
    n = coords.shape[0]
    mapping = np.zeros((3, n), dtype=np.int64)
    coords_new = np.concatenate([coords, np.ones([n, 1])], axis=1).T
    if world_to_camera is None:
        world_to_camera = np.linalg.inv(camera_to_world)
    p = np.matmul(world_to_camera, coords_new)
    with np.errstate(divide="ignore", invalid="ignore"):
        p[0] = (p[0] * intrinsic[0][0]) / p[2] + intrinsic[0][2]
        p[1] = (p[1] * intrinsic[1][1]) / p[2] + intrinsic[1][2]
        pi = np.round(p).astype(np.int64)
    inside = (pi[0] >= cut_bound) & (pi[1] >= cut_bound) & (pi[0] < image_dim[0] - cut_bound) & (pi[1] < image_dim[1] - cut_bound)
    if depth is not None:
        d = depth[pi[1][inside], pi[0][inside]]
        occ = np.abs(d - p[2][inside]) <= vis_thres * d
        inside[inside] = occ
    else:
        inside = (p[2] > 0) & inside
    mapping[0][inside] = pi[1][inside]
    mapping[1][inside] = pi[0][inside]
    mapping[2][inside] = 1
    return mapping.T


def accumulate(sum_features, counter, feat_2d, mapping):
    """One view of scannet_openseg.py:93-106.  sum_features [N, D] fp32, counter [N, 1] fp32 (updated in place),
    feat_2d [D, H, W] fp32, mapping int64 [N, 3]."""
    mapping = torch.as_tensor(mapping)
    mask = mapping[:, 2]
    feat_2d_3d = feat_2d[:, mapping[:, 0], mapping[:, 1]].permute(1, 0)
    counter[mask != 0] += 1
    sum_features[mask != 0] += feat_2d_3d[mask != 0]


def finish(sum_features, counter):
    """scannet_openseg.py:108-109 -> feat_bank [N, D] fp32."""
    counter = counter.clone()
    counter[counter == 0] = 1e-5
    return sum_features / counter
