"""Multi-view feature fusion on the GPU: the reference's
``scripts/feature_fusion/fusion_util.py:93-139`` ``PointCloudToImageMapper`` (same constructor, same
``compute_mapping`` signature and result, as a device tensor) and the per-scene running mean of
``scripts/feature_fusion/scannet_openseg.py:75-111``.  The 2-D feature extractor (TensorFlow OpenSeg /
LSeg) is outside this library: ``FeatureFusion.add_view`` takes its per-pixel output ``[D, H, W]``."""
import math

import numpy as np
import torch

from . import ops


def make_intrinsic(fx, fy, mx, my):
    """4 x 4 pinhole matrix (focal lengths on the diagonal, principal point in column 2): what fusion_util.py:17-25 returns."""
    K = np.diag([float(fx), float(fy), 1.0, 1.0])
    K[:2, 2] = (mx, my)
    return K


def adjust_intrinsic(intrinsic, intrinsic_image_dim, image_dim):
    """Rescale a pinhole matrix from the calibration resolution to the working resolution, IN PLACE (the caller's matrix is
    returned, untouched when the two resolutions agree) -- the contract of fusion_util.py:27-39.  Focal lengths scale with the
    aspect-preserving resize (the width the calibration image would have at the new height), the principal point with the pixel
    CENTRES (dim - 1): the reference's crop convention, pinned by tests/golden/fusion_mapping.npz."""
    (w0, h0), (w1, h1) = intrinsic_image_dim, image_dim
    if (w0, h0) == (w1, h1):                       # (every factor below would be exactly 1.0)
        return intrinsic
    w_keep_aspect = int(math.floor(h1 * float(w0) / float(h0)))
    scale = {(0, 0): float(w_keep_aspect) / float(w0), (1, 1): float(h1) / float(h0),
             (0, 2): float(w1 - 1) / float(w0 - 1), (1, 2): float(h1 - 1) / float(h0 - 1)}
    for cell, f in scale.items():
        intrinsic[cell] *= f
    return intrinsic


def _dev_f64(x, device):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64))
    return x.to(device=device, dtype=torch.float64)


class PointCloudToImageMapper(object):
    def __init__(self, image_dim, visibility_threshold=0.25, cut_bound=0, intrinsics=None, device="cuda"):
        self.image_dim = image_dim                      # (width, height), as in the reference
        self.vis_thres = visibility_threshold
        self.cut_bound = cut_bound
        self.intrinsics = intrinsics
        self.device = torch.device(device)

    def compute_mapping(self, camera_to_world, coords, depth=None, intrinsic=None):
        """camera_to_world 4 x 4, coords N x 3, depth H x W (metres) or None, intrinsic 3 x 3 / 4 x 4
        -> int64 [N, 3] device tensor (row, column, mask), zeros where the point is not visible."""
        if self.intrinsics is not None:                 # global intrinsics
            intrinsic = self.intrinsics
        world_to_camera = np.linalg.inv(np.asarray(camera_to_world, dtype=np.float64))
        coords = _dev_f64(coords, self.device)
        depth = None if depth is None else _dev_f64(depth, self.device)
        k = np.asarray(intrinsic, dtype=np.float64)
        mapping = ops.fusion_project(coords, world_to_camera, (k[0][0], k[1][1], k[0][2], k[1][2]), depth,
                                     (self.image_dim[1], self.image_dim[0]), self.cut_bound, self.vis_thres)
        mapping.image_hw = (int(self.image_dim[1]), int(self.image_dim[0]))     # lets add_view refuse a feature map of another size
        return mapping


class FeatureFusion(object):
    """sum / counter of scannet_openseg.py:75-77, one `add_view` per image (:93-106), `finish` = :108-110."""

    def __init__(self, n_points, feat_dim, device="cuda"):
        self.sum_features = torch.zeros((n_points, feat_dim), dtype=torch.float32, device=device)
        self.counter = torch.zeros((n_points, 1), dtype=torch.float32, device=device)

    def add_view(self, feat_2d, mapping):
        """feat_2d float [D, H, W] (the extractor's output, permuted as in fusion_util.py:57-66), mapping from
        compute_mapping (its image size must be feat_2d's H x W: IndexError otherwise, as the reference's indexing).
        Always accumulates and returns True: a view without visible points adds nothing (the reference skips it,
        :90-91, after a host-side `mapping[:, 2].sum() == 0` -- the same result without the read-back)."""
        ops.fusion_accumulate(feat_2d, mapping, self.sum_features, self.counter, getattr(mapping, "image_hw", None))
        return True

    def finish(self):
        """-> (feat_bank [N, D] fp32, point_ids int64 [M] of the points seen in at least one view)."""
        bank = ops.fusion_finish(self.sum_features, self.counter)
        point_ids = torch.nonzero(self.counter[:, 0] > 0, as_tuple=False)[:, 0]
        return bank, point_ids
