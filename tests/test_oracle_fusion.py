"""oracle/fusion.py against the golden outputs of the reference's own PointCloudToImageMapper
(tests/golden/fusion_mapping.npz, minted by tests/golden/make_golden.py from scripts/feature_fusion/fusion_util.py)."""
import os

import numpy as np
import torch

from oracle import fusion as of

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fusion_mapping.npz"))


def test_intrinsics_helpers():
    raw = of.make_intrinsic(577.870605, 577.870605, 319.5, 239.5)
    assert np.array_equal(raw, G["intrinsic_raw"])
    assert np.array_equal(of.adjust_intrinsic(raw, [640, 480], (320, 240)), G["intrinsic"])
    assert of.adjust_intrinsic(raw, (320, 240), (320, 240)) is raw           # fusion_util.py:30-31


def test_mapping_bit_exact_vs_reference_outputs():
    dim = tuple(int(x) for x in G["image_dim"])
    both = 0
    for v in range(3):
        got = of.compute_mapping(G["pose%d" % v], G["coords"], G["depth%d" % v], G["intrinsic"], dim,
                                 float(G["vis_thres"]), int(G["cut_bound"]))
        want = G["mapping%d" % v]
        assert got.dtype == want.dtype and np.array_equal(got, want)
        both += int(want[:, 2].sum())
    assert both > 5000                                                        # the views do see the cloud
    got = of.compute_mapping(G["pose0"], G["coords"], None, G["intrinsic"], dim, float(G["vis_thres"]), int(G["cut_bound"]))
    assert np.array_equal(got, G["mapping_nodepth"])
    # occluded / invalid-depth points exist: in-frame without depth, rejected with it
    assert int(G["mapping_nodepth"][:, 2].sum()) > int(G["mapping0"][:, 2].sum())


def test_running_mean_over_views():
    n, d = G["coords"].shape[0], 16
    g = torch.Generator().manual_seed(0)
    s = torch.zeros(n, d)
    c = torch.zeros(n, 1)
    seen = np.zeros(n, bool)
    ref = torch.zeros(n, d, dtype=torch.float64)
    for v in range(3):
        f = torch.randn(d, 240, 320, generator=g)
        m = G["mapping%d" % v]
        of.accumulate(s, c, f, m)
        vis = m[:, 2] == 1
        seen |= vis
        ref[vis] += f[:, m[vis, 0], m[vis, 1]].T.double()
    bank = of.finish(s, c)
    cnt = sum((G["mapping%d" % v][:, 2] == 1).astype(np.float64) for v in range(3))
    assert np.array_equal(c[:, 0].numpy(), cnt.astype(np.float32))
    want = ref[seen] / torch.from_numpy(cnt[seen]).unsqueeze(1)
    assert (bank[seen].double() - want).abs().max().item() < 1e-5
    assert bank[~seen].abs().max().item() == 0.0                              # 0 / 1e-5
