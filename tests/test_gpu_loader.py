"""HIP loader-side batch assembly (osn_feature_remap, osn_batch_coords, openscene_amd.loader) --
bit-exact (integer / gather work) against the reference's real FusedFeatureLoader outputs
(tests/golden/loader_fused.npz) and, at full size, against the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import loader as ol

import loader_cases

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


@pytest.mark.parametrize("split,eval_all,input_color", [("train", False, False), ("val", True, True)])
def test_loader_matches_the_reference_loader(golden_dir, split, eval_all, input_color):
    d = loader_cases.load(golden_dir)
    loader_cases.check(d, loader_cases.run(d, dev(), split, eval_all, input_color), split, eval_all)


@pytest.mark.parametrize("tag,input_color", [("ones", False), ("color", True), ("color2", True)])
def test_loader_training_augmentation_matches_the_reference_loader(golden_dir, tag, input_color):
    """dataset/feature_loader.py with aug=True (point_loader.py:101-113), outputs of the reference's own loader."""
    d = loader_cases.load_aug(golden_dir)
    loader_cases.check_aug(d, loader_cases.run_aug(d, dev(), tag, input_color), tag)


def test_remap_full_size_and_edges():
    from openscene_amd import ops, _lib
    rng = np.random.default_rng(9)
    n_pts, n_vox = 550000, 230000                       # the largest ScanNet scene
    mask = rng.random(n_pts) < 0.55
    vox = np.sort(rng.choice(n_pts, n_vox, replace=False))[rng.permutation(n_vox)].astype(np.int64)
    want = ol.remap(mask, vox)
    got = ops.feature_remap(torch.from_numpy(mask).to(dev()), torch.from_numpy(vox).to(dev()))
    for w, g in zip(want, got):
        assert np.array_equal(w, g.cpu().numpy())
    # the selected rows are a strictly increasing function of the point index: order-preserving compaction
    sel_pts = vox[want[0]]
    assert np.array_equal(np.argsort(sel_pts, kind="stable"), np.argsort(got[2].cpu().numpy(), kind="stable"))
    # nothing / everything selected, empty voxel list, uint8 mask
    none = ops.feature_remap(torch.zeros(1000, dtype=torch.bool, device=dev()), torch.arange(0, 1000, 3, device=dev()))
    assert none[2].numel() == 0 and not none[0].any() and bool((none[1] == -1).all())
    allsel = ops.feature_remap(torch.ones(1000, dtype=torch.uint8, device=dev()), torch.arange(999, -1, -1, device=dev()))
    assert torch.equal(allsel[2].cpu(), torch.arange(999, -1, -1)) and bool(allsel[0].all())
    empty = ops.feature_remap(torch.ones(10, dtype=torch.bool, device=dev()), torch.zeros(0, dtype=torch.int64, device=dev()))
    assert empty[0].numel() == 0 and empty[2].numel() == 0
    with pytest.raises(_lib.OpenSceneAmdError):
        ops.feature_remap(torch.ones(10, dtype=torch.bool, device=dev()), torch.tensor([3, 10], device=dev()))


def test_batch_coords_rows():
    from openscene_amd import ops
    xyz = torch.randint(-500, 500, (1001, 3), dtype=torch.int32, device=dev())
    out = torch.full((2002, 4), -7, dtype=torch.int32, device=dev())
    ops.batch_coords(xyz, 3, out[1001:])
    assert bool((out[:1001] == -7).all())
    assert bool((out[1001:, 0] == 3).all()) and torch.equal(out[1001:, 1:], xyz)
