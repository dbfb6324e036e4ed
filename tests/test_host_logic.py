"""Host logic of openscene_amd (coordinate manager, autograd wiring, fused module
tree, ME alias, state-dict layout) exercised WITHOUT a GPU: ``tests/cpu_backend.py``
stands in for the HIP ops, and the result is checked against the independent
end-to-end oracle ``oracle.sparse_ops.unet_forward`` (float64)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import sparse_ops as so

import cpu_backend

REFERENCE = "/root/reference"


def cloud(seed=0, n=260, extent=11, batch=2):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(batch):
        g = np.unique(rng.integers(0, extent, (n, 3)), axis=0)
        g = g[rng.permutation(g.shape[0])]
        rows.append(np.concatenate([np.full((g.shape[0], 1), b), g], 1))
    return torch.from_numpy(np.concatenate(rows, 0).astype(np.int32))


def state_to_oracle(model):
    return {k: v.detach().clone().double() for k, v in model.state_dict().items() if v.dtype.is_floating_point}


@pytest.fixture
def cpu_ops(monkeypatch):
    cpu_backend.install(monkeypatch)


@pytest.mark.parametrize("arch,train", [("MinkUNet14A", True), ("MinkUNet14A", False), ("MinkUNet18A", True)])
def test_unet_matches_oracle(cpu_ops, arch, train):
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(3)
    model = mink_unet(3, 12, 3, arch).double()
    # make eval-mode statistics non-trivial
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    model.train(train)
    coords = cloud(1)
    feats = torch.rand(coords.shape[0], 3, dtype=torch.float64)
    p = state_to_oracle(model)
    for v in p.values():
        v.requires_grad_(True)
    ref = so.unet_forward({k: (v if "running" not in k else v.detach().clone()) for k, v in p.items()},
                          feats, coords.numpy(), arch, train=train)
    out = model(SparseTensor(feats, coords))
    assert out.shape == (coords.shape[0], 12)
    assert torch.allclose(out, ref, rtol=1e-9, atol=1e-9)

    target = torch.randn_like(out)
    (out * target).sum().backward()
    (ref * target).sum().backward()
    checked = 0
    for name, prm in model.named_parameters():
        g_ref = p[name].grad
        assert g_ref is not None, name
        scale = g_ref.abs().max().item() + 1e-12
        assert (prm.grad - g_ref).abs().max().item() <= 1e-8 * scale + 1e-10, name
        checked += 1
    assert checked == len(list(model.parameters()))
    if train:   # running statistics moved exactly like torch's BatchNorm1d
        ref_p = state_to_oracle(model)
        assert int(model.bn0.bn.num_batches_tracked) == 1
        assert not torch.equal(ref_p["bn0.bn.running_mean"], p["bn0.bn.running_mean"].detach())


def test_running_stats_match_torch_batchnorm(cpu_ops):
    from openscene_amd import functional as F_
    torch.manual_seed(0)
    bn_a = torch.nn.BatchNorm1d(8).double()
    bn_b = torch.nn.BatchNorm1d(8).double()
    x = torch.randn(50, 8, dtype=torch.float64) * 3 + 1
    ya = bn_a(x)
    yb = F_.batch_norm_act(x, bn_b)
    assert torch.allclose(ya, yb, atol=1e-12)
    assert torch.allclose(bn_a.running_mean, bn_b.running_mean, atol=1e-12)
    assert torch.allclose(bn_a.running_var, bn_b.running_var, atol=1e-12)
    assert int(bn_b.num_batches_tracked) == 1


def test_state_dict_layout():
    from openscene_amd.disnet import DisNet

    class Cfg:
        arch_3d = "MinkUNet18A"
        feature_2d_extractor = "openseg"

    net = DisNet(Cfg())
    sd = net.state_dict()
    assert sd["net3d.conv0p1s1.kernel"].shape == (125, 3, 32)
    assert sd["net3d.final.kernel"].shape == (96, 768)              # K == 1 kernels are 2-D like ME's
    assert sd["net3d.block2.0.downsample.0.kernel"].shape == (32, 64)
    assert sd["net3d.block2.0.downsample.1.bn.running_var"].shape == (64,)
    assert sd["net3d.convtr4p16s2.kernel"].shape == (8, 256, 128)
    assert sd["net3d.block1.0.norm1.bn.weight"].shape == (32,)
    n_conv = sum(v.numel() for k, v in sd.items() if k.endswith(".kernel"))
    assert n_conv == 15_554_272
    oracle_keys = {"net3d." + k for k in so.init_params("MinkUNet18A", 3, 768)}
    mine = {k for k in sd if "num_batches_tracked" not in k}
    assert mine == oracle_keys
    Cfg.feature_2d_extractor = "lseg"
    assert DisNet(Cfg()).state_dict()["net3d.final.kernel"].shape == (96, 512)


def test_init_scheme():
    """models/resnet_base.py:73-80: kaiming fan_out on convs, ME default uniform on transposed convs."""
    from openscene_amd.mink_unet import mink_unet
    torch.manual_seed(0)
    m = mink_unet(3, 20, 3, "MinkUNet18A")
    k = m.block3[1].conv1.kernel                    # [27, 128, 128]
    assert abs(k.std().item() - (2.0 / (128 * 27)) ** 0.5) / (2.0 / (128 * 27)) ** 0.5 < 0.02
    t = m.convtr4p16s2.kernel                       # [8, 256, 128], U(+-1/sqrt(128*8))
    s = 1.0 / (128 * 8) ** 0.5
    assert t.abs().max().item() <= s and t.abs().max().item() > 0.95 * s
    assert torch.all(m.bn0.bn.weight == 1) and torch.all(m.bn0.bn.bias == 0)


def test_duplicate_coordinates_keep_first(cpu_ops):
    from openscene_amd.sparse import SparseTensor
    c = torch.tensor([[0, 1, 1, 1], [0, 2, 2, 2], [0, 1, 1, 1], [0, 3, 3, 3]], dtype=torch.int32)
    f = torch.arange(4.0).reshape(4, 1)
    st = SparseTensor(f, c)
    assert st.F.reshape(-1).tolist() == [0.0, 1.0, 3.0]
    assert st.coordinate_manager.inverse_mapping.tolist() == [0, 1, 0, 2]


def test_cat_and_iadd_require_same_map(cpu_ops):
    from openscene_amd.sparse import SparseTensor, cat
    def line(y):
        return torch.tensor([[0, x, y, 0] for x in range(5)], dtype=torch.int32)
    a = SparseTensor(torch.ones(5, 2), line(0))
    b = a._like(torch.full((5, 3), 2.0))
    assert cat(a, b).F.shape == (5, 5)
    other = SparseTensor(torch.ones(5, 2), line(1))
    with pytest.raises(ValueError):
        cat(a, other)
    a += a._like(torch.ones(5, 2))
    assert torch.all(a.F == 2)


def test_no_cpu_fallback_in_product():
    """CPU tensors must be refused by the real ops (no silent fallback)."""
    from openscene_amd import ops, _lib
    with pytest.raises(_lib.OpenSceneAmdError):
        ops.spconv_fwd(torch.ones(4, 4), torch.ones(1, 4, 4), None, 4)
    with pytest.raises(_lib.OpenSceneAmdError):
        ops.coords_unique(torch.zeros((3, 4), dtype=torch.int32))


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present on this machine")
def test_reference_model_file_runs_unchanged(cpu_ops, monkeypatch):
    """The reference's own models/mink_unet.py + models/disnet.py import `MinkowskiEngine`, get
    openscene_amd.minkowski through the alias, and produce the same numbers as our mirror."""
    import openscene_amd
    from openscene_amd.mink_unet import mink_unet as mine
    from openscene_amd.sparse import SparseTensor
    for name in [m for m in sys.modules if m == "MinkowskiEngine" or m.startswith("MinkowskiEngine.")
                 or m == "models" or m.startswith("models.")]:
        monkeypatch.delitem(sys.modules, name)
    openscene_amd.install_minkowski_alias()
    monkeypatch.syspath_prepend(REFERENCE)
    from models.mink_unet import mink_unet as theirs            # noqa: E402  (reference file, unmodified)
    import MinkowskiEngine as ME
    assert ME.__openscene_amd__
    torch.manual_seed(0)
    ref_model = theirs(3, 10, 3, "MinkUNet14A").double().eval()
    my_model = mine(3, 10, 3, "MinkUNet14A").double().eval()
    assert list(ref_model.state_dict().keys()) == list(my_model.state_dict().keys())
    assert [tuple(v.shape) for v in ref_model.state_dict().values()] == \
           [tuple(v.shape) for v in my_model.state_dict().values()]
    my_model.load_state_dict(ref_model.state_dict(), strict=True)
    coords = cloud(5)
    feats = torch.rand(coords.shape[0], 3, dtype=torch.float64)
    a = ref_model(ME.SparseTensor(feats, coords))
    b = my_model(SparseTensor(feats, coords))
    assert torch.allclose(a, b, atol=1e-10)
    ref_model.train(); my_model.train()
    a = ref_model(ME.SparseTensor(feats, coords))
    b = my_model(SparseTensor(feats, coords))
    assert torch.allclose(a, b, atol=1e-9)
    for name in [m for m in sys.modules if m == "MinkowskiEngine" or m.startswith("MinkowskiEngine.")
                 or m == "models" or m.startswith("models.")]:
        monkeypatch.delitem(sys.modules, name)


def test_tile_ordered_maps_do_not_change_results(cpu_ops, monkeypatch):
    """kmap_tiles (rows ordered by offset-occupancy mask + out_rows indirection) is a pure
    scheduling change: forward and every gradient are identical with and without it."""
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, SparseTensor
    torch.manual_seed(5)
    model = mink_unet(3, 8, 3, "MinkUNet14A").double()
    coords = cloud(9)
    feats = torch.rand(coords.shape[0], 3, dtype=torch.float64)
    res = []
    for min_rows in (10 ** 9, 0):
        monkeypatch.setattr(CoordinateManager, "SORT_MIN_ROWS", min_rows)
        monkeypatch.setattr(CoordinateManager, "SORT_MIN_ROWS_K8", min_rows)      # (2^3 maps are not ordered by default)
        model.zero_grad()
        st = SparseTensor(feats, coords)
        out = model(st)
        out.square().sum().backward()
        used = [k for k, v in st.coordinate_manager._kmaps.items() if k[0] == "tiles" and any(t is not None for t in v)]
        res.append((out.detach().clone(), [p.grad.clone() for p in model.parameters()], used))
    assert not res[0][2] and len(res[1][2]) >= 9          # 5 stride-1 maps + 4 down + 4 up tables
    assert torch.allclose(res[0][0], res[1][0], atol=1e-12)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, atol=1e-10 * (1 + a.abs().max().item()))


@pytest.mark.parametrize("split,eval_all,input_color", [("train", False, False), ("val", True, True)])
def test_gpu_resident_loader_matches_the_reference_loader(monkeypatch, golden_dir, split, eval_all, input_color):
    """openscene_amd.loader (host logic, stand-in ops) == the reference's FusedFeatureLoader + collation."""
    import loader_cases
    cpu_backend.install(monkeypatch)
    d = loader_cases.load(golden_dir)
    loader_cases.check(d, loader_cases.run(d, torch.device("cpu"), split, eval_all, input_color), split, eval_all)


@pytest.mark.parametrize("tag,input_color", [("ones", False), ("color", True), ("color2", True)])
def test_loader_training_augmentation_matches_the_reference_loader(monkeypatch, golden_dir, tag, input_color):
    """aug=True (the training configuration): elastic-distortion draws, horizontal flip, chromatic transforms -- same
    random stream, bit-identical batch as the reference's FusedFeatureLoader (tests/golden/loader_fused_aug.npz)."""
    import loader_cases
    cpu_backend.install(monkeypatch)
    d = loader_cases.load_aug(golden_dir)
    loader_cases.check_aug(d, loader_cases.run_aug(d, torch.device("cpu"), tag, input_color), tag)


def test_forward_features_is_the_input_of_the_final_conv(cpu_ops):
    """MinkUNetBase.forward == final(forward_features): the fused-head query folds exactly
    the last 1x1 convolution (models/mink_unet.py:108-113,174) and nothing else."""
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.query import head_times_text
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(9)
    model = mink_unet(3, 24, 3, "MinkUNet14A").double().eval()
    coords = cloud(4)
    feats = torch.rand(coords.shape[0], 3, dtype=torch.float64)
    out = model(SparseTensor(feats, coords))
    f = model.forward_features(SparseTensor(feats, coords))
    assert f.shape == (coords.shape[0], 96)
    assert torch.allclose(out, f @ model.final.kernel, atol=1e-12)
    text = torch.randn(5, 24, dtype=torch.float64)
    assert torch.allclose(out @ text.t(), f @ head_times_text(model.final.kernel, text).double(), atol=1e-5)


@pytest.mark.parametrize("loss_type", ["cosine", "l1"])
def test_distill_loss_oracle_is_the_reference_expression(loss_type):
    """oracle/loss.py pinned to torch's own forward and autograd of run/distill.py:322-328 (float64): the loss on
    `output[mask]` and its gradient with respect to the FULL output -- zeros on the rows the mask leaves out."""
    from oracle import loss as ol
    g = torch.Generator().manual_seed(3)
    n, d = 300, 48
    out = torch.randn(n, d, generator=g, dtype=torch.float64) * 3
    mask = torch.zeros(n, dtype=torch.bool)
    mask[torch.randperm(n, generator=g)[:120]] = True
    feat_3d = torch.nn.functional.normalize(torch.randn(120, d, generator=g, dtype=torch.float64), dim=1)
    out[mask.nonzero()[5]] = 0.0                                   # a zero row: CosineSimilarity's eps clamp
    x = out.clone().requires_grad_(True)
    output_3d = x[mask]
    if loss_type == "cosine":
        ref = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()
    else:
        ref = torch.nn.L1Loss()(output_3d, feat_3d)
    ref.backward()
    loss, grad = ol.distill_loss(out.numpy(), mask.nonzero().squeeze(1).numpy(), feat_3d.numpy(), loss_type)
    assert abs(loss - ref.item()) <= 1e-12 * max(1.0, abs(ref.item()))
    np.testing.assert_allclose(grad, x.grad.numpy(), rtol=1e-10, atol=1e-14)
    assert not grad[~mask.numpy()].any()


def test_sparse_add_never_broadcasts():
    """tensors on one coordinate map have equal shapes; a mismatch is a coordinate-map bug and raises (MinkowskiEngine's
    sparse add does not broadcast either), it does not fall back to torch's broadcasting `a + b`."""
    import pytest
    import torch
    from openscene_amd import functional as F_
    with pytest.raises(ValueError, match="same coordinate map"):
        F_.add(torch.zeros(8, 4), torch.zeros(1, 4))
    with pytest.raises(ValueError, match="same coordinate map"):
        F_.add(torch.zeros(8, 4), torch.zeros(8, 1))
