"""The REAL host path (openscene_amd.ops wrappers, autograd glue, modules, optimizer step) executed on the
CPU against a mock of the C library: every entry point of include/openscene_amd.h is replaced by a C stub
that returns OSN_OK (generated from the ctypes prototypes, compiled with gcc), the map-building ops by the
CPU stand-in so that sizes are real.  No numerics are checked here (the GPU tests do that); what this
guards is that the wrapper layer runs end to end -- argument counts and types against the prototypes,
shape logic, workspace handling, autograd wiring -- and how many C-ABI calls a training step makes."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import cpu_backend


@pytest.fixture()
def mock_ops(monkeypatch):
    import host_profile
    from openscene_amd import _lib, ops
    mock = host_profile.build_mock()
    calls = {}

    class Counting:
        """Counts calls per entry point; ctypes still checks every argument against the prototype."""

        def __getattr__(self, name):
            fn = getattr(mock, name)

            def wrapped(*a):
                calls[name] = calls.get(name, 0) + 1
                return fn(*a)
            return wrapped

    lib = Counting()
    monkeypatch.setattr(_lib, "_lib", lib)
    monkeypatch.setattr(_lib, "require_device", lambda dev: None)
    monkeypatch.setattr(ops, "_prep", lambda dev: lib)
    monkeypatch.setattr(ops, "_stream", lambda dev: None)
    monkeypatch.setattr(ops, "_idx", lambda dev: 0)
    monkeypatch.setattr(ops, "_tl_counters", {})
    monkeypatch.setattr(ops, "_ws", lambda nbytes, dev: torch.empty(max(int(nbytes), 16), dtype=torch.uint8))
    monkeypatch.setattr(ops, "_size_cache", {})
    monkeypatch.setattr(ops, "_plan_cache", {})

    class NoDev:
        def __init__(self, dev):
            pass

        def __enter__(self):
            pass

        def __exit__(self, *a):
            pass
    monkeypatch.setattr(ops, "_Dev", NoDev)
    for n in ("HashTable", "coords_unique", "coords_pyramid", "kmap_build", "kmap_transpose", "kmap_sort", "kmap_count"):
        monkeypatch.setattr(ops, n, getattr(cpu_backend, n))
    return calls


def test_training_step_runs_through_the_real_wrappers(mock_ops):
    from openscene_amd import ops
    from openscene_amd import synthetic as syn
    ops.clear_weight_cache()
    from openscene_amd.disnet import DisNet
    from openscene_amd.sparse import SparseTensor

    class Cfg:
        arch_3d = "MinkUNet18A"
        feature_2d_extractor = "openseg"
    torch.manual_seed(0)
    model = DisNet(Cfg())
    optim = torch.optim.Adam(model.parameters(), lr=1e-4)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0, n_pts=20000), 0.05), 0)
    coords = torch.from_numpy(syn.batch_coords([vox]))
    feats = torch.ones(coords.shape[0], 3)
    for _ in range(2):
        mock_ops.clear()
        out = model(SparseTensor(feats, coords))
        assert out.shape == (coords.shape[0], 768)
        loss = out.sum()
        optim.zero_grad(set_to_none=True)
        loss.backward()
        optim.step()
    c = dict(mock_ops)
    from openscene_amd import minkowski as me
    n_conv = sum(isinstance(m, (me.MinkowskiConvolution, me.MinkowskiConvolutionTranspose)) for m in model.modules())
    n_bn = sum(isinstance(m, me.MinkowskiBatchNorm) for m in model.modules())
    assert (n_conv, n_bn) == (49, 48)                 # MinkUNet18A (a9 / a10)
    # one forward launch per conv, one input gradient each except the stem (its input needs none), one weight
    # gradient each; three BN calls per BatchNorm; at most one weight-prep launch per conv
    n_fwd = c.get("osn_spconv_fwd", 0) + c.get("osn_spconv_fwd_x6", 0) + c.get("osn_spconv_fwd_tl", 0) + c.get("osn_spconv_fwd_tl_pc", 0) + c.get("osn_stem_conv_fwd", 0) + c.get("osn_spconv_fwd_ws", 0) + c.get("osn_dense_fwd", 0)
    assert n_fwd == 2 * n_conv - 1, c
    assert c.get("osn_spconv_wgrad", 0) + c.get("osn_spconv_wgrad_tl", 0) + c.get("osn_stem_conv_wgrad", 0) == n_conv, c
    assert c.get("osn_pair_lists_build", 0) <= 10, c            # pair arrays: once per map, not per conv
    assert c["osn_bn_forward_train"] == n_bn and c["osn_bn_backward_multi2"] == n_bn, c  # one C call per BN and direction
    # weight images of parameters: served from the per-device cache, refreshed by ONE batched launch per optimizer step
    # (this is the second step: every image exists, every parameter version has changed once)
    assert c.get("osn_weight_prep_x6_pair", 0) + c.get("osn_weight_prep_x6", 0) + c.get("osn_weight_prep_tl", 0) == 0, c
    assert c.get("osn_weight_prep_batch", 0) == 1, c
    assert all(p.grad is not None for p in model.parameters())
    # a third forward after the optimizer step: one refresh; an eval forward with unchanged weights: none at all
    mock_ops.clear()
    model(SparseTensor(feats, coords))
    assert mock_ops.get("osn_weight_prep_batch", 0) == 1, dict(mock_ops)
    mock_ops.clear()
    model.eval()
    with torch.no_grad():
        model(SparseTensor(feats, coords))
    assert mock_ops.get("osn_weight_prep_batch", 0) == 0 and "osn_bn_forward_train" not in mock_ops, dict(mock_ops)
    # a write that bumps the version of ONE parameter refreshes (only what is stale, in one launch)
    with torch.no_grad():
        model.net3d.final.kernel.mul_(1.0)
        mock_ops.clear()
        model(SparseTensor(feats, coords))
    assert mock_ops.get("osn_weight_prep_batch", 0) == 1, dict(mock_ops)
    ops.clear_weight_cache()


def test_loader_and_query_wrappers_run(mock_ops):
    from openscene_amd import ops
    x = torch.randn(100, 64)
    text = torch.randn(20, 64).half()
    idx = torch.randint(0, 100, (150,))
    from openscene_amd.query import query_distill
    pred = query_distill(x, text, idx)
    assert pred.shape[0] == 150
    mv, src, ind = ops.feature_remap(torch.ones(50, dtype=torch.bool), torch.arange(0, 50, 2))
    assert mv.shape == (25,) and src.shape == (25,)
    out = torch.empty(10, 4, dtype=torch.int32)
    ops.batch_coords(torch.zeros(10, 3, dtype=torch.int32), 1, out)
    assert mock_ops["osn_feature_remap"] == 1 and mock_ops["osn_batch_coords"] == 1
