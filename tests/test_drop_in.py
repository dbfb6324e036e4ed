"""The fast path behind the reference's OWN class: after `install_minkowski_alias()`, a module named
`*.mink_unet` written against `import MinkowskiEngine` -- the reference's models/mink_unet.py where it exists, and the
stand-in tests/foreign/mink_unet.py everywhere -- gets the network executor behind an unchanged `model(sinput)` call, and
falls back to its own forward for every pass the executor does not compile.  CPU only: launches are checked with the null
HIP runtime of tools/dryrun, numbers with tests/cpu_backend.py (the GPU parity of the same path is tests/test_gpu_unet.py)."""
import collections
import copy
import io
import os
import pickle
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def _purge(monkeypatch):
    for name in [m for m in sys.modules if m == "MinkowskiEngine" or m.startswith("MinkowskiEngine.") or m == "models"
                 or m.startswith("models.") or m == "foreign" or m.startswith("foreign.")]:
        monkeypatch.delitem(sys.modules, name)


@pytest.fixture()
def alias(monkeypatch):
    import openscene_amd
    from openscene_amd import drop_in
    _purge(monkeypatch)
    openscene_amd.install_minkowski_alias()
    monkeypatch.syspath_prepend(HERE)
    yield openscene_amd
    drop_in.remove_import_hook()
    _purge(monkeypatch)


def _factories(monkeypatch):
    out = {}
    import foreign.mink_unet as fm
    out["foreign"] = lambda cin, cout, D, arch: getattr(fm, arch)(cin, cout, D)
    if os.path.isdir(REFERENCE):
        monkeypatch.syspath_prepend(REFERENCE)
        from models.mink_unet import mink_unet as theirs            # the reference file, unmodified
        out["reference"] = theirs
    return out


def test_the_import_hook_gives_foreign_classes_the_dispatcher(alias, monkeypatch):
    from openscene_amd import drop_in, executor as E
    for name, make in _factories(monkeypatch).items():
        model = make(3, 16, 3, "MinkUNet14A")
        assert "openscene_amd" not in type(model).__module__
        assert drop_in.is_unet(model) and drop_in.accelerated(model), name
        f = type(model).forward
        assert f.__osn_accelerated__ and callable(f.__wrapped__)
        ex = E.for_model(model)
        assert ex is not None and {id(q) for q in ex.program.params} == {id(q) for q in model.parameters()}, name
        alias.accelerate(model)                                        # explicit call: idempotent
        assert type(model).forward is f


def test_explicit_accelerate_on_an_instance_inside_a_wrapper(alias, monkeypatch):
    """No import hook (a file with another name, or the alias installed late): accelerate(model) on a DisNet-like wrapper."""
    from openscene_amd import drop_in
    drop_in.remove_import_hook()
    for m in [m for m in sys.modules if m.startswith("foreign")]:
        monkeypatch.delitem(sys.modules, m)
    import foreign.mink_unet as fm
    assert not getattr(fm.MinkUNetBase.forward, "__osn_accelerated__", False)

    class Wrapper(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net3d = fm.MinkUNet14A(3, 8, 3)

        def forward(self, x):
            return self.net3d(x)
    w = Wrapper()
    assert alias.accelerate(w) is w and drop_in.accelerated(w.net3d)
    with pytest.raises(TypeError):
        alias.accelerate(torch.nn.Linear(3, 3))


def test_foreign_model_falls_back_to_its_own_forward_and_matches_the_mirror(alias, monkeypatch):
    """On host tensors the executor is not usable: the dispatcher must run the ORIGINAL un-fused forward; same numbers as the
    mirror with the same state dict."""
    sys.path.insert(0, HERE)
    import cpu_backend
    import openscene_amd.ops as ops
    for n in cpu_backend._NAMES:
        monkeypatch.setattr(ops, n, getattr(cpu_backend, n))
    from openscene_amd.mink_unet import mink_unet as mine
    from openscene_amd.sparse import SparseTensor
    import numpy as np
    rng = np.random.default_rng(5)
    g = np.unique(rng.integers(0, 9, (200, 3)), axis=0)
    coords = torch.from_numpy(np.concatenate([np.zeros((g.shape[0], 1)), g], 1).astype(np.int32))
    feats = torch.rand(coords.shape[0], 3, dtype=torch.float64)
    for name, make in _factories(monkeypatch).items():
        torch.manual_seed(0)
        theirs = make(3, 10, 3, "MinkUNet14A").double().train()
        ours = mine(3, 10, 3, "MinkUNet14A").double().train()
        assert list(theirs.state_dict().keys()) == list(ours.state_dict().keys()), name
        ours.load_state_dict(theirs.state_dict())
        a = theirs(SparseTensor(feats, coords))
        b = ours(SparseTensor(feats, coords))
        assert torch.allclose(a, b, atol=1e-9), name
        a.square().sum().backward()
        b.square().sum().backward()
        for (n1, p1), (_, p2) in zip(theirs.named_parameters(), ours.named_parameters()):
            assert torch.allclose(p1.grad, p2.grad, atol=1e-8, rtol=1e-6), (name, n1)


@pytest.mark.parametrize("arch", ["MinkUNet18A"])
def test_dry_run_foreign_model_plays_the_executors_stage_program(arch, alias, monkeypatch):
    """Launch multiset (kernel instance, grid) of a training step of the foreign / reference class through the alias
    == the mirror's executor step: fused stages, no cat / add / relu kernels, tile-list + pair-array weight-gradient kernels."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "dryrun"))
    import dry_step
    lib = dry_step.install(dry_step.build_dry_lib(), monkeypatch.setattr)
    ref_logs, _s, sizes = dry_step.step_logs(lib, arch, points=12000, setattr_=monkeypatch.setattr)
    want = [l for l in ref_logs["executor"].split("\n") if l.startswith("K ") or " K " in l]
    for name, make in _factories(monkeypatch).items():
        logs, _s2, sizes2 = dry_step.step_logs(lib, arch, points=12000, setattr_=monkeypatch.setattr, make_model=make)
        assert sizes2 == sizes
        got = logs["executor"]
        assert collections.Counter(dry_step.conv_launches(got)) == collections.Counter(dry_step.conv_launches(ref_logs["executor"])), name
        assert collections.Counter(dry_step.bn_launches(got)) == collections.Counter(dry_step.bn_launches(ref_logs["executor"])), name
        assert "cat2_kernel" not in got and "ew_kernel" not in got, name
        # and with the executor off the same class runs its own un-fused chain (elementwise kernels present)
        assert "cat2_kernel" in logs["modules"] and "ew_kernel" in logs["modules"], name


def test_models_with_an_executor_stay_copyable_and_picklable(alias):
    """the executor holds ctypes structures with pointers; it must not live in the module's state."""
    from openscene_amd import executor as E
    from openscene_amd.mink_unet import mink_unet
    model = mink_unet(3, 8, 3, "MinkUNet14A")
    assert E.for_model(model) is not None
    assert "_osn_executor" not in model.__dict__
    twin = copy.deepcopy(model)
    assert E.for_model(twin) is not E.for_model(model)
    buf = io.BytesIO()
    torch.save(model, buf)
    pickle.dumps(model)
    del twin
    import gc
    gc.collect()
    assert len(E._EXECUTORS) >= 1


def test_executor_declines_passes_it_would_get_wrong(alias, monkeypatch):
    """an input-feature gradient, or BN modules whose training flags differ from the model's -> module path."""
    from openscene_amd import executor as E
    from openscene_amd.mink_unet import mink_unet
    monkeypatch.setattr(E, "_DRY_RUN", True)
    model = mink_unet(3, 8, 3, "MinkUNet14A").train()
    ex = E.for_model(model)

    class X:
        tensor_stride = 1
        F = torch.ones(5, 3)
    assert ex.usable(X(), model)
    X.F = torch.ones(5, 3, requires_grad=True)
    assert not ex.usable(X(), model)
    with torch.no_grad():
        assert ex.usable(X(), model)
    X.F = torch.ones(5, 3)
    model.bn0.eval()                                   # a frozen BN inside a training model
    assert not ex.usable(X(), model)
    model.eval()
    assert ex.usable(X(), model)
