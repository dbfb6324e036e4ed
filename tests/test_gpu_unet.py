"""End-to-end MinkUNet on the HIP kernels vs the float64 oracle network
(oracle.sparse_ops.unet_forward): forward in train and eval mode, all parameter
gradients, running statistics; the reference-shaped DisNet step; the ME alias.

Stated tolerance: fp32 network vs float64 oracle, relative L2 error
of the output <= 2e-4 and max |delta| <= 1e-3 * max |reference|; every parameter gradient
rel-L2 <= 2e-4 ON THE SAME ReLU ACTIVATION PATTERN: a pre-activation within fp32 rounding of
zero flips its ReLU between fp32 and float64 (any fp32 implementation does, a CPU fp32 run
flips the same elements), and a single flipped element moves a gradient's relative L2 error by
~1/sqrt(#elements) (2e-3 on these small scenes), so the oracle's backward is evaluated with the
activation pattern of the run under test (at most a handful of elements differ; asserted)."""
import numpy as np
import pytest
import torch

from oracle import sparse_ops as so
from openscene_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).norm() / (ref.norm() + 1e-30)).item()


class record_relu_masks:
    """Record (y > 0) of every fused BN(+residual)+ReLU, in call order (= the oracle's ReLU order), through the
    observer hook both the per-module path and the network executor honour."""

    def __init__(self):
        from openscene_amd import functional as F_
        self.F_ = F_
        self.masks = []
        F_.set_relu_observer(lambda y: self.masks.append((y.detach() > 0).cpu()))

    def stop(self):
        self.F_.set_relu_observer(None)
        return self.masks


@pytest.fixture
def modules_only(monkeypatch):
    """Run the model module by module (minkowski.py / functional.py), not through the network executor."""
    from openscene_amd import executor
    monkeypatch.setattr(executor, "ENABLED", False)


def scene_coords(seed, n_pts, voxel, batch=1):
    return syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(seed + b, n_pts=n_pts), voxel), seed + b)
                             for b in range(batch)])


@pytest.mark.parametrize("arch,out_dim,train,path", [("MinkUNet14A", 16, True, "executor"), ("MinkUNet18A", 20, False, "executor"),
                                                     ("MinkUNet18A", 64, True, "executor"), ("MinkUNet34C", 32, True, "executor"),
                                                     ("MinkUNet14A", 16, True, "modules"), ("MinkUNet18A", 20, False, "modules"),
                                                     ("MinkUNet18A", 64, True, "modules")])
def test_unet_vs_oracle(arch, out_dim, train, path, monkeypatch):
    """Both host paths: the network executor (one C call per pass, the default) and the per-module path (the surface
    the reference's own models/mink_unet.py runs on)."""
    from openscene_amd import executor
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    monkeypatch.setattr(executor, "ENABLED", path == "executor")
    torch.manual_seed(7)
    model = mink_unet(3, out_dim, 3, arch)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    model.train(train)
    coords = scene_coords(11, 7000, 0.04, batch=2)                  # ~2 x 5 k voxels
    feats = torch.rand(coords.shape[0], 3)
    p = {k: v.detach().clone().double() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    for k, v in p.items():
        if "running" not in k:
            v.requires_grad_(True)
    free = so.unet_forward({k: v.detach().clone() for k, v in p.items()}, feats.double(), coords, arch, train=train)

    model = model.to(dev())
    masks = record_relu_masks()
    out = model(SparseTensor(feats.to(dev()), torch.from_numpy(coords).to(dev())))
    masks = masks.stop()
    assert out.shape == free.shape and out.dtype == torch.float32
    e = rel_l2(out, free)
    assert e <= 2e-4, "output rel-L2 %.3e" % e
    assert (out.double().cpu() - free.detach()).abs().max().item() <= 1e-3 * free.abs().max().item()

    # float64 oracle on the activation pattern of the fp32 run (see module docstring)
    ref = so.unet_forward(p, feats.double(), coords, arch, train=train, relu_masks=masks)
    assert rel_l2(ref, free) <= 1e-6, "prescribing the fp32 activation pattern changed the oracle output"
    target = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (ref * target).sum().backward()
    (out * target.float().to(dev())).sum().backward()
    worst = ("", 0.0)
    for name, prm in model.named_parameters():
        g = rel_l2(prm.grad, p[name].grad)
        if g > worst[1]:
            worst = (name, g)
    assert worst[1] <= 2e-4, "gradient of %s rel-L2 %.3e" % worst
    if train:
        for name, buf in model.named_buffers():
            if "running" in name:
                assert rel_l2(buf, p[name]) <= 1e-5, name


def test_disnet_distill_step_and_row_order():
    """One reference-shaped distillation step (run/distill.py:315-334): shift coords, forward,
    output[mask], cosine loss, backward -- rows must come back in INPUT order."""
    from openscene_amd.disnet import DisNet
    from openscene_amd.sparse import SparseTensor

    class Cfg:
        arch_3d = "MinkUNet18A"
        feature_2d_extractor = "lseg"

    torch.manual_seed(0)
    net = DisNet(Cfg()).to(dev())
    coords = scene_coords(21, 9000, 0.04)
    coords[:, 1:4] += (np.random.default_rng(0).random(3) * 100).astype(np.int32)
    feats = torch.ones(coords.shape[0], 3)
    c = torch.from_numpy(coords).to(dev())
    out = net(SparseTensor(feats.to(dev()), c))
    assert out.shape == (coords.shape[0], 512)
    # permuting the input rows permutes the output rows identically (row-order contract)
    perm = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(3))
    net.eval()
    a = net(SparseTensor(feats.to(dev()), c))
    b = net(SparseTensor(feats.to(dev()), c[perm.to(dev())]))
    assert rel_l2(b, a[perm.to(dev())]) < 1e-5
    net.train()
    mask = torch.zeros(coords.shape[0], dtype=torch.bool)
    mask[torch.randperm(coords.shape[0])[:2000]] = True
    target = torch.nn.functional.normalize(torch.randn(2000, 512), dim=1).to(dev())
    out = net(SparseTensor(feats.to(dev()), c))
    loss = (1 - torch.nn.CosineSimilarity()(out[mask.to(dev())], target)).mean()
    loss.backward()
    assert torch.isfinite(loss).item()
    grads = [p.grad for p in net.parameters()]
    assert all(g is not None and torch.isfinite(g).all().item() for g in grads)
    assert sum(float(g.abs().sum()) for g in grads) > 0


def test_head_on_selected_rows_equals_the_indexed_full_output():
    """Round 6: model(sinput, rows=sel) runs the final 1x1 convolution (and, in training, both of its gradients) on the supervised
    rows only -- run/distill.py:321-322 indexes the output with the mask before anything reads it.  Forward rows bitwise the indexed
    full output (same kernel, same row-wise product); every parameter gradient equal to the full-output path's to fp32 round-off;
    eval mode; all rows / empty selection fall back to indexing."""
    from openscene_amd import losses
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(11)
    model = mink_unet(3, 768, 3, "MinkUNet14A").to(dev()).train()
    coords = torch.from_numpy(scene_coords(5, 9000, 0.05)).to(dev())
    n = coords.shape[0]
    feats = torch.rand(n, 3, device=dev())
    g = torch.Generator().manual_seed(3)
    sel = torch.randperm(n, generator=g)[: n // 5].sort()[0].to(dev())
    target = torch.nn.functional.normalize(torch.randn(sel.shape[0], 768, device=dev()), dim=1)
    res = []
    for rows in (True, False):
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        model.zero_grad(set_to_none=True)
        if rows:
            out = model(SparseTensor(feats, coords), rows=sel)
            loss = losses.distill_loss(out, None, target)
        else:
            full = model(SparseTensor(feats, coords))
            out = full.index_select(0, sel)
            loss = losses.distill_loss(full, sel, target)
        loss.backward()
        res.append((out.detach().clone(), float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]), "head on the selected rows differs from the indexed full output"
    assert abs(res[0][1] - res[1][1]) <= 1e-6 * abs(res[1][1])
    for k in res[0][2]:
        assert rel_l2(res[0][2][k], res[1][2][k]) <= 2e-6, k
    model.eval()
    with torch.no_grad():
        a = model(SparseTensor(feats, coords), rows=sel)
        b = model(SparseTensor(feats, coords)).index_select(0, sel)
        every = torch.arange(n, device=dev())
        c = model(SparseTensor(feats, coords), rows=every)
    assert torch.equal(a, b) and c.shape == (n, 768)
    with pytest.raises(ValueError):
        model(SparseTensor(feats, coords), rows=sel.int())


def test_row_sparse_head_with_a_narrow_head(monkeypatch):
    """A head of <= 128 channels plans the pair-array weight gradient; the row-compacted path runs the table kernel instead
    and needs its scratch (the executor's plan reserves it)."""
    from openscene_amd import executor as E, losses
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(4)
    model = mink_unet(3, 64, 3, "MinkUNet14A").to(dev()).train()
    coords = torch.from_numpy(scene_coords(9, 9000, 0.05)).to(dev())
    n = coords.shape[0]
    feats = torch.rand(n, 3, device=dev())
    sel = torch.arange(0, n, 4, device=dev())
    target = torch.nn.functional.normalize(torch.randn(sel.shape[0], 64, device=dev()), dim=1)
    res = []
    for sparse in (True, False):
        monkeypatch.setattr(E, "ROW_SPARSE_HEAD", sparse)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        model.zero_grad(set_to_none=True)
        losses.distill_loss(model(SparseTensor(feats, coords)), sel, target).backward()
        res.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    for k in res[0]:
        if k == "final.kernel":
            assert rel_l2(res[0][k], res[1][k]) <= 2e-6
        else:
            assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.parametrize("kind", ["cosine", "l1"])
def test_row_sparse_head_gradients_equal_the_dense_path(kind, monkeypatch):
    """Round 4: distill_loss hands the executor the non-zero rows of the output gradient (the loss sees `output[sel]`,
    run/distill.py:322) and the head's weight / input gradients run on those rows only.  Against the dense head backward
    (OSN_ROW_SPARSE_HEAD=0) and against torch's own `out[mask]` + CosineSimilarity / L1Loss chain: the input gradient rows
    are computed by the same kernel row by row (=> everything upstream is BITWISE equal), the head's weight gradient sums
    the same products without the zero rows (fp32 round-off)."""
    from openscene_amd import executor as E, losses
    from openscene_amd.disnet import DisNet
    from openscene_amd.sparse import SparseTensor

    class Cfg:
        arch_3d = "MinkUNet18A"
        feature_2d_extractor = "lseg"

    torch.manual_seed(11)
    net = DisNet(Cfg()).to(dev()).train()
    coords = torch.from_numpy(scene_coords(5, 12000, 0.04)).to(dev())
    n = coords.shape[0]
    feats = torch.rand(n, 3, device=dev())
    g = torch.Generator().manual_seed(2)
    sel = torch.randperm(n, generator=g)[:n // 5].sort()[0].to(dev())
    target = torch.nn.functional.normalize(torch.randn(sel.shape[0], 512, generator=g), dim=1).to(dev())
    used = []
    real = E.UNetExecutor._run_backward
    monkeypatch.setattr(E.UNetExecutor, "_run_backward",
                        lambda self, st, gout: (used.append(getattr(gout, "_osn_rows", None) is not None), real(self, st, gout))[1])

    def step(sparse, torch_loss=False):
        monkeypatch.setattr(E, "ROW_SPARSE_HEAD", sparse)
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        net.zero_grad(set_to_none=True)
        out = net(SparseTensor(feats, coords))
        if torch_loss:
            o = out.index_select(0, sel)
            loss = (1 - torch.nn.CosineSimilarity()(o, target)).mean() if kind == "cosine" else torch.nn.L1Loss()(o, target)
        else:
            loss = losses.distill_loss(out, sel, target, kind)
        loss.backward()
        return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}

    l_sparse, g_sparse = step(True)
    l_dense, g_dense = step(False)
    l_torch, g_torch = step(True, torch_loss=True)
    assert used == [True, True, False]                       # the hint travels with the HIP loss's gradient only
    assert torch.equal(l_sparse, l_dense)
    head = "net3d.final.kernel"
    for k in g_dense:
        if k == head:
            assert rel_l2(g_sparse[k], g_dense[k]) <= 2e-6, k
        else:
            assert torch.equal(g_sparse[k], g_dense[k]), k
        assert rel_l2(g_sparse[k], g_torch[k]) <= (2e-5 if kind == "cosine" else 2e-4), k


def test_row_hint_is_dropped_when_a_hook_edits_the_gradient_in_place():
    """The compacted rows that travel with distill_loss's gradient describe the gradient AS THE LOSS WROTE IT: a tensor hook on
    the network output that scales the gradient in place must reach every parameter, the head included (round 4's advisor
    finding: the head's two gradients would have been computed from the stale rows)."""
    from openscene_amd import losses
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(5)
    net = mink_unet(3, 64, 3, "MinkUNet14A").to(dev()).train()
    coords = torch.from_numpy(scene_coords(7, 6000, 0.05)).to(dev())
    n = coords.shape[0]
    feats = torch.rand(n, 3, device=dev())
    g = torch.Generator().manual_seed(3)
    sel = torch.randperm(n, generator=g)[:n // 4].sort()[0].to(dev())
    target = torch.nn.functional.normalize(torch.randn(sel.shape[0], 64, generator=g), dim=1).to(dev())

    def step(hook):
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        net.zero_grad(set_to_none=True)
        out = net(SparseTensor(feats, coords))
        if hook:
            out.register_hook(lambda gr: gr.mul_(2.0))
        losses.distill_loss(out, sel, target, "cosine").backward()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters()}

    plain, doubled = step(False), step(True)
    for k in plain:
        assert rel_l2(doubled[k], 2.0 * plain[k]) <= 2e-6, k


def test_segmented_backward_pass_is_bitwise_the_single_call(monkeypatch):
    """The backward pass played in 4 segments (what an attached gradient exchange does, distributed.FlatGradAllReduce.attach):
    the hook sees disjoint slices that tile the kernels' region of the flat gradient buffer, highest ops first, and every
    gradient is bitwise the one-call pass's."""
    from openscene_amd import executor as E
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(21)
    model = mink_unet(3, 64, 3, "MinkUNet18A").to(dev()).train()
    coords = torch.from_numpy(scene_coords(3, 20000, 0.04)).to(dev())
    feats = torch.rand(coords.shape[0], 3, device=dev())
    ex = E.for_model(model)

    def grads(hook, segments=4):
        ex.grad_ready_hook, ex.grad_segments, ex._cuts = hook, segments, None
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        model.zero_grad(set_to_none=True)
        out = model(SparseTensor(feats, coords))
        out.square().mean().backward()
        return [p.grad.detach().clone() for p in model.parameters()]

    try:
        ref = grads(None)
        seen = []
        got = grads(lambda flat, lo, hi, last: seen.append((lo, hi, last, flat.data_ptr())))
    finally:
        ex.grad_ready_hook = None
    assert len(seen) == 4 and [s[2] for s in seen] == [False, False, False, True]
    assert seen[0][1] == ex.conv_grad_end and seen[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(seen[:-1], seen[1:]))
    assert len({s[3] for s in seen}) == 1
    for (n, _), a, b in zip(model.named_parameters(), ref, got):
        assert torch.equal(a, b), n


def test_executor_equals_the_module_path(monkeypatch):
    """The executor plays the same kernels in the same order as the per-module path: forward outputs, feature taps and
    running statistics are BITWISE equal, in training and in evaluation mode; parameter gradients agree to fp32
    round-off (sums of three gradient sources are associated in consumer order instead of autograd's order)."""
    from openscene_amd import executor
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    d = dev()
    coords = torch.from_numpy(scene_coords(51, 30000, 0.03, batch=2)).to(d)
    feats = torch.rand(coords.shape[0], 3, device=d)
    target = torch.randn(coords.shape[0], 48, device=d)
    results = {}
    for path in ("modules", "executor"):
        monkeypatch.setattr(executor, "ENABLED", path == "executor")
        torch.manual_seed(9)
        model = mink_unet(3, 48, 3, "MinkUNet18A").to(d).train()
        used = []
        real = executor.UNetExecutor._run_forward
        monkeypatch.setattr(executor.UNetExecutor, "_run_forward", lambda self, *a, **k: (used.append(1), real(self, *a, **k))[1])
        out = model(SparseTensor(feats, coords))
        (out * target).sum().backward()
        grads = {n: q.grad.clone() for n, q in model.named_parameters()}
        bufs = {n: b.clone() for n, b in model.named_buffers()}
        model.eval()
        with torch.no_grad():
            ev = model(SparseTensor(feats, coords))
            ft = model.forward_features(SparseTensor(feats, coords)).clone()
        monkeypatch.setattr(executor.UNetExecutor, "_run_forward", real)
        assert bool(used) == (path == "executor")
        results[path] = (out.detach().clone(), ev.clone(), ft, grads, bufs)
    a, b = results["modules"], results["executor"]
    assert torch.equal(a[0], b[0]), "training-mode forward differs"
    assert torch.equal(a[1], b[1]), "evaluation-mode forward differs"
    assert torch.equal(a[2], b[2]), "forward_features differs"
    for n in a[4]:
        assert torch.equal(a[4][n], b[4][n]), n
    worst = max(rel_l2(b[3][n], a[3][n]) for n in a[3])
    assert worst <= 2e-6, "worst parameter-gradient difference between the two host paths: %.3e" % worst


@pytest.mark.parametrize("arch,n_pts,vox", [("MinkUNet18A", 30000, 0.03), ("MinkUNet34C", 120000, 0.02), ("MinkUNet14A", 3000, 0.05)])
def test_inference_bn_epilogue_is_bitwise_the_separate_launches(arch, n_pts, vox, monkeypatch):
    """Round 6: in an inference pass every stage's evaluation-mode batch norm (+ residual, ReLU, cat store) runs in the epilogue of the
    kernel that finishes the stage's convolution (csrc/epilogue.h) -- tile-list (single and split launches), weight-stationary (reduce and
    direct), register-gather, 1x1 and stem kernels.  Same expression, same order: outputs and feature taps bitwise equal to the pass with
    the separate osn_bn_apply2 launches (OSN_NET_RUN_NO_BN_EPILOGUE), which test_executor_equals_the_module_path ties to the module path."""
    from openscene_amd import executor
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    d = dev()
    coords = torch.from_numpy(scene_coords(61, n_pts, vox, batch=2 if n_pts < 100000 else 1)).to(d)
    feats = torch.rand(coords.shape[0], 3, device=d)
    torch.manual_seed(11)
    model = mink_unet(3, 48, 3, arch).to(d).train()
    with torch.no_grad():
        for _ in range(2):
            model(SparseTensor(feats, coords))              # running statistics away from (0, 1)
    model.eval()
    got = {}
    for on in (False, True):
        monkeypatch.setattr(executor, "BN_EPILOGUE", on)
        with torch.no_grad():
            out = model(SparseTensor(feats, coords)).clone()
            ft = model.forward_features(SparseTensor(feats, coords))
            ft = (ft[0] if isinstance(ft, tuple) else ft).clone()
        got[on] = (out, ft)
    assert torch.equal(got[True][0], got[False][0]) and torch.equal(got[True][1], got[False][1])
    assert got[True][0].abs().max().item() > 0 and torch.isfinite(got[True][0]).all()


def test_executor_with_frozen_and_eval_mode_gradients():
    """A frozen parameter gets no gradient; evaluation-mode BN (running statistics) inside a graph that needs gradients
    back-propagates through the executor like the module path does."""
    from openscene_amd import executor
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    d = dev()
    coords = torch.from_numpy(scene_coords(52, 8000, 0.04)).to(d)
    feats = torch.rand(coords.shape[0], 3, device=d)
    got = {}
    for on in (False, True):
        executor.ENABLED = on
        try:
            torch.manual_seed(3)
            model = mink_unet(3, 16, 3, "MinkUNet14A").to(d).eval()
            model.block2[0].conv1.kernel.requires_grad_(False)
            out = model(SparseTensor(feats, coords))
            out.square().mean().backward()
            assert model.block2[0].conv1.kernel.grad is None
            got[on] = {n: q.grad.clone() for n, q in model.named_parameters() if q.grad is not None}
        finally:
            executor.ENABLED = True
    assert got[False].keys() == got[True].keys() and len(got[True]) > 50
    assert max(rel_l2(got[True][n], got[False][n]) for n in got[True]) <= 2e-6


def test_unfused_module_chain_equals_fused():
    """Calling the modules one by one the way models/mink_unet.py:116-174 does (conv, bn, relu,
    `out += residual`) gives the fused result."""
    import openscene_amd.minkowski as ME
    from openscene_amd.sparse import SparseTensor
    torch.manual_seed(2)
    d = dev()
    conv = ME.MinkowskiConvolution(3, 32, kernel_size=5, dimension=3).to(d)
    bn = ME.MinkowskiBatchNorm(32).to(d)
    relu = ME.MinkowskiReLU(inplace=True)
    blk = ME.BasicBlock(32, 32, dimension=3).to(d)
    coords = torch.from_numpy(scene_coords(31, 5000, 0.05)).to(d)
    x = SparseTensor(torch.rand(coords.shape[0], 3, device=d), coords)
    y = relu(bn(conv(x)))
    res = y
    z = blk.conv1(y); z = blk.norm1(z); z = blk.relu(z); z = blk.conv2(z); z = blk.norm2(z)
    z += res
    z = blk.relu(z)
    bn.bn.running_mean.zero_(); bn.bn.running_var.fill_(1)
    fused = blk(y)
    assert rel_l2(z.F, fused.F) < 1e-6


def test_tile_ordered_maps_same_result_and_deterministic():
    """Mask-ordered tiles + unit splitting are scheduling changes only.  The row order of the tiles does
    not change the arithmetic at all (tests/test_gpu_spconv.py checks that bitwise); splitting a tile's
    offsets into units re-associates the fp32 sum over offsets (partials added in a fixed order), so
    against the unordered path the forward agrees to fp32 round-off (gradients: up to ReLU sign flips),
    and the ordered path itself is bitwise reproducible run to run."""
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, SparseTensor
    torch.manual_seed(4)
    model = mink_unet(3, 32, 3, "MinkUNet18A").to(dev())
    coords = torch.from_numpy(scene_coords(41, 30000, 0.03)).to(dev())       # ~25 k voxels: above the sort threshold
    feats = torch.rand(coords.shape[0], 3, device=dev())
    outs = []
    old = CoordinateManager.SORT_MIN_ROWS
    try:
        for min_rows in (10 ** 9, 0, 0):
            CoordinateManager.SORT_MIN_ROWS = min_rows
            model.zero_grad()
            for m in model.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            out = model(SparseTensor(feats, coords))
            out.square().mean().backward()
            outs.append((out.detach().clone(), model.block8[0].conv1.kernel.grad.clone(),
                         model.conv0p1s1.kernel.grad.clone()))
    finally:
        CoordinateManager.SORT_MIN_ROWS = old
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b), "tile-ordered path is not bitwise reproducible"
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 2e-5 * outs[0][0].abs().max().item()
    # gradients additionally see ReLU sign flips of pre-activations that sit within fp32 rounding of zero
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert ((a - b).norm() / a.norm()).item() <= 2e-3


def test_prefetched_maps_give_the_same_step():
    """MapPrefetcher (maps built on a side stream one batch ahead) changes nothing but the timing: same maps,
    bitwise-identical forward output and gradients."""
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, MapPrefetcher, SparseTensor
    d = torch.device("cuda", 0)
    v = syn.shuffled(syn.grid_voxels(syn.room_points(2, n_pts=40000), 0.03), 2)
    coords = torch.from_numpy(syn.batch_coords([v])).to(d)
    feats = torch.rand(coords.shape[0], 3, device=d)
    torch.manual_seed(0)
    model = mink_unet(3, 16, 3, "MinkUNet14A").to(d).train()
    pf = MapPrefetcher(d)
    handles = [pf.submit(coords + s) for s in (0, 0)]             # two batches in flight on the side stream
    cm = pf.take(handles[0])
    ref = CoordinateManager(coords)
    ref.prebuild()
    for key in ((1, 1, 3), (1, 2, 2), (2, 1, 2), (4, 4, 3), (1, 1, 5)):
        a, b = cm.kmap(*key), ref.kmap(*key)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    outs = []
    for x in (SparseTensor(feats, coordinate_manager=cm), SparseTensor(feats, coords),
              SparseTensor(feats, coordinate_manager=pf.take(handles[1]))):
        model.zero_grad(set_to_none=True)
        out = model(x)
        out.square().mean().backward()
        outs.append((out.detach().clone(), model.conv0p1s1.kernel.grad.clone(), model.final.kernel.grad.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_wgrad_on_the_auxiliary_stream_and_cached_weight_images_change_nothing(monkeypatch, modules_only):
    """The backward of a convolution on a small map forks the weight gradient onto an auxiliary stream and joins before
    it returns; weight images of parameters come from a per-device cache refreshed once per optimizer step.  Both are
    pure scheduling: two optimizer steps with them on must be bitwise identical to two steps with them off."""
    from openscene_amd import functional as F_, ops
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    d = torch.device("cuda", 0)
    v = syn.shuffled(syn.grid_voxels(syn.room_points(6, n_pts=30000), 0.03), 6)
    coords = torch.from_numpy(syn.batch_coords([v])).to(d)
    feats = torch.rand(coords.shape[0], 3, device=d)

    def two_steps(overlap_rows, cache):
        monkeypatch.setattr(F_, "WGRAD_OVERLAP_MAX_ROWS", overlap_rows)
        monkeypatch.setattr(ops, "WEIGHT_CACHE", cache)
        ops.clear_weight_cache()
        torch.manual_seed(5)
        model = mink_unet(3, 32, 3, "MinkUNet18A").to(d).train()
        optim = torch.optim.SGD(model.parameters(), lr=0.05)
        for _ in range(2):
            optim.zero_grad(set_to_none=True)
            out = model(SparseTensor(feats, coords))
            out.square().mean().backward()
            optim.step()
        torch.cuda.synchronize(d)
        return [out.detach().clone()] + [p.detach().clone() for p in model.parameters()]

    ref = two_steps(0, False)
    for got in (two_steps(1 << 30, True), two_steps(40000, True), two_steps(1 << 30, False)):
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    ops.clear_weight_cache()


def test_foreign_class_through_the_alias_runs_the_executor_on_the_gpu(monkeypatch):
    """a U-Net class written against `import MinkowskiEngine` (tests/foreign/mink_unet.py: the
    reference's attribute names and un-fused forward) reached through install_minkowski_alias() gets the network executor on
    the HIP path -- output bitwise equal to the mirror's executor, gradients equal -- and with OSN_EXECUTOR=0 its OWN un-fused
    conv / BN / ReLU / ME.cat chain runs on the HIP kernels and agrees to fp32 round-off.  One optimizer step with
    torch.optim.Adam created before .to(device), as run/distill.py:141-152 does."""
    import os
    import sys
    import openscene_amd
    from openscene_amd import drop_in, executor as E
    from openscene_amd.mink_unet import mink_unet
    here = os.path.dirname(os.path.abspath(__file__))
    for name in [m for m in sys.modules if m == "MinkowskiEngine" or m.startswith("MinkowskiEngine.") or m.startswith("foreign")]:
        monkeypatch.delitem(sys.modules, name)
    openscene_amd.install_minkowski_alias()
    monkeypatch.syspath_prepend(here)
    try:
        import foreign.mink_unet as fm
        import MinkowskiEngine as ME
        d = dev()
        coords = torch.from_numpy(scene_coords(77, 30000, 0.03)).to(d)
        feats = torch.rand(coords.shape[0], 3, device=d)
        torch.manual_seed(3)
        theirs = fm.MinkUNet18A(3, 48, 3)
        optimizer = torch.optim.Adam(theirs.parameters(), lr=1e-3)
        theirs = theirs.to(d).train()
        ours = mink_unet(3, 48, 3, "MinkUNet18A").to(d).train()
        ours.load_state_dict(theirs.state_dict())
        assert drop_in.accelerated(theirs) and E.for_model(theirs) is not None
        launches = []
        real = E.UNetExecutor._run_forward
        monkeypatch.setattr(E.UNetExecutor, "_run_forward", lambda self, *a, **k: (launches.append(self), real(self, *a, **k))[1])
        mask = torch.rand(coords.shape[0], device=d) < 0.3
        tgt = torch.nn.functional.normalize(torch.randn(int(mask.sum()), 48, device=d), dim=1)

        def step(model, x):
            out = model(x)
            loss = (1 - torch.nn.CosineSimilarity()(out[mask], tgt)).mean()
            model.zero_grad(set_to_none=True)
            loss.backward()
            return out.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]

        a_out, a_g = step(theirs, ME.SparseTensor(feats, coords))
        assert len(launches) == 1 and launches[0] is E.for_model(theirs)          # the foreign class ran the executor
        b_out, b_g = step(ours, ME.SparseTensor(feats, coords))
        assert torch.equal(a_out, b_out)
        for (n, _), ga, gb in zip(theirs.named_parameters(), a_g, b_g):
            assert torch.equal(ga, gb), n
        # its own forward (executor off): un-fused modules on the HIP kernels, same numbers to round-off
        monkeypatch.setattr(E, "ENABLED", False)
        for m in theirs.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.reset_running_stats()
        c_out, c_g = step(theirs, ME.SparseTensor(feats, coords))
        assert len(launches) == 2                                          # (the mirror's step was the second executor pass; none since)
        assert rel_l2(c_out, a_out) <= 2e-5
        for (n, _), ga, gc in zip(theirs.named_parameters(), a_g, c_g):
            assert rel_l2(gc, ga) <= 3e-3, n                                 # ReLU sign flips at round-off (see the module docstring)
        monkeypatch.setattr(E, "ENABLED", True)
        before = [p.detach().clone() for p in theirs.parameters()]
        step(theirs, ME.SparseTensor(feats, coords))
        optimizer.step()
        assert all(not torch.equal(p, q) for p, q in zip(theirs.parameters(), before))
        # inference through the same class: run/evaluate.py:289-292 as written -- the row gather stays lazy and the matmul is the fused
        # query kernel (openscene_amd/lazy_rows.py); training outputs (above) were plain tensors
        from openscene_amd.lazy_rows import GatheredRows, NetworkOutput
        from openscene_amd.query import query_distill
        assert type(a_out) is torch.Tensor
        theirs.eval()
        inds_reverse = torch.randint(0, coords.shape[0], (70000,), device=d)
        text_features = torch.nn.functional.normalize(torch.randn(20, 48, device=d), dim=1).half()
        with torch.no_grad():
            predictions = theirs(ME.SparseTensor(feats, coords))
            assert type(predictions) is NetworkOutput
            dense = predictions.clone()
            predictions = predictions[inds_reverse, :]
            assert type(predictions) is GatheredRows
            pred = predictions.half() @ text_features.t()
            logits_pred = torch.max(pred, 1)[1].cpu()
            assert predictions._real is None and type(pred) is torch.Tensor
            labels, scores = query_distill(dense, text_features, inds_reverse, return_scores=True)
            assert torch.equal(pred, scores) and torch.equal(logits_pred, labels.cpu())
    finally:
        drop_in.remove_import_hook()
