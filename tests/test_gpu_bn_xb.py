"""The one-launch batch-norm kernels (csrc/bn.hip: bn_xb_fwd_kernel / bn_xb_bwd_kernel behind osn_bn_forward_train3 /
osn_bn_backward_multi3) against the three-launch path they replace and against float64 torch.

Stated tolerance: the two HIP paths use the same formulas and differ only in how the fp64 column sums are associated, so
mean / var / parameter gradients agree to 2e-6 of their largest element and y / gx to 2e-6 as well (bitwise except on a
rounding tie; asserted loosely, reported exactly); vs float64 torch the tolerances of tests/test_gpu_dense.py apply.
Also checked: the in-launch hand-off under uneven load (a second stream saturating the device), slot rotation and the
self-resetting counters over many calls, run-to-run bitwise reproducibility, and the whole network with the kernels on."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)


@pytest.fixture
def xb():
    """Turns the one-launch kernels on for every row count up to 20 000 (forward from 1 row) and off again."""
    from openscene_amd import ops

    class Switch:
        def on(self):
            ops.bn_xb_config(1, 1, 20000)

        def off(self):
            ops.bn_xb_config(0, 0, 16384)

    s = Switch()
    yield s
    s.off()


def case(n, c, nsrc, seed=0):
    g = torch.Generator().manual_seed(1000 * seed + n + c + nsrc)
    d = dev()
    x = (torch.randn(n, c, generator=g) * 1.7 - 0.3).to(d)
    res = torch.randn(n, c, generator=g).to(d)
    gamma = (torch.rand(c, generator=g) + 0.5).to(d)
    beta = (torch.rand(c, generator=g) - 0.5).to(d)
    wide = torch.randn(n, c + 24, generator=g).to(d)                   # one gradient source is columns [8, 8 + c) of this
    srcs = [wide[:, 8:8 + c]] + [torch.randn(n, c, generator=g).to(d) for _ in range(nsrc - 1)]
    return x, res, gamma, beta, srcs


def run_both_directions(x, res, gamma, beta, srcs, relu, with_res, from_x):
    """forward (+ second destination) and backward through the ops wrappers -> every output, in a dict."""
    from openscene_amd import ops
    n, c = x.shape
    d = x.device
    rm, rv = torch.full((c,), 0.25, device=d), torch.full((c,), 0.75, device=d)
    cat = torch.full((n, c + 40), -7.0, device=d)
    y, mean, var = ops.bn_forward_train2(x, gamma, beta, 1e-5, res if with_res else None, relu, rm, rv, 0.1, cat[:, 16:16 + c])
    assert torch.equal(cat[:, 16:16 + c], y) and bool((cat[:, :16] == -7).all()) and bool((cat[:, 16 + c:] == -7).all())
    gx, gres, ggamma, gbeta = ops.bn_backward_multi(x, None if from_x else y, srcs, mean, var, gamma, 1e-5, relu, True, with_res,
                                                    beta=beta if from_x else None)
    out = dict(y=y, mean=mean.clone(), var=var.clone(), rm=rm, rv=rv, gx=gx, ggamma=ggamma, gbeta=gbeta)
    if with_res:
        out["gres"] = gres
    return out


SHAPES = [(1, 32, 1), (37, 64, 2), (383, 4, 1), (385, 68, 3), (700, 256, 3), (730, 512, 1), (3326, 128, 2), (3326, 384, 1),
          (4096, 32, 1), (4097, 96, 3), (12289, 64, 2), (13393, 192, 3), (16384, 128, 1), (20000, 32, 2)]


@pytest.mark.parametrize("n,c,nsrc", SHAPES)
@pytest.mark.parametrize("relu,with_res,from_x", [(False, False, False), (True, False, True), (True, True, False)])
def test_one_launch_batchnorm_equals_the_three_launch_path_and_torch(xb, n, c, nsrc, relu, with_res, from_x):
    from openscene_amd import ops
    x, res, gamma, beta, srcs = case(n, c, nsrc)
    xb.off()
    ref = run_both_directions(x, res, gamma, beta, srcs, relu, with_res, from_x)
    xb.on()
    got = run_both_directions(x, res, gamma, beta, srcs, relu, with_res, from_x)
    ops.bn_sync_check(dev())
    state = ops.bn_sync(dev())
    assert int(state[:1024].abs().sum()) == 0, "counters or the error word are not back at zero"
    worst = max((rel(got[k], ref[k]), k) for k in ref)
    assert worst[0] <= 2e-6, "one-launch vs three-launch: %s differs by %.3e" % (worst[1], worst[0])
    # float64 torch on the same inputs
    bn = torch.nn.BatchNorm1d(c).double()
    bn.weight.data.copy_(gamma.cpu()); bn.bias.data.copy_(beta.cpu())
    bn.running_mean.fill_(0.25); bn.running_var.fill_(0.75)
    x64 = x.double().cpu().requires_grad_(True)
    r64 = res.double().cpu().requires_grad_(True)
    if n > 1:
        yr = bn(x64) + (r64 if with_res else 0)
        yr = torch.relu(yr) if relu else yr
        yr.backward(sum(t.double().cpu() for t in srcs))
        assert rel(got["y"], yr) < 2e-5
        assert rel(got["rm"], bn.running_mean) < 1e-5 and rel(got["rv"], bn.running_var) < 1e-5
        assert rel(got["gx"], x64.grad) < 5e-5
        assert rel(got["ggamma"], bn.weight.grad) < 5e-5 and rel(got["gbeta"], bn.bias.grad) < 5e-5
        if with_res:
            assert rel(got["gres"], r64.grad) < 2e-6


def test_one_launch_batchnorm_is_reproducible_over_many_calls_and_slots(xb):
    """200 calls of alternating shapes walk the 16 state slots a dozen times: every repeat of a shape returns the same bits."""
    from openscene_amd import ops
    xb.on()
    cases = [case(n, c, 2, seed=3) for n, c in ((13393, 64), (3326, 256), (730, 512), (5000, 96))]
    first = [None] * len(cases)
    for it in range(50):
        for i, (x, res, gamma, beta, srcs) in enumerate(cases):
            out = run_both_directions(x, res, gamma, beta, srcs, True, True, False)
            if first[i] is None:
                first[i] = out
            else:
                for k in out:
                    assert torch.equal(out[k], first[i][k]), (it, i, k)
    ops.bn_sync_check(dev())
    assert int(ops.bn_sync(dev())[:1024].abs().sum()) == 0


def test_one_launch_batchnorm_under_uneven_load(xb):
    """The hand-off between workgroups must not depend on placement or timing: the same call while a second stream keeps the
    device busy with large matrix products (workgroups of the batch norm start late and unevenly) returns the same bits."""
    from openscene_amd import ops
    xb.on()
    d = dev()
    x, res, gamma, beta, srcs = case(13393, 192, 3, seed=5)
    quiet = run_both_directions(x, res, gamma, beta, srcs, True, True, False)
    torch.cuda.synchronize()
    a = torch.randn(4096, 4096, device=d)
    side = torch.cuda.Stream(d)
    for rep in range(10):
        with torch.cuda.stream(side):
            for _ in range(6):
                a = torch.tanh(a @ a * 1e-2)
        for _ in range(8):
            out = run_both_directions(x, res, gamma, beta, srcs, True, True, False)
            for k in out:
                assert torch.equal(out[k], quiet[k]), (rep, k)
        torch.cuda.synchronize()
    ops.bn_sync_check(d)
    assert int(ops.bn_sync(d)[:1024].abs().sum()) == 0


@pytest.mark.parametrize("path", ["executor", "modules"])
def test_network_with_one_launch_batchnorm(xb, path, monkeypatch):
    """MinkUNet18A training step (two small scenes: every level is below the 20 000-row limit set here) with the kernels
    on vs off: output, loss gradient of every parameter, running statistics."""
    from openscene_amd import executor, ops, synthetic as syn
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    monkeypatch.setattr(executor, "ENABLED", path == "executor")
    d = dev()
    coords = syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(21 + b, n_pts=9000), 0.04), 21 + b) for b in range(2)])
    feats = torch.rand(coords.shape[0], 3).to(d)
    c4 = torch.from_numpy(coords).to(d)
    res = {}
    for mode in ("off", "on"):
        getattr(xb, mode)()
        torch.manual_seed(4)
        model = mink_unet(3, 32, 3, "MinkUNet18A").to(d).train()
        out = model(SparseTensor(feats, c4))
        target = torch.randn(out.shape, generator=torch.Generator().manual_seed(2)).to(d)
        (out * target).sum().backward()
        res[mode] = (out.detach(), {k: v.grad.clone() for k, v in model.named_parameters()},
                     {k: v.clone() for k, v in model.named_buffers() if "running" in k})
    ops.bn_sync_check(d)
    assert rel(res["on"][0], res["off"][0]) <= 1e-5
    worst = max((rel(res["on"][1][k], res["off"][1][k]), k) for k in res["off"][1])
    assert worst[0] <= 1e-4, worst
    worst = max((rel(res["on"][2][k], res["off"][2][k]), k) for k in res["off"][2])
    assert worst[0] <= 1e-5, worst
