"""HIP batch-norm (+ReLU +residual), open-vocabulary query and voxeliser vs their
oracles.  Tolerances are stated next to each check."""
import os

import numpy as np
import pytest
import torch

from oracle import query as oq
from oracle import voxelize as ov

import cpu_backend

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return (got - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)


# ------------------------------------------------------------------ the executor's batch-norm entry points
@pytest.mark.parametrize("n,c,nsrc", [(700, 256, 3), (3052, 128, 2), (4096, 32, 1), (4097, 32, 3), (47618, 96, 3), (20000, 64, 2)])
def test_batchnorm_summed_gradient_sources_and_second_destination(n, c, nsrc):
    """osn_bn_forward_train2 (the result also lands in a column window of a wider matrix: ME.cat in place) and
    osn_bn_backward_multi (incoming gradient = sum of up to three matrices, column windows included), on both sides of
    the 4096-row switch between the single-launch and the three-launch kernels, vs float64 torch."""
    from openscene_amd import ops
    g = torch.Generator().manual_seed(n + c + nsrc)
    x = torch.randn(n, c, generator=g) * 1.7 - 0.3
    res = torch.randn(n, c, generator=g)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.rand(c, generator=g) - 0.5
    wide = torch.randn(n, c + 24, generator=g)                        # one source is columns [8, 8 + c) of this
    srcs = [wide[:, 8:8 + c]] + [torch.randn(n, c, generator=g) for _ in range(nsrc - 1)]
    d = dev()
    bn = torch.nn.BatchNorm1d(c).double()
    bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
    x64 = x.double().requires_grad_(True)
    r64 = res.double().requires_grad_(True)
    y_ref = torch.relu(bn(x64) + r64)
    y_ref.backward(sum(t.double() for t in srcs))
    rm = torch.zeros(c, device=d); rv = torch.ones(c, device=d)
    cat = torch.full((n, c + 40), -7.0, device=d)
    y, mean, var = ops.bn_forward_train2(x.to(d), gamma.to(d), beta.to(d), 1e-5, res.to(d), True, rm, rv, 0.1, cat[:, 16:16 + c])
    assert rel(y, y_ref) < 2e-5
    assert torch.equal(cat[:, 16:16 + c], y) and bool((cat[:, :16] == -7).all()) and bool((cat[:, 16 + c:] == -7).all())
    assert rel(rm, bn.running_mean) < 1e-5 and rel(rv, bn.running_var) < 1e-5
    wide_d = wide.to(d)
    gsrc = [wide_d[:, 8:8 + c]] + [t.to(d) for t in srcs[1:]]
    gx, gres, ggamma, gbeta = ops.bn_backward_multi(x.to(d), y, gsrc, mean, var, gamma.to(d), 1e-5, True, True, True)
    assert rel(gx, x64.grad) < 5e-5 and rel(gres, r64.grad) < 2e-6
    assert rel(ggamma, bn.weight.grad) < 5e-5 and rel(gbeta, bn.bias.grad) < 5e-5
    # one source, contiguous: bitwise the plain entry point
    one = sum(t for t in srcs).to(d).contiguous()
    a = ops.bn_backward(x.to(d), y, one, mean, var, gamma.to(d), 1e-5, True, True, True)
    b = ops.bn_backward_multi(x.to(d), y, [one], mean, var, gamma.to(d), 1e-5, True, True, True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)


# ------------------------------------------------------------------ elementwise (a11)
@pytest.mark.parametrize("n,ca,cb", [(1, 4, 4), (37, 32, 96), (3052, 128, 128), (100999, 96, 32), (700, 8, 256)])
def test_relu_add_cat_kernels(n, ca, cb):
    """ME.MinkowskiReLU, the un-fused residual add and ME.cat (models/mink_unet.py:114,147-171) as HIP kernels,
    forward and backward, against torch on the CPU: copies and max/add of fp32 values are EXACT."""
    from openscene_amd import functional as F_
    g = torch.Generator().manual_seed(n + ca)
    a = torch.randn(n, ca, generator=g)
    b = torch.randn(n, cb, generator=g)
    a2 = torch.randn(n, ca, generator=g)
    gcat = torch.randn(n, ca + cb, generator=g)
    ga = torch.randn(n, ca, generator=g)
    ar, br, a2r = (t.clone().requires_grad_(True) for t in (a, b, a2))
    ref = torch.cat([torch.relu(ar) + a2r, br], 1)
    ref.backward(gcat)
    ad, bd, a2d = (t.to(dev()).requires_grad_(True) for t in (a, b, a2))
    out = F_.cat([F_.add(F_.relu(ad), a2d), bd])
    out.backward(gcat.to(dev()))
    assert torch.equal(out.cpu(), ref)
    for got, want in ((ad, ar), (bd, br), (a2d, a2r)):
        assert torch.equal(got.grad.cpu(), want.grad)
    # odd element counts (tail path of the elementwise kernel) through the raw ops
    from openscene_amd import ops
    v = torch.randn(n * ca + 3, generator=g)
    w = torch.randn(n * ca + 3, generator=g)
    assert torch.equal(ops.relu_fwd(v.to(dev())).cpu(), torch.relu(v))
    assert torch.equal(ops.add(v.to(dev()), w.to(dev())).cpu(), v + w)
    assert torch.equal(ops.relu_bwd(v.to(dev()), w.to(dev())).cpu(), w * (v > 0))
    del ga


def test_sparse_tensor_cat_and_iadd_run_on_the_hip_kernels():
    """`ME.cat(a, b)` and `out += residual` on SparseTensors (sparse.py) go through the elementwise entry points."""
    from openscene_amd import ops
    from openscene_amd.sparse import SparseTensor, cat
    calls = []
    real_cat, real_add = ops.cat2, ops.add
    ops.cat2 = lambda *a: (calls.append("cat2"), real_cat(*a))[1]
    ops.add = lambda *a: (calls.append("add"), real_add(*a))[1]
    try:
        c = torch.tensor([[0, 0, 0, 0], [0, 1, 0, 0], [0, 5, 5, 5]], dtype=torch.int32, device=dev())
        x = SparseTensor(torch.randn(3, 8, device=dev()), c)
        y = x._like(torch.randn(3, 4, device=dev()))
        z = cat(x, y)
        assert z.F.shape == (3, 12) and torch.equal(z.F[:, :8], x.F) and torch.equal(z.F[:, 8:], y.F)
        w = x._like(x.F.clone())
        w += x
        assert torch.equal(w.F, 2 * x.F)
    finally:
        ops.cat2, ops.add = real_cat, real_add
    assert calls == ["cat2", "add"]



def test_elementwise_inputs_the_hip_kernels_do_not_take_go_to_torch_on_the_device():
    """MinkowskiEngine's `+`, ME.cat and MinkowskiReLU accept any dtype / width; the HIP kernels take contiguous
    16-byte-aligned fp32 (cat: widths in multiples of 4).  Everything else stays on the device through torch's operators
    instead of raising; NaN propagates through the HIP ReLU as through torch.relu; host tensors are still refused."""
    from openscene_amd import _lib, functional as F_
    g = torch.Generator().manual_seed(3)
    a = torch.randn(50, 6, generator=g).to(dev())
    b = torch.randn(50, 3, generator=g).to(dev())
    assert torch.equal(F_.cat([a, b]), torch.cat([a, b], 1))                       # widths 6 and 3
    h = torch.randn(50, 8, generator=g).half().to(dev())
    assert torch.equal(F_.relu(h), torch.relu(h)) and F_.relu(h).dtype == torch.float16
    assert torch.equal(F_.add(h, h), h + h)
    wide = torch.randn(50, 9, generator=g).to(dev())
    view = wide[:, 1:]                                                           # strided, 4-byte-aligned view
    assert torch.equal(F_.relu(view), torch.relu(view)) and torch.equal(F_.add(view, view), view + view)
    x = torch.tensor([[float("nan"), -1.0, 2.0, float("inf")]], device=dev())
    y = F_.relu(x)
    assert torch.isnan(y[0, 0]) and y[0, 1] == 0 and y[0, 2] == 2 and torch.isinf(y[0, 3])
    xr = x.clone().requires_grad_(True)
    F_.relu(xr).backward(torch.ones_like(x))
    assert xr.grad.tolist() == [[0.0, 0.0, 1.0, 1.0]]
    with pytest.raises(_lib.OpenSceneAmdError):
        F_.relu(torch.ones(4, 4))
    with pytest.raises(_lib.OpenSceneAmdError):
        F_.add(torch.ones(4, 4), torch.ones(4, 4))


# ------------------------------------------------------------------ batch norm
@pytest.mark.parametrize("n,c", [(1, 32), (7, 64), (700, 256), (3052, 128), (47618, 96), (100999, 32), (5000, 768)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batchnorm_train_forward_backward(n, c, relu, res):
    from openscene_amd import functional as F_
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, c, generator=g) * 2.5 + 0.7
    r = torch.randn(n, c, generator=g) if res else None
    gy = torch.randn(n, c, generator=g)
    bn_ref = torch.nn.BatchNorm1d(c).double()
    bn_ref.weight.data.uniform_(0.5, 1.5, generator=g)
    bn_ref.bias.data.uniform_(-0.5, 0.5, generator=g)
    bn = torch.nn.BatchNorm1d(c)
    bn.load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in bn_ref.state_dict().items()})
    bn = bn.to(dev())
    if n == 1:
        bn_ref.eval(); bn.eval()                     # torch refuses batch statistics of one row
    x64 = x.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if res else None
    y_ref = bn_ref(x64)
    if res:
        y_ref = y_ref + r64
    if relu:
        y_ref = torch.relu(y_ref)
    y_ref.backward(gy.double())

    xg = x.to(dev()).requires_grad_(True)
    rg = r.to(dev()).requires_grad_(True) if res else None
    y = F_.batch_norm_act(xg, bn, residual=rg, relu=relu)
    y.backward(gy.to(dev()))
    # fp32 elementwise math vs float64: 2e-5 relative to the tensor's max
    assert rel(y, y_ref) < 2e-5
    assert rel(xg.grad, x64.grad) < 5e-5
    assert rel(bn.weight.grad, bn_ref.weight.grad) < 5e-5
    assert rel(bn.bias.grad, bn_ref.bias.grad) < 5e-5
    if res:
        assert rel(rg.grad, r64.grad) < 1e-6
    # running statistics: 1e-5 relative
    assert rel(bn.running_mean, bn_ref.running_mean) < 1e-5
    assert rel(bn.running_var, bn_ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize("n,c", [(37, 32), (3052, 128), (47618, 96), (100999, 32)])
@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_backward_recomputes_the_relu_mask_from_x(n, c, training):
    """Round 4: for bn -> relu without a residual the backward kernels rebuild (y > 0) from x with the forward pass's own
    expression instead of reading y (one tensor less per pass).  Bitwise the same gradients as with y, train and eval mode,
    including rows whose pre-activation is exactly zero or a denormal away from it."""
    from openscene_amd import ops
    g = torch.Generator().manual_seed(n * 7 + c)
    x = (torch.randn(n, c, generator=g) * 1.5 + 0.2).to(dev())
    gamma = torch.empty(c).uniform_(0.5, 1.5, generator=g).to(dev())
    beta = torch.empty(c).uniform_(-0.5, 0.5, generator=g).to(dev())
    rm = (torch.randn(c, generator=g) * 0.1).to(dev())
    rv = torch.empty(c).uniform_(0.5, 2.0, generator=g).to(dev())
    if training:
        y, mean, var = ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm.clone(), rv.clone(), 0.1)
    else:
        mean, var = rm, rv
        y = ops.bn_apply(x, mean, var, gamma, beta, 1e-5, None, True)
    gy = torch.randn(n, c, generator=g).to(dev())
    a = ops.bn_backward(x, y, gy, mean, var, gamma, 1e-5, True, training, False)
    b = ops.bn_backward(x, None, gy, mean, var, gamma, 1e-5, True, training, False, beta=beta)
    for u, v in zip(a, b):
        assert (u is None and v is None) or torch.equal(u, v)
    assert 0.2 < float((y > 0).float().mean()) < 0.8
    with pytest.raises(ValueError):
        ops.bn_backward(x, None, gy, mean, var, gamma, 1e-5, True, training, True, beta=beta)       # a residual needs y


def test_batchnorm_eval_mode():
    from openscene_amd import functional as F_
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2000, 96, generator=g)
    bn_ref = torch.nn.BatchNorm1d(96).double().eval()
    bn_ref.running_mean.uniform_(-1, 1, generator=g)
    bn_ref.running_var.uniform_(0.3, 2, generator=g)
    bn = torch.nn.BatchNorm1d(96)
    bn.load_state_dict({k: (v.float() if v.dtype.is_floating_point else v) for k, v in bn_ref.state_dict().items()})
    bn = bn.to(dev()).eval()
    x64 = x.double().requires_grad_(True)
    y_ref = torch.relu(bn_ref(x64))
    y_ref.sum().backward()
    xg = x.to(dev()).requires_grad_(True)
    y = F_.batch_norm_act(xg, bn, relu=True)
    y.sum().backward()
    assert rel(y, y_ref) < 2e-5 and rel(xg.grad, x64.grad) < 2e-5
    assert torch.equal(bn.running_mean.cpu().double(), bn_ref.running_mean.float().double())


# ----------------------------------------------------------------------- query
def _query_inputs(n_vox, n_pts, d, c, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_vox, d, generator=g)
    x = x / x.norm(dim=1, keepdim=True) * (0.5 + 1.5 * torch.rand(n_vox, 1, generator=g))
    t = torch.randn(c, d, generator=g)
    t = (t / t.norm(dim=1, keepdim=True)).half()
    gather = torch.randint(0, n_vox, (n_pts,), generator=g)
    return x, t, gather


@pytest.mark.parametrize("n,c,n_pts", [(5000, 20, 12000), (300, 160, 0), (1, 21, 7), (4000, 3, 4000)])
def test_rows_argmax_with_gather_equals_torch(n, c, n_pts):
    """ops.rows_argmax = torch's argmax (first maximum) + the point -> voxel gather of run/evaluate.py:290-292, on a column
    slice of a wider matrix (the fused-head scores are padded to a multiple of four columns), ties and -inf rows included."""
    from openscene_amd import ops
    dv = dev()
    g = torch.Generator().manual_seed(n + c)
    wide = torch.randn(n, c + 3, generator=g)
    wide[::7, : c] = wide[::7, : c].round()                 # ties
    if n > 10:
        wide[3, :c] = float("-inf")
        wide[5, :c] = 1.25                                   # a constant row: label 0
    scores = wide.to(dv)[:, :c]
    inds = torch.randint(0, n, (n_pts,), generator=g).to(dv) if n_pts else None
    got = ops.rows_argmax(scores, inds)
    want = scores.argmax(1)
    want = want if inds is None else want[inds]
    assert got.dtype == torch.int64 and torch.equal(got, want)


@pytest.mark.parametrize("flat_grads", [False, True])
@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_flat_adam_equals_torch_adam(flat_grads, weight_decay):
    """openscene_amd.optim.FlatAdam (one launch over a flat buffer) against torch.optim.Adam on the same gradients for five
    steps: parameters within 1e-6 relative of torch's, moments too; gradients given as separate tensors (gathered) and as
    slices of one flat buffer in the optimizer's layout (read in place, as the network executor delivers them); the
    checkpoint layout is torch's."""
    from openscene_amd.optim import FlatAdam
    dv = dev()
    g = torch.Generator().manual_seed(5)
    shapes = [(27, 32, 32), (96,), (96, 768), (3, 5), (1,)]          # odd sizes: every slice still starts 16-byte aligned
    init = [torch.randn(s, generator=g) for s in shapes]
    ref_p = [torch.nn.Parameter(t.clone().to(dv)) for t in init]
    our_p = [torch.nn.Parameter(t.clone().to(dv)) for t in init]
    ref = torch.optim.Adam(ref_p, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    ours = FlatAdam(our_p, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=weight_decay)
    assert all(torch.equal(a.detach(), b.detach()) for a, b in zip(ref_p, our_p))          # flattening kept the values
    for step in range(5):
        grads = [torch.randn(s, generator=g) * (10.0 ** (step - 2)) for s in shapes]
        flat = torch.zeros(ours.total, device=dv)
        for p, q, gr, o in zip(ref_p, our_p, grads, ours.offsets):
            p.grad = gr.to(dv)
            if flat_grads:
                v = flat[o:o + gr.numel()].view(gr.shape)
                v.copy_(gr)
                q.grad = v
            else:
                q.grad = gr.to(dv)
        assert ours._flat_grads()[1] == flat_grads
        versions = [q._version for q in our_p]
        ref.step()
        ours.step()
        assert all(q._version > v for q, v in zip(our_p, versions))
        for p, q in zip(ref_p, our_p):
            assert torch.allclose(q.detach(), p.detach(), rtol=2e-6, atol=1e-7), (step, (q - p).abs().max().item())
    sd_r, sd_o = ref.state_dict(), ours.state_dict()
    for i in range(len(shapes)):
        for key in ("exp_avg", "exp_avg_sq"):          # (the first moment cancels: absolute bound at fp32 resolution of its largest element)
            r_ = sd_r["state"][i][key]
            assert torch.allclose(sd_o["state"][i][key], r_, rtol=2e-6, atol=2e-6 * float(r_.abs().max())), (i, key)
        assert float(sd_o["state"][i]["step"]) == float(sd_r["state"][i]["step"]) == 5.0
    # torch's checkpoint loads into the flat optimizer
    again = FlatAdam([torch.nn.Parameter(t.clone().to(dv)) for t in init], lr=1.0)
    again.load_state_dict(sd_r)
    assert again.steps == 5 and again.param_groups[0]["lr"] == 3e-3
    assert torch.allclose(again.state_dict()["state"][2]["exp_avg"], sd_r["state"][2]["exp_avg"])


@pytest.mark.parametrize("loss_type", ["cosine", "l1"])
@pytest.mark.parametrize("n,n_sel,d", [(5000, 1200, 768), (777, 777, 512), (64, 1, 20)])
def test_distill_loss_forward_and_gradient(n, n_sel, d, loss_type):
    """a14: loss of run/distill.py:322-328 on output[sel] and its gradient with respect to the full output, against the
    float64 oracle (itself pinned to torch's autograd of those lines): loss within 1e-6 relative, gradient rows within
    2e-6 of the row's largest element, rows outside the selection exactly zero; the upstream gradient scales it."""
    from oracle import loss as ol
    from openscene_amd.losses import distill_loss
    g = torch.Generator().manual_seed(n + d)
    out = torch.randn(n, d, generator=g) * 2.5
    sel = torch.randperm(n, generator=g)[:n_sel].sort().values
    target = torch.nn.functional.normalize(torch.randn(n_sel, d, generator=g), dim=1).half().float()
    ref_loss, ref_grad = ol.distill_loss(out.numpy(), sel.numpy(), target.numpy(), loss_type)
    dv = dev()
    x = out.to(dv).requires_grad_(True)
    loss = distill_loss(x, sel.to(dv), target.to(dv), loss_type, validate=True)
    (loss * 2.5).backward()
    assert abs(loss.item() - ref_loss) <= 1e-6 * max(1.0, abs(ref_loss))
    got = x.grad.cpu().double().numpy() / 2.5
    scale = np.abs(ref_grad).max(axis=1, keepdims=True)
    assert (np.abs(got - ref_grad) <= 2e-6 * scale + 1e-30).all()
    keep = np.zeros(n, dtype=bool)
    keep[sel.numpy()] = True
    assert not got[~keep].any()
    # the bool mask itself is accepted (resolved with a host synchronisation)
    mask = torch.from_numpy(keep).to(dv)
    loss2 = distill_loss(x.detach(), mask, target.to(dv), loss_type)
    assert loss2.item() == loss.item()


def test_distill_loss_rejects_bad_selections():
    from openscene_amd._lib import OpenSceneAmdError
    from openscene_amd.losses import distill_loss
    dv = dev()
    out = torch.randn(100, 32, device=dv)
    tgt = torch.randn(3, 32, device=dv)
    with pytest.raises(OpenSceneAmdError):
        distill_loss(out, torch.tensor([1, 5, 5], device=dv), tgt, validate=True)          # a row twice
    with pytest.raises(OpenSceneAmdError):
        distill_loss(out, torch.tensor([1, 5, 100], device=dv), tgt, validate=True)        # outside the output
    with pytest.raises(ValueError):
        distill_loss(out, torch.tensor([1, 5], device=dv), tgt)
    with pytest.raises(ValueError):
        distill_loss(out, torch.tensor([1, 5, 7], device=dv), tgt, loss_type="l2")


@pytest.mark.parametrize("n_vox,n_pts,d,c", [(4000, 6001, 768, 20), (3000, 3000, 512, 21), (2500, 4000, 768, 160),
                                             (1000, 1500, 768, 43), (700, 900, 512, 300), (130, 1, 768, 2),
                                             # column-split kernel (> 64 labels): two tiles per wave, ragged last feature
                                             # chunk; exactly one full group; one label more than the narrow kernel takes
                                             (2000, 3000, 520, 100), (1500, 2500, 768, 192), (900, 1300, 512, 65)])
def test_query_scores_and_argmax(n_vox, n_pts, d, c):
    from openscene_amd import ops
    x, t, gather = _query_inputs(n_vox, n_pts, d, c, n_vox + c)
    ref_scores, ref_arg = oq.query(x, t, gather)
    scores, arg = ops.cosine_query(x.to(dev()), t.to(dev()), gather.to(dev()))
    scores, arg = scores.cpu(), arg.cpu()
    # fp16 outputs of O(1) magnitude, fp32 accumulation in a different order: 2e-3 absolute
    assert (scores.float() - ref_scores.float()).abs().max().item() <= 2e-3
    # the label must be a maximiser of OUR rounded scores, lowest index on ties ...
    first_max = (scores == scores.max(1, keepdim=True)[0]).float().argmax(1)
    assert torch.equal(arg, first_max)
    # ... and agree with the reference wherever the reference's top-2 margin exceeds the score tolerance
    top2 = ref_scores.float().topk(min(2, c), dim=1)[0]
    clear = (top2[:, 0] - top2[:, -1]) > 4e-3 if c > 1 else torch.ones(n_pts, dtype=torch.bool)
    assert torch.equal(arg[clear], ref_arg[clear])
    # no gather = identity
    s2, a2 = ops.cosine_query(x.to(dev()), t.to(dev()), None, want_scores=False)
    assert s2 is None and a2.shape[0] == n_vox


@pytest.mark.parametrize("n_vox,n_pts,d,c", [(3000, 4097, 768, 160), (5000, 10001, 768, 65), (4000, 8200, 512, 96),
                                             (6000, 9000, 768, 128), (2000, 4096, 512, 160)])
def test_query_many_labels_persistent_kernel(n_vox, n_pts, d, c):
    """Round 5: 65 .. 160 labels at the CLIP widths and >= 4096 points take query_wide_kernel (text matrix resident in the consumer
    waves' registers, producer waves streaming point rows; argmax on DPP keys).  Same contract as the narrow kernel: scores within
    2e-3 of the reference expression, label = the FIRST maximiser of our own rounded scores (torch.max) -- checked on a text matrix
    with duplicated rows, where exact ties are guaranteed -- and equal to the reference's off ties; ragged last tile, labels-only."""
    from openscene_amd import ops
    x, t, gather = _query_inputs(n_vox, n_pts, d, c, n_vox + c)
    t[c - 1] = t[3]                                       # two exact ties per point: the lower label has to win
    t[c // 2 + 1] = t[c // 2]
    ref_scores, ref_arg = oq.query(x, t, gather)
    scores, arg = ops.cosine_query(x.to(dev()), t.to(dev()), gather.to(dev()))
    scores, arg = scores.cpu(), arg.cpu()
    assert (scores.float() - ref_scores.float()).abs().max().item() <= 2e-3
    assert torch.equal(scores[:, c - 1], scores[:, 3]) and torch.equal(scores[:, c // 2 + 1], scores[:, c // 2])
    first_max = (scores == scores.max(1, keepdim=True)[0]).float().argmax(1)
    assert torch.equal(arg, first_max)
    assert not bool((arg == c - 1).any()) and not bool((arg == c // 2 + 1).any())
    top3 = ref_scores.float().topk(3, dim=1)[0]
    clear = (top3[:, 0] - top3[:, 2]) > 4e-3              # (third best: a duplicated row makes top-1 == top-2 exactly, in both engines)
    assert clear.float().mean().item() > 0.3
    assert torch.equal(arg[clear], ref_arg[clear])
    s2, a2 = ops.cosine_query(x.to(dev()), t.to(dev()), gather.to(dev()), want_scores=False)
    assert s2 is None and torch.equal(a2.cpu(), arg)


@pytest.mark.parametrize("n_pts,d,c", [(5000, 768, 160), (4200, 512, 96), (3000, 768, 20)])
def test_query_signed_zero_scores_tie_like_torch_max(n_pts, d, c):
    """scores that round to +0.0 and -0.0 in fp16 are EQUAL for torch.max (run/evaluate.py:292: the lowest label wins);
    the persistent kernel's argmax orders raw fp16 bits and used to rank +0 above -0.  One tiny feature against text rows of
    alternating sign makes every fp16 score a signed zero, label 0's negative: the label must be 0 everywhere.  A second set adds one
    clearly positive label per point."""
    from openscene_amd import ops
    g = torch.Generator().manual_seed(5)
    t = torch.zeros(c, d)
    t[:, 0] = 0.01 * torch.tensor([-1.0 if j % 2 == 0 else 1.0 for j in range(c)])
    t = t.half()
    x = torch.zeros(n_pts, d)
    x[:, 0] = 1e-6 * (1 + torch.rand(n_pts, generator=g))                    # products of 1e-8: below half the smallest fp16 subnormal
    ref_scores, ref_arg = oq.query(x, t, None)
    assert bool((ref_scores == 0).all()) and bool(torch.signbit(ref_scores.float()[:, 0]).all()) and int(ref_arg.max()) == 0
    scores, arg = ops.cosine_query(x.to(dev()), t.to(dev()), None)
    sc = scores.cpu().float()
    assert bool((sc == 0).all()) and bool(torch.signbit(sc[:, 0]).all()) and not bool(torch.signbit(sc[:, 1]).any())
    assert int(arg.abs().max()) == 0, "signed zeros must tie: label 0 everywhere"
    _, arg2 = ops.cosine_query(x.to(dev()), t.to(dev()), None, want_scores=False)
    assert int(arg2.abs().max()) == 0
    # one label with a real (positive) score per point among the signed zeros: that label wins wherever it is
    t2 = t.clone().float()
    t2[:, 1:] = torch.nn.functional.normalize(torch.randn(c, d - 1, generator=g), dim=1)
    t2 = t2.half()
    want = torch.randint(0, c, (n_pts,), generator=g)
    x2 = x.clone()
    x2[:, 1:] = t2.float()[want][:, 1:]
    ref_scores, ref_arg = oq.query(x2, t2, None)
    _, arg3 = ops.cosine_query(x2.to(dev()), t2.to(dev()), None)
    top2 = ref_scores.float().topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 4e-3
    assert clear.float().mean().item() > 0.9
    assert torch.equal(arg3.cpu()[clear], ref_arg[clear])


def test_query_ensemble_many_labels():
    """The three-pass ensemble (row norms, per-source best score, per-point source selection: run/evaluate.py:302-324) through the
    persistent kernel's rowdiv / rowmax / sel paths (160 labels, 6000 points)."""
    from openscene_amd import ops
    xd, t, gather = _query_inputs(3000, 6000, 768, 160, 11)
    xf, _, _ = _query_inputs(3000, 6000, 768, 160, 12)
    ref_scores, ref_arg, ref_sel = cpu_backend.query_ensemble(xd, xf, t, gather, gather)
    scores, arg, sel = ops.query_ensemble(xd.to(dev()), xf.to(dev()), t.to(dev()), gather.to(dev()), gather.to(dev()))
    fd, ff = xd[gather], xf[gather]
    pd = oq.half_matmul((fd / (fd.norm(dim=-1, keepdim=True) + 1e-5)).half(), t).float().max(1)[0]
    pf = oq.half_matmul((ff / (ff.norm(dim=-1, keepdim=True) + 1e-5)).half(), t).float().max(1)[0]
    clear = (pd - pf).abs() > 4e-3
    assert torch.equal(sel.cpu()[clear], ref_sel[clear])
    same = sel.cpu() == ref_sel
    assert (scores.cpu().float()[same] - ref_scores.float()[same]).abs().max().item() <= 2e-3
    assert same.float().mean().item() > 0.95
    first_max = (scores.cpu() == scores.cpu().max(1, keepdim=True)[0]).float().argmax(1)
    assert torch.equal(arg.cpu(), first_max)


def test_fused_head_query_matches_the_unfused_path():
    """labels from features96 @ (W_final @ text^T) equal the labels of the reference expression
    (model output [inds_reverse].half() @ text^T, run/evaluate.py:288-292) wherever the reference's top-2 margin exceeds
    the fp16 rounding it applies to the 768-d vector; scores within that rounding (stated: 1e-3 * max(1, max|score|))."""
    from openscene_amd.disnet import DisNet
    from openscene_amd.query import query_distill_fused
    from openscene_amd.sparse import SparseTensor
    from openscene_amd import synthetic as syn

    class Cfg:
        arch_3d = "MinkUNet14A"
        feature_2d_extractor = "openseg"
    torch.manual_seed(3)
    net = DisNet(Cfg()).to(dev()).eval()
    coords = syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(5, n_pts=9000), 0.05), 5)])
    c = torch.from_numpy(coords).to(dev())
    feats = torch.rand(coords.shape[0], 3, device=dev())
    g = torch.Generator().manual_seed(1)
    inds_reverse = torch.randint(0, coords.shape[0], (12000,), generator=g)
    for n_labels in (20, 21, 160):
        text = torch.nn.functional.normalize(torch.randn(n_labels, 768, generator=g), dim=1).half()
        with torch.no_grad():
            pred = net(SparseTensor(feats, c))
            f96, w_final = net.forward_features(SparseTensor(feats, c))
            labels, scores = query_distill_fused(f96, w_final, text.to(dev()), inds_reverse.to(dev()), return_scores=True)
        assert f96.shape == (coords.shape[0], 96) and scores.shape == (coords.shape[0], n_labels)
        ref_scores, ref_labels = oq.query(pred.cpu(), text, inds_reverse)
        ref_scores = ref_scores.float()
        tol = 1e-3 * max(1.0, ref_scores.abs().max().item())
        assert (scores.cpu()[inds_reverse] - ref_scores).abs().max().item() <= tol
        top2 = ref_scores.topk(2, dim=1)[0]
        clear = (top2[:, 0] - top2[:, 1]) > 2 * tol
        assert clear.float().mean().item() > 0.5
        assert torch.equal(labels.cpu()[clear], ref_labels[clear])


def test_fused_head_ensemble_query_matches_the_reference_expression():
    """SURVEY.md 8(f) row 2, ensemble mode: selection and labels from 96-d quantities (W text^T and the Gram matrix W W^T)
    against run/evaluate.py:302-324 evaluated on the expanded 768-d features (oracle/query.py)."""
    from openscene_amd.query import query_ensemble_fused
    g = torch.Generator().manual_seed(11)
    n_vox, n_pts, d = 7000, 11000, 768
    f96 = torch.randn(n_vox, 96, generator=g) * (0.5 + torch.rand(n_vox, 1, generator=g))
    W = torch.randn(96, d, generator=g) / 96 ** 0.5
    inds = torch.randint(0, n_vox, (n_pts,), generator=g)
    for n_labels in (20, 160):
        text = torch.nn.functional.normalize(torch.randn(n_labels, d, generator=g), dim=1).half()
        # fused features correlated with the distilled ones so that both sources win for a share of the points
        fd = (f96 @ W)[inds]
        ff = (fd * (0.6 + 0.8 * torch.rand(n_pts, 1, generator=g)) + 0.9 * fd.std() * torch.randn(n_pts, d, generator=g)).half().float()
        ref_scores, ref_labels, _ = oq.query_ensemble(f96 @ W, ff, text, inds)        # gathers the distilled source only ...
        ref_scores = ref_scores.float()
        fdn = fd / (fd.norm(dim=-1, keepdim=True) + 1e-5)
        ffn = ff / (ff.norm(dim=-1, keepdim=True) + 1e-5)
        pd, pf = oq.half_matmul(fdn.half(), text).float().max(1)[0], oq.half_matmul(ffn.half(), text).float().max(1)[0]
        ref_scores2, ref_labels2, _ = oq.query_ensemble(fd, ff, text)                    # ... so evaluate per point directly
        ref_scores2 = ref_scores2.float()
        labels, used, scores = query_ensemble_fused(f96.to(dev()), W.to(dev()), ff.to(dev()), text.to(dev()), inds.to(dev()),
                                                    return_scores=True)
        labels, used, scores = labels.cpu(), used.cpu(), scores.cpu()
        sel_ref = pd < pf
        decided = (pd - pf).abs() > 4e-3
        assert 0.15 < sel_ref.float().mean().item() < 0.85 and decided.float().mean().item() > 0.7
        assert torch.equal(used[decided], sel_ref[decided])
        tol = 1e-3 * max(1.0, ref_scores2.abs().max().item()) + 2e-3
        assert (scores[decided] - ref_scores2[decided]).abs().max().item() <= tol
        top2 = ref_scores2.topk(2, dim=1)[0]
        clear = decided & ((top2[:, 0] - top2[:, 1]) > 2 * tol)
        assert clear.float().mean().item() > 0.4
        assert torch.equal(labels[clear], ref_labels2[clear])
        del ref_scores, ref_labels


def test_query_ensemble():
    from openscene_amd import ops
    xd, t, gather = _query_inputs(3000, 5000, 768, 20, 1)
    xf, _, _ = _query_inputs(3000, 5000, 768, 20, 2)
    ref_scores, ref_arg, ref_sel = cpu_backend.query_ensemble(xd, xf, t, gather, gather)
    scores, arg, sel = ops.query_ensemble(xd.to(dev()), xf.to(dev()), t.to(dev()), gather.to(dev()), gather.to(dev()))
    # selection can only differ where the two best normalised scores are within fp16 rounding of each other
    fd, ff = xd[gather], xf[gather]
    pd = oq.half_matmul((fd / (fd.norm(dim=-1, keepdim=True) + 1e-5)).half(), t).float().max(1)[0]
    pf = oq.half_matmul((ff / (ff.norm(dim=-1, keepdim=True) + 1e-5)).half(), t).float().max(1)[0]
    clear = (pd - pf).abs() > 4e-3
    assert torch.equal(sel.cpu()[clear], ref_sel[clear])
    same = sel.cpu() == ref_sel
    assert (scores.cpu().float()[same] - ref_scores.float()[same]).abs().max().item() <= 2e-3
    assert same.float().mean().item() > 0.98


# ------------------------------------------------------------------- voxelizer
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["voxelize_a.npz", "voxelize_b.npz"])
def test_voxelizer_matches_reference_golden(name):
    """Bit-exact against outputs of the reference's own Voxelizer (tests/golden/make_golden.py)."""
    from openscene_amd.voxelizer import Voxelizer
    g = np.load(os.path.join(GOLDEN, name))
    rot = ((-np.pi / 64, np.pi / 64), (-np.pi / 64, np.pi / 64), (-np.pi, np.pi))
    vox = Voxelizer(voxel_size=float(g["voxel_size"]), clip_bound=None, use_augmentation=True,
                    scale_augmentation_bound=(0.9, 1.1), rotation_augmentation_bound=rot,
                    translation_augmentation_ratio_bound=((-0.2, 0.2), (-0.2, 0.2), (0, 0)))
    rng = np.random.default_rng(int(g["np_seed"]))
    n = g["xyz"].shape[0]
    feats = rng.random((n, 3)) * 255
    labels = np.arange(n) % 20
    np.random.seed(int(g["np_seed"]))
    c, f, l, inv, inds = vox.voxelize(g["xyz"], feats, labels, return_ind=True)
    assert np.array_equal(inds, g["inds"])
    assert np.array_equal(inv, g["inverse"])
    assert np.array_equal(c, g["coords"])
    assert np.array_equal(f, feats[g["inds"]]) and np.array_equal(l, labels[g["inds"]])


@pytest.mark.parametrize("name", ["quantize_small.npz", "quantize_frac.npz"])
def test_quantize_golden(name):
    from openscene_amd import ops
    g = np.load(os.path.join(GOLDEN, name))
    # identity transform with unit voxel: voxelize == sparse_quantize(floor(coords - min))
    coords = g["coords"]
    T = np.eye(4)
    grid, inds, inv = ops.voxelize_fnv(torch.from_numpy(coords).to(dev()), T)
    ref_inds, ref_inv = ov.quantize_first_occurrence(np.floor(np.floor(coords) - np.floor(coords).min(0)))
    assert np.array_equal(inds.cpu().numpy(), ref_inds) and np.array_equal(inv.cpu().numpy(), ref_inv)
    if np.floor(coords).min() == 0:     # then it is literally the golden case
        assert np.array_equal(inds.cpu().numpy(), g["inds"]) and np.array_equal(inv.cpu().numpy(), g["inverse"])


def test_fnv_known_answers():
    from openscene_amd import ops
    g = np.load(os.path.join(GOLDEN, "hash_kat.npz"))
    keys = ops.fnv_hash(torch.from_numpy(g["coords"]).to(dev())).cpu().numpy().view(np.uint64)
    assert [int(v) for v in keys[:3]] == [15658191375538532279, 15657232601398921515, 15489006222804313940]
    assert np.array_equal(keys, g["fnv"])


def test_ravel_known_answers():
    """osn_ravel_hash vs outputs of the reference's own ravel_hash_vec (tests/golden/hash_kat.npz)."""
    from openscene_amd import ops
    g = np.load(os.path.join(GOLDEN, "hash_kat.npz"))
    keys = ops.ravel_hash(torch.from_numpy(g["coords"]).to(dev())).cpu().numpy().view(np.uint64)
    assert np.array_equal(keys, g["ravel"])
    small = np.array([[0, 0, 0], [1, 2, 3], [241, 181, 121]], dtype=np.float64)
    k3 = ops.ravel_hash(torch.from_numpy(small).to(dev())).cpu().numpy().view(np.uint64)
    assert [int(v) for v in k3] == [0, 22451, 5373367]          # minted from the reference function


def test_voxelizer_full_size_properties():
    """ScanNet-size cloud (200 k points): bit-exact vs the numpy oracle + np.unique structure."""
    from openscene_amd import ops
    from openscene_amd import synthetic as syn
    xyz = syn.room_points(9, n_pts=200000)
    np.random.seed(4)
    T = ov.draw_transform(0.02)
    grid, inds, inv = ops.voxelize_fnv(torch.from_numpy(xyz).to(dev()), T)
    rc, ri, rv = ov.voxelize_with_matrix(xyz, T)
    grid, inds, inv = grid.cpu().numpy(), inds.cpu().numpy(), inv.cpu().numpy()
    assert np.array_equal(inds, ri) and np.array_equal(inv, rv) and np.array_equal(grid[inds], rc)
    keys = ov.fnv_keys(grid[inds])
    assert np.all(keys[1:] > keys[:-1])                               # ascending distinct keys
    assert np.array_equal(grid[inds][inv], grid)                      # every point maps to its voxel
    first = np.full(inds.shape[0], xyz.shape[0]); np.minimum.at(first, inv, np.arange(xyz.shape[0]))
    assert np.array_equal(first, inds)                                # first occurrence
