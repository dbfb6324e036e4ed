"""Host logic of openscene_amd.optim.FlatAdam on the CPU: the kernel behind osn_adam_step is replaced by a numpy restatement
of its update rule (float32, operating on the pointers the wrapper passes), everything else is the product code -- parameter
flattening, the executor-ordered layout, in-place use of a flat gradient buffer against the gathered fallback, version
counters, torch.optim.Adam's checkpoint layout."""
import ctypes

import numpy as np
import pytest
import torch


class _FakeLib:
    """osn_adam_step in numpy, same argument list as include/openscene_amd.h."""

    def __init__(self):
        self.calls = 0

    @staticmethod
    def _arr(ptr, n):
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))

    def osn_adam_step(self, p, g, m, v, n, step, lr, b1, b2, eps, wd, stream):
        self.calls += 1
        p, g, m, v = (self._arr(x, n) for x in (p, g, m, v))
        f = np.float32
        grad = g + f(wd) * p if wd else g.copy()
        m += (f(1) - f(b1)) * (grad - m)
        v[:] = f(b2) * v + (f(1) - f(b2)) * grad * grad
        bc1 = 1.0 - float(b1) ** step
        bc2 = 1.0 - float(b2) ** step
        denom = np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)
        p -= f(lr / bc1) * m / denom
        return 0


@pytest.fixture()
def fake_adam(monkeypatch):
    from openscene_amd import ops
    lib = _FakeLib()
    monkeypatch.setattr(ops, "_prep", lambda dev: lib)
    monkeypatch.setattr(ops, "_stream", lambda dev: None)

    class NoDev:
        def __init__(self, dev):
            pass

        def __enter__(self):
            pass

        def __exit__(self, *a):
            pass
    monkeypatch.setattr(ops, "_Dev", NoDev)
    return lib


@pytest.mark.parametrize("flat_grads", [False, True])
def test_flat_adam_layout_versions_and_checkpoints(fake_adam, flat_grads):
    from openscene_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(8, 4, 4), (6,), (3, 5), (1,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    ref_p = [torch.nn.Parameter(t.clone()) for t in init]
    our_p = [torch.nn.Parameter(t.clone()) for t in init]
    ref = torch.optim.Adam(ref_p, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
    ours = FlatAdam(our_p, lr=2e-3, betas=(0.9, 0.98), weight_decay=0.01)
    assert ours.offsets == [0, 128, 136, 152] and ours.total == 156          # every slice starts on a 16-byte boundary
    assert all(p.data_ptr() == ours.flat.data_ptr() + 4 * o for p, o in zip(our_p, ours.offsets))
    assert all(torch.equal(a.detach(), b.detach()) for a, b in zip(ref_p, our_p))
    for step in range(4):
        grads = [torch.randn(s, generator=g) for s in shapes]
        flat = torch.zeros(ours.total)
        for p, q, gr, o in zip(ref_p, our_p, grads, ours.offsets):
            p.grad = gr.clone()
            q.grad = flat[o:o + gr.numel()].view(gr.shape).copy_(gr) if flat_grads else gr.clone()
        got, in_place = ours._flat_grads()
        assert in_place == flat_grads and (got.data_ptr() == flat.data_ptr()) == flat_grads
        before = [q._version for q in our_p]
        ref.step()
        ours.step()
        assert all(q._version > b for q, b in zip(our_p, before))
        for p, q in zip(ref_p, our_p):
            assert torch.allclose(q.detach(), p.detach(), rtol=1e-5, atol=1e-6)
    assert fake_adam.calls == 4
    # a parameter without a gradient contributes zeros through the gathered path
    our_p[1].grad = None
    assert ours._flat_grads()[1] is False
    sd = ours.state_dict()
    ref_sd = ref.state_dict()
    assert set(sd["state"]) == set(ref_sd["state"]) and sd["param_groups"][0]["lr"] == 2e-3
    for i in range(len(shapes)):
        assert torch.allclose(sd["state"][i]["exp_avg"], ref_sd["state"][i]["exp_avg"], rtol=1e-5, atol=1e-6)
        assert sd["state"][i]["exp_avg"].shape == torch.Size(shapes[i])
    again = FlatAdam([torch.nn.Parameter(t.clone()) for t in init], lr=1.0)
    again.load_state_dict(ref_sd)
    assert again.steps == 4 and again.param_groups[0]["betas"] == (0.9, 0.98)


def test_flat_adam_adopts_the_executor_order_for_a_model(fake_adam):
    """Given a module, the optimizer lays its buffer out in the order of the network executor's gradient buffer (convolution
    kernels first, then the batch-norm pairs), so that the executor's gradients can be read in place."""
    from openscene_amd import executor as E
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.optim import FlatAdam
    model = mink_unet(3, 20, 3, "MinkUNet14A")
    ex = E.for_model(model)
    opt = FlatAdam(model, lr=1e-3)
    assert [id(p) for p in opt._params] == [id(p) for p in ex.program.params]
    assert opt.offsets == ex.grad_off and opt.total == ex.grad_total
    assert len(opt._params) == len(list(model.parameters()))


def test_flat_adam_checkpoints_round_trip_with_torch_adam_on_a_model(fake_adam):
    """the flat layout follows the executor (kernels first, BN pairs after), the CHECKPOINT follows
    model.parameters() order -- a torch.optim.Adam state dict loads into FlatAdam and back with every moment on its own
    parameter (same-shaped parameters would otherwise be silently swapped)."""
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.optim import FlatAdam
    torch.manual_seed(3)
    model_a = mink_unet(3, 20, 3, "MinkUNet14A")
    model_b = mink_unet(3, 20, 3, "MinkUNet14A")
    model_b.load_state_dict(model_a.state_dict())
    ref = torch.optim.Adam(model_a.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(9)
    for _ in range(2):
        for p in model_a.parameters():
            p.grad = torch.randn(p.shape, generator=g)
        ref.step()
    ref_sd = ref.state_dict()
    ours = FlatAdam(model_b, lr=1.0)
    assert [id(p) for p in ours._params] != [id(p) for p in model_b.parameters()]       # the layouts really differ
    ours.load_state_dict(ref_sd)
    assert ours.steps == 2 and ours.param_groups[0]["lr"] == 1e-3
    names = [n for n, _ in model_b.named_parameters()]
    by_id = {id(p): (o, p) for p, o in zip(ours._params, ours.offsets)}
    for i, p in enumerate(model_b.parameters()):
        o, _ = by_id[id(p)]
        assert torch.equal(ours.exp_avg[o:o + p.numel()].view_as(p), ref_sd["state"][i]["exp_avg"]), names[i]
        assert torch.equal(ours.exp_avg_sq[o:o + p.numel()].view_as(p), ref_sd["state"][i]["exp_avg_sq"]), names[i]
    back = ours.state_dict()
    assert list(back["state"]) == list(ref_sd["state"]) and back["param_groups"][0]["params"] == ref_sd["param_groups"][0]["params"]
    fresh = torch.optim.Adam(model_b.parameters(), lr=1.0)
    fresh.load_state_dict(back)                                     # torch does not check shapes: compare the values
    for i, p in enumerate(model_b.parameters()):
        assert torch.equal(fresh.state[p]["exp_avg"], ref_sd["state"][i]["exp_avg"]), names[i]
        assert float(fresh.state[p]["step"]) == 2.0
    # a state dict of a different model is refused instead of mis-assigned
    other = torch.optim.Adam(mink_unet(3, 20, 3, "MinkUNet18A").parameters(), lr=1e-3).state_dict()
    with pytest.raises(ValueError):
        ours.load_state_dict(other)


def test_flat_adam_accepts_a_parameter_generator(fake_adam):
    """torch.optim.Adam(model.parameters()) is the reference's call (run/distill.py:141): a generator must work here too."""
    from openscene_amd.optim import FlatAdam
    lin = torch.nn.Linear(4, 3)
    opt = FlatAdam(lin.parameters(), lr=1e-2)
    assert len(opt._params) == 2 and opt._ckpt_index == [0, 1]
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert fake_adam.calls == 1


class _FlatNode(torch.autograd.Function):
    """What the network executor's autograd node does: every parameter's gradient is a slice of ONE buffer made in backward()."""
    made = []

    @staticmethod
    def forward(ctx, x, *ps):
        ctx.shapes = [p.shape for p in ps]
        return x.sum() + sum((p * p).sum() for p in ps)

    @staticmethod
    def backward(ctx, g):
        offs, off = [], 0
        for s in ctx.shapes:
            offs.append(off)
            off += (s.numel() + 3) // 4 * 4
        flat = torch.arange(off, dtype=torch.float32)
        _FlatNode.made.append(flat.data_ptr())
        return (None,) + tuple(flat[o:o + s.numel()].view(s) for o, s in zip(offs, ctx.shapes))


def test_gradients_that_came_through_autograd_are_still_found_in_their_flat_buffer(fake_adam):
    """AccumulateGrad keeps the memory of the slices the node returned but detaches them: `p.grad._base` is None although
    every gradient still lives in the node's flat buffer.  Both consumers (FlatAdam, FlatGradAllReduce) must find the buffer
    by storage -- round 4 found the `_base` test never true after a real backward pass."""
    from openscene_amd.optim import FlatAdam, shared_flat
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 2))]
    opt = FlatAdam(ps, lr=1e-3)
    _FlatNode.apply(torch.zeros(2), *ps).backward()
    assert all(p.grad._base is None for p in ps)                         # the premise
    flat = shared_flat([p.grad for p in ps])
    assert flat is not None and flat.data_ptr() == _FlatNode.made[-1] and flat.numel() == 24
    assert torch.equal(flat, torch.arange(24, dtype=torch.float32))
    got, in_place = opt._flat_grads()
    assert in_place and got.data_ptr() == _FlatNode.made[-1] and got.numel() == opt.total
    # separately allocated gradients, a missing one, or a different layout: no shared buffer / the gathered copy
    ps[1].grad = ps[1].grad.clone()
    assert shared_flat([p.grad for p in ps]) is None and opt._flat_grads()[1] is False
    ps[1].grad = None
    assert shared_flat([p.grad for p in ps]) is None
    assert shared_flat([]) is None


def test_flat_adam_checkpoint_indices_count_frozen_parameters_like_torch(fake_adam):
    """torch.optim.Adam(model.parameters()) (run/distill.py:141) keys its state by position over ALL the parameters it
    was given, frozen ones included; FlatAdam numbers its checkpoint the same way, so the two interchange for a model with a
    frozen parameter (before: shifted indices and a length error)."""
    from openscene_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(4, 4), (6,), (4, 4), (2,)]
    init = [torch.randn(s, generator=g) for s in shapes]

    def params():
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        ps[1].requires_grad_(False)                   # frozen, in the middle
        return ps
    ref_p, our_p = params(), params()
    ref = torch.optim.Adam(ref_p, lr=1e-3)
    ours = FlatAdam(our_p, lr=1e-3)
    for _ in range(2):
        for p, q in zip(ref_p, our_p):
            if p.requires_grad:
                p.grad = torch.randn(p.shape, generator=g)
                q.grad = p.grad.clone()
        ref.step()
        ours.step()
    sd, ref_sd = ours.state_dict(), ref.state_dict()
    assert sorted(sd["state"]) == sorted(ref_sd["state"]) == [0, 2, 3]
    assert sd["param_groups"][0]["params"] == ref_sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    for i in (0, 2, 3):
        assert torch.allclose(sd["state"][i]["exp_avg_sq"], ref_sd["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    again = FlatAdam(params(), lr=1.0)
    again.load_state_dict(ref_sd)                     # a torch checkpoint with a stateless frozen entry
    assert again.steps == 2
    ref2 = torch.optim.Adam(params(), lr=1e-3)
    ref2.load_state_dict(sd)                          # and back
    assert sorted(ref2.state_dict()["state"]) == [0, 2, 3]
