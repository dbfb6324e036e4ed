"""Adam over one flat parameter buffer -- the mirror of ``torch.optim.Adam(model.parameters(), lr=...)``
(run/distill.py:170-178, stepped at :333) for the case the network executor creates: every gradient is a slice of ONE flat
fp32 buffer.  ``FlatAdam`` lays the parameters and both moment estimates out the same way (each parameter's slice starts on
a 16-byte boundary) and an optimizer step is one launch (csrc/optim.hip: ``osn_adam_step``) instead of torch's fused
multi-tensor Adam (5 launches, 236 us per step for MinkUNet18A; this: ~60 us).

    optim = FlatAdam(model, lr=1e-4)        # re-points every parameter's storage into the flat buffer (values kept)
    loss.backward(); optim.step()

Update rule, hyper-parameters and state semantics are torch.optim.Adam's (amsgrad=False, maximize=False, L2 weight decay);
`state_dict()` / `load_state_dict()` use torch's layout (`exp_avg`, `exp_avg_sq`, `step` per parameter, keyed by the
parameter's position in `model.parameters()` order whatever the flat layout is), so checkpoints
written by the reference's save_checkpoint (run/distill.py:232-239) load here and vice versa."""
import torch

from . import ops
from ._lib import check


def shared_flat(grads):
    """-> the ONE contiguous 1-D tensor all `grads` live in (a view over their common storage), or None.

    Decided by STORAGE, not by `Tensor._base`: a gradient that came through autograd is `new_grad.detach()` of the slice the
    backward function returned (AccumulateGrad keeps the memory, torch/csrc/autograd/functions/accumulate_grad.h) -- it
    still aliases the executor's flat buffer but is no longer a view in the autograd sense, `_base` is None.  (Round 4 found
    the `_base` test never true on a real backward pass: FlatAdam gathered a copy every step and, worse, an attached sliced
    exchange fell through to a second full-buffer average on top of its in-flight slices.)"""
    st = None
    first = None
    for g in grads:
        if g is None or not g.is_contiguous() or g.is_sparse:
            return None
        s = g.untyped_storage()
        if first is None:
            first, st = g, s.data_ptr()
        elif s.data_ptr() != st or g.dtype != first.dtype or g.device != first.device:
            return None
    if first is None or st == 0:
        return None
    n = first.untyped_storage().nbytes() // first.element_size()
    return torch.empty(0, dtype=first.dtype, device=first.device).set_(first.untyped_storage(), 0, (n,))


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, model_or_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not isinstance(model_or_params, torch.nn.Module):
            model_or_params = list(model_or_params)          # (a generator such as model.parameters() is consumed once)
        params = self._ordered(model_or_params)
        # checkpoint index of every parameter = its position in `model.parameters()` order (what torch.optim.Adam, built the
        # way run/distill.py:141 builds it, keys its state by) -- NOT its position in the flat layout, which follows the
        # executor's gradient buffer (all conv kernels first, then the BN weight / bias pairs)
        given = (list(model_or_params.parameters()) if isinstance(model_or_params, torch.nn.Module) else list(model_or_params))
        # torch keys by position over ALL the parameters it was given, frozen ones included (they simply never get a state entry)
        torch_pos = {}
        for q in given:
            torch_pos.setdefault(id(q), len(torch_pos))
        self._n_given = len(torch_pos)
        self._ckpt_index = [torch_pos[id(q)] for q in params]
        if not params:
            raise ValueError("FlatAdam got no parameters")
        if not all(p.dtype == torch.float32 and p.device == params[0].device for p in params):
            raise ValueError("FlatAdam needs float32 parameters on one device")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        dev = params[0].device
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.steps = 0
        self._grad_scratch = None
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = self.flat[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view                      # same values, new storage: cached weight images follow the data pointer
        ops.clear_weight_cache()
        self._params = params

    @staticmethod
    def _ordered(model_or_params):
        """The parameters in the order of the executor's gradient buffer when the model has one (so that the gradients arrive
        as one flat array with the optimizer's own layout), else in the order given."""
        if isinstance(model_or_params, torch.nn.Module):
            from . import executor as E
            from .drop_in import is_unet
            seen, out = set(), []
            for m in model_or_params.modules():
                if is_unet(m):
                    ex = E.for_model(m) if E.ENABLED else None
                    if ex is not None:
                        for p in ex.program.params:
                            if id(p) not in seen:
                                seen.add(id(p)); out.append(p)
            for p in model_or_params.parameters():
                if id(p) not in seen:
                    seen.add(id(p)); out.append(p)
            return [p for p in out if p.requires_grad]
        return [p for p in model_or_params if p.requires_grad]

    def _flat_grads(self):
        """The gradients as one flat array in this optimizer's layout: the executor's buffer itself when every p.grad is a
        slice of it at the right offset, else a gathered copy (parameters without a gradient contribute zeros: like torch's
        Adam they would be skipped -- no MinkUNet parameter is unused)."""
        ps = self._params
        root = shared_flat([p.grad for p in ps])
        if (root is not None and root.dtype == torch.float32 and root.numel() >= self.total and
                all(p.grad.storage_offset() == o for p, o in zip(ps, self.offsets))):
            return root[:self.total], True
        if self._grad_scratch is None:
            self._grad_scratch = torch.zeros_like(self.flat)
        views = [self._grad_scratch[o:o + p.numel()].view_as(p) for p, o in zip(ps, self.offsets)]
        have = [(v, p.grad) for v, p in zip(views, ps) if p.grad is not None]
        for v, p in zip(views, ps):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        return self._grad_scratch, False

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        grads, _ = self._flat_grads()
        dev = self.flat.device
        lib = ops._prep(dev)
        self.steps += 1
        b1, b2 = group["betas"]
        with ops._Dev(dev):
            check(lib.osn_adam_step(ops._p(self.flat), ops._p(grads), ops._p(self.exp_avg), ops._p(self.exp_avg_sq), self.total,
                                    self.steps, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                    float(group["weight_decay"]), ops._stream(dev)), "osn_adam_step")
        # the kernel wrote the parameters behind autograd's back: move their version counters (weight-image cache, saved-tensor checks)
        torch.autograd.graph.increment_version(self._params)
        return loss

    # ---- torch.optim.Adam's checkpoint layout
    def state_dict(self):
        state = {}
        for i, p, o in sorted(zip(self._ckpt_index, self._params, self.offsets), key=lambda t: t[0]):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        groups = [{**{k: v for k, v in g.items() if k != "params"}, "params": list(range(self._n_given))} for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        steps = 0
        if len(sd["param_groups"][0]["params"]) != self._n_given:
            raise ValueError("loaded state dict holds %d parameters, this optimizer was built over %d"
                             % (len(sd["param_groups"][0]["params"]), self._n_given))
        for i, p, o in zip(self._ckpt_index, self._params, self.offsets):
            st = sd["state"].get(i)
            if st is None:
                continue
            n = p.numel()
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError("state of parameter %d has shape %s, the parameter %s" % (i, tuple(st["exp_avg"].shape), tuple(p.shape)))
            self.exp_avg[o:o + n].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
            steps = max(steps, int(float(st["step"])))
        self.steps = steps
