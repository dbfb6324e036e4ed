"""Distillation loss on the supervised rows -- the mirror of run/distill.py:322-328:

    output_3d = model(sinput)[mask]
    loss = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()      # loss_type: cosine
    loss = torch.nn.L1Loss()(output_3d, feat_3d)                             # loss_type: l1

as ONE autograd node over the full network output: the forward pass reads the selected rows once, the backward pass
writes the [N, D] output gradient once (zeros on the rows the loss does not see), csrc/loss.hip."""
import torch
from torch.autograd import Function

from . import ops
from ._lib import check

KINDS = {"cosine": 0, "l1": 1}
ROWS_HINT = True          # attach the compacted non-zero rows to the output gradient (see _DistillLoss.backward)


class _DistillLoss(Function):
    @staticmethod
    def forward(ctx, out, sel, target, kind, validate):
        dev = out.device
        lib = ops._prep(dev)
        out = ops._f32c(out, "output")
        target = ops._f32c(target, "target")
        if sel.dtype != torch.int64 or sel.dim() != 1 or sel.device != dev:
            raise ValueError("sel must be an int64 vector of row indices on the output's device")
        sel = sel.contiguous()
        n, d = out.shape
        n_sel = sel.shape[0]
        if tuple(target.shape) != (n_sel, d):
            raise ValueError("target is %s, the selected output rows are (%d, %d)" % (tuple(target.shape), n_sel, d))
        if n_sel == 0:
            raise ValueError("the loss needs at least one supervised row")
        state = torch.empty(int(ops._cached("osn_distill_loss_state_bytes", n, n_sel)), dtype=torch.uint8, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        with ops._Dev(dev):
            check(lib.osn_distill_loss_fwd(ops._p(out), ops._p(sel), ops._p(target), n, n_sel, d, kind, ops._p(loss), ops._p(state),
                                           state.numel(), ops._stream(dev)), "osn_distill_loss_fwd")
            if validate:
                check(lib.osn_distill_loss_check(ops._p(state), n, n_sel, ops._stream(dev)), "osn_distill_loss_check")
        ctx.save_for_backward(out, target, state)
        ctx.sel = sel
        ctx.cfg = (n, n_sel, d, kind)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        out, target, state = ctx.saved_tensors
        n, n_sel, d, kind = ctx.cfg
        dev = out.device
        lib = ops._prep(dev)
        gloss = gloss.to(torch.float32).contiguous()
        gout = torch.empty_like(out)
        # the gradient of the full output is zero outside the n_sel supervised rows: hand those rows out once more, compacted,
        # with the row <-> position tables of the forward pass.  A consumer that understands the hint (the network executor's
        # head) works on n_sel rows instead of n; everyone else sees the ordinary dense gradient.
        # (zeros, not empty: the kernel writes row j only where pos[sel[j]] == j -- with a duplicate in `sel`, which validate=False
        # does not catch, the losing copy's row would be uninitialised memory multiplied into the head's gradients)
        grows = torch.zeros((n_sel, d), dtype=torch.float32, device=dev) if ROWS_HINT and n_sel < n else None
        with ops._Dev(dev):
            check(lib.osn_distill_loss_bwd_rows(ops._p(out), ops._p(target), ops._p(gloss), n, n_sel, d, kind, ops._p(gout),
                                                ops._p(grows), ops._p(state), state.numel(), ops._stream(dev)),
                  "osn_distill_loss_bwd_rows")
        if grows is not None:
            gout._osn_rows = {"ptr": gout.data_ptr(), "shape": tuple(gout.shape), "idx": ctx.sel, "rows": grows,
                              "pos_ptr": state.data_ptr(), "state": state, "version": gout._version}
        return gout, None, None, None, None


_ident = {}


def _identity(n, dev):
    key = (n, str(dev))
    t = _ident.get(key)
    if t is None:
        if len(_ident) > 8:
            _ident.clear()
        t = _ident[key] = torch.arange(n, dtype=torch.int64, device=dev)
    return t


def distill_loss(output, sel, target, loss_type="cosine", validate=False):
    """Scalar loss of run/distill.py:322-328 over `output[sel]` against `target` (feat_3d).
    output float32 [N, D] (the network output, input row order); sel: the supervised rows -- int64 indices (CONTRACT: distinct and
    in [0, N): an index out of range reads out of bounds, a duplicate is counted once in the backward pass -- validate=True checks both; e.g.
    mask.nonzero().squeeze(1), which the loader hands out next to the mask) or the bool mask itself (resolved here: a host
    synchronisation in the middle of the step); target float [len(sel), D].
    validate: check the indices on the device and raise on an index out of range or a duplicate (synchronises)."""
    if loss_type not in KINDS:
        raise ValueError("loss_type %r (the reference has 'cosine' and 'l1')" % (loss_type,))
    if sel is None:
        # `output` already IS output[sel] (model(sinput, rows=sel)): every row is supervised
        sel = _identity(output.shape[0], output.device)
    if sel.dtype == torch.bool:
        sel = sel.nonzero(as_tuple=False).squeeze(1)
    return _DistillLoss.apply(output, sel, target, KINDS[loss_type], bool(validate))
