"""The query call site of the reference, UNCHANGED, on the fused HIP kernel.

``run/evaluate.py:289-292`` and ``run/distill.py:421-425`` read

    predictions = model(sinput)
    predictions = predictions[inds_reverse, :]            # voxel -> point gather: [N_pts, 768] float32, 460 MB at 150 k points
    pred = predictions.half() @ text_features.t()         # cast (230 MB), then the matmul
    logits_pred = torch.max(pred, 1)[1]

i.e. three torch launches that write and re-read the gathered matrix (0.365 ms on an MI355X against 0.077 ms for ``osn_cosine_query``,
which gathers, casts, multiplies and rounds in one pass).  An edited call site (``openscene_amd.query.query_distill``) collects that; this
module collects it WITHOUT the edit: the network output that the accelerated forward returns in inference is a ``NetworkOutput`` (an
ordinary tensor, same storage), whose row indexing with an int64 index vector yields a ``GatheredRows`` -- a tensor WITHOUT storage that
remembers (matrix, index).  ``.half()`` keeps it lazy, ``@`` with an fp16 matrix runs the fused kernel and returns the real fp16 score
matrix; EVERY other use (``.norm``, ``/``, ``.clone()``, ``.cpu()``, printing ...: the ensemble branch, run/evaluate.py:307-330) gathers
the rows once with torch, caches them and proceeds on the real tensor -- exactly what the reference computes, only later.  Nothing here
is active under autograd (``output_3d[mask]`` of the training loop, run/distill.py:322, takes torch's path).
"""
import os

import torch

ENABLED = os.environ.get("OSN_LAZY_ROWS", "1") != "0"
_META = frozenset(("shape", "dtype", "device", "size", "dim", "ndim", "ndimension", "numel", "nelement", "is_cuda", "requires_grad",
                   "__len__", "is_floating_point", "is_complex", "element_size", "layout", "grad_fn", "is_leaf", "grad",
                   "is_sparse", "is_quantized", "is_meta", "names", "itemsize", "nbytes", "get_device", "is_contiguous"))
_HALF = frozenset(("half",))
_MATMUL = frozenset(("matmul", "__matmul__", "mm"))


def _name(func):
    n = getattr(func, "__name__", "")
    if n == "__get__":                       # a property of torch.Tensor: func is <getset descriptor>.__get__
        return getattr(getattr(func, "__self__", None), "__name__", "")
    return n


def _plain(x):
    return x.as_subclass(torch.Tensor) if isinstance(x, NetworkOutput) else x


def _row_index(index, base):
    """index of `base[index]` -> the int64 row-index vector when the expression is a plain row gather, else None."""
    if isinstance(index, tuple):
        if len(index) == 2 and isinstance(index[1], slice) and index[1] == slice(None, None, None):
            index = index[0]
        elif len(index) == 1:
            index = index[0]
        else:
            return None
    if isinstance(index, torch.Tensor) and not isinstance(index, GatheredRows) and index.dtype == torch.int64 and index.dim() == 1 \
            and index.device == base.device and base.dim() == 2:
        return index.as_subclass(torch.Tensor) if type(index) is not torch.Tensor else index
    return None


class NetworkOutput(torch.Tensor):
    """The [N_vox, D] output of an accelerated forward pass in inference: a tensor like any other, except that a row gather is lazy."""

    def __reduce_ex__(self, proto):                      # pickles / torch.save as the plain tensor it is
        return self.as_subclass(torch.Tensor).__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.as_subclass(torch.Tensor).__deepcopy__(memo)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if _name(func) == "__getitem__" and len(args) == 2 and isinstance(args[0], NetworkOutput):
            base = args[0]
            if ENABLED and base.dtype == torch.float32 and not (torch.is_grad_enabled() and base.requires_grad):
                idx = _row_index(args[1], base)
                if idx is not None:
                    return GatheredRows(base.as_subclass(torch.Tensor), idx, False)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


class GatheredRows(torch.Tensor):
    """`matrix[index]` not gathered yet (no storage).  half() stays lazy; matmul with an fp16 matrix = the fused query kernel."""

    @staticmethod
    def __new__(cls, matrix, index, half):
        r = torch.Tensor._make_wrapper_subclass(cls, (index.shape[0], matrix.shape[1]), dtype=torch.float16 if half else matrix.dtype,
                                                device=matrix.device, requires_grad=False)
        r._matrix, r._index, r._is_half, r._real = matrix, index, half, None
        return r

    def __init__(self, *a, **k):
        pass

    def __reduce_ex__(self, proto):
        return self.materialize().__reduce_ex__(proto)

    def __deepcopy__(self, memo):
        return self.materialize().clone()

    def materialize(self):
        """The rows, gathered by torch (cached): what the reference's expression holds at this point."""
        if self._real is None:
            t = self._matrix.index_select(0, self._index)
            self._real = t.half() if self._is_half else t
        return self._real

    def _fused(self, other):
        """self @ other on osn_cosine_query, or None when the operands are not the query's (fp16 text matrix [D, C])."""
        if not (self._is_half and self._real is None and self._matrix.is_cuda and type(other) is torch.Tensor and other.dtype == torch.float16 and other.dim() == 2
                and other.device == self._matrix.device and other.shape[0] == self._matrix.shape[1] and not other.requires_grad):
            return None
        from . import ops
        try:
            scores, _labels = ops.cosine_query(self._matrix, other.t(), self._index, want_scores=True)
        except (ValueError, TypeError, RuntimeError):
            return None                      # a shape the kernel does not take: torch's path
        return scores

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = _name(func)
        me = args[0] if args and isinstance(args[0], GatheredRows) else None
        if me is not None:
            if name in _META:
                with torch._C.DisableTorchFunctionSubclass():
                    return func(*args, **kwargs)
            if name in _HALF and len(args) == 1 and not kwargs and me._real is None:
                return me if me._is_half else GatheredRows(me._matrix, me._index, True)
            if name in _MATMUL and len(args) == 2 and not kwargs:
                out = me._fused(args[1])
                if out is not None:
                    return out
        return _on_real(func, args, kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # safety net: anything that reaches the dispatcher with a storage-less tensor runs on the gathered rows
        return _on_real(func, args, kwargs or {})


def _real(x):
    if isinstance(x, GatheredRows):
        return x.materialize()
    if type(x) in (list, tuple):
        return type(x)(_real(v) for v in x)
    if type(x) is dict:
        return {k: _real(v) for k, v in x.items()}
    return _plain(x)


def _on_real(func, args, kwargs):
    with torch._C.DisableTorchFunctionSubclass():
        return func(*_real(tuple(args)), **_real(dict(kwargs)))


def wrap_output(out):
    """Called by the accelerated forward: inference outputs become NetworkOutput (same storage); anything under autograd is left alone."""
    if ENABLED and type(out) is torch.Tensor and out.dim() == 2 and out.dtype == torch.float32 and not out.requires_grad \
            and not torch.is_grad_enabled():
        return out.as_subclass(NetworkOutput)
    return out
