"""The path's one exchange step: the data-parallel gradient all-reduce of
``run/distill.py:149-150`` (``DistributedDataParallel`` around the model), sized for xGMI.

torch's DDP works on these modules unchanged (plain ``nn.Parameter``s, tested) -- but its reducer handles every
parameter separately: per step and parameter one copy into the bucket and one division by the world size, 290 extra
launches and +2.3 ms on a 14.7 ms step for the 145 parameters of MinkUNet18A (measured with one rank on an MI355X,
``tools/ab_ddp.sh``), before a byte moves.  The model's gradients are 62 MB: over xGMI that is ONE collective of a
fraction of a millisecond, so there is nothing to hide behind backward and no reason to bucket.
``FlatGradAllReduce`` therefore keeps one flat fp32 buffer, gathers the gradients into it with one multi-tensor copy
after backward, averages it with one all-reduce (backend ``nccl`` = RCCL on ROCm; ``gloo`` in the CPU tests) and hands
the optimizer views of that buffer.  Same arithmetic as DDP: parameters (and buffers) broadcast from rank 0 at
construction, gradient = mean over ranks.  DDP's default also re-broadcasts rank 0's buffers (BN running statistics)
before EVERY forward; training never reads them (batch statistics are local, no SyncBN in the reference), so here
``sync_buffers()`` is an explicit call for the places that do: before validation (``run/distill.py:207-216``) and
before a checkpoint is written from a rank other than 0 (the reference saves from rank 0 only).  Measured with one
rank: calling it every step costs 1.3 ms (two collectives + their stream hand-overs)."""
import torch
import torch.distributed as dist


class FlatGradAllReduce(object):
    def __init__(self, module, broadcast_buffers=True, process_group=None, single_rank_collectives=False):
        """single_rank_collectives: issue the broadcasts / the all-reduce also in a one-rank group (readiness checks of the
        N > 1 path on one GPU; a real one-rank job skips them)."""
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.collectives = self.world > 1 or single_rank_collectives
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatGradAllReduce needs all parameters on one device with one dtype")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self._pending, self._sliced = [], None
        self.buffers = [b for b in module.buffers() if b.is_floating_point()] if broadcast_buffers else []
        self.int_buffers = [b for b in module.buffers() if not b.is_floating_point()] if broadcast_buffers else []
        # rank 0's parameters (ALL of them, frozen ones included: DDP syncs the whole module state) and buffers
        # everywhere; a real one-rank job has nothing to receive and skips the collectives
        if self.collectives:
            self._broadcast(list(module.parameters()))
            self._broadcast(list(module.buffers()))

    def _broadcast(self, tensors):
        """rank 0's values into `tensors` on every rank: one flat buffer, one collective and one multi-tensor copy per dtype."""
        by_type = {}
        for t in tensors:
            by_type.setdefault(t.dtype, []).append(t)
        for ts in by_type.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, 0, group=self.group)
            if self.rank == 0:
                continue                            # the source already holds these values
            views, off = [], 0
            for t in ts:
                views.append(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            # in place on the parameters / buffers themselves (not on `.data`): the copy bumps their version counters,
            # which is what keys the convolution weight-image cache of openscene_amd.ops (a forward that ran before this
            # exchange was built must not leave images of the pre-broadcast weights behind)
            with torch.no_grad():
                torch._foreach_copy_(ts, views)
        from . import ops
        ops.clear_weight_cache()

    def sync_buffers(self):
        """Before a forward: rank 0's buffers (BN running statistics, counters) on every rank, as DDP's
        broadcast_buffers=True does.  One collective per dtype."""
        if self.collectives and (self.buffers or self.int_buffers):
            self._broadcast(self.buffers)
            self._broadcast(self.int_buffers)

    def _common_base(self):
        """The one contiguous fp32 buffer all gradients live in (same on every rank: the executor's layout is a function of
        the module tree) as a 1-D tensor over their common storage, or None.  See optim.shared_flat for why this is decided
        by storage and not by `_base`."""
        from .optim import shared_flat
        base = shared_flat([p.grad for p in self.params])
        if base is None or base.dtype != self.flat.dtype or base.device != self.flat.device:
            return None
        return base

    # ---- overlapped exchange of the executor's flat gradient buffer --------------------------------------------
    def attach(self, executor, segments=4):
        """Exchange the executor's convolution weight gradients in `segments` pieces WHILE the backward pass runs: the pass
        is played in that many segments (highest ops first) and each finished slice is all-reduced asynchronously; what is
        left for reduce_gradients() is the batch-norm region and the wait.  Same values as the one-collective exchange
        (mean over ranks; a sum's chunking does not change its elements)."""
        executor.grad_segments = max(1, int(segments))
        executor._cuts = None
        executor.grad_ready_hook = self._on_slice
        self._pending = []
        self._sliced = None

    def _avg(self, t, async_op):
        """t <- mean over ranks.  RCCL averages inside the collective (no separate division launch); gloo gets div + sum."""
        if not self.collectives:
            return None
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        if self.world > 1:
            t.div_(self.world)
        return dist.all_reduce(t, group=self.group, async_op=async_op)

    def _on_slice(self, flat, lo, hi, last):
        if self._sliced is None or self._sliced[0] is not flat:
            self._sliced = [flat, hi, hi]              # [buffer, lowest float exchanged so far, end of the kernels' region]
        if hi > lo:
            w = self._avg(flat[lo:hi], async_op=True)
            if w is not None:
                self._pending.append(w)
        self._sliced[1] = min(self._sliced[1], lo)
        if last:
            # before autograd sees the gradients: it may keep the slices' memory (the normal case) or CLONE them (hooks on a
            # parameter, a second use) -- a clone taken with a collective in flight would hold half-exchanged values.  With
            # RCCL this is a stream-side wait (the host does not block), and it is the wait reduce_gradients() would do anyway.
            for w in self._pending:
                w.wait()
            self._pending = []

    def reduce_gradients(self):
        """After backward: p.grad <- mean over ranks, as views of the flat buffer (no copy back).
        A parameter without a gradient on this rank contributes zeros (DDP with find_unused_parameters) and, like
        there, ends up with the (possibly all-zero) mean as its gradient: an optimizer with momentum / weight decay
        then still advances its state for a parameter NO rank used -- a documented deviation from a single-process
        run, where such a parameter keeps grad None (no parameter of the MinkUNet family is unused)."""
        # the network executor (openscene_amd/executor.py) hands every gradient out as a view of ONE flat buffer it wrote
        # in place: exchange that buffer itself -- no gather copy, the optimizer keeps reading the same views
        for w in self._pending:                     # (a backward pass that did not reach its last segment)
            w.wait()
        self._pending = []
        base = self._common_base()
        if base is not None:
            sl = getattr(self, "_sliced", None)
            if sl is not None and sl[0].untyped_storage().data_ptr() == base.untyped_storage().data_ptr():
                # slices [sl[1], sl[2]) went out during the backward pass: what is left is the head of the kernels' region
                # (nothing, normally) and the batch-norm region behind it
                if sl[1] > 0:
                    self._avg(base[:sl[1]], async_op=False)
                if sl[2] < base.numel():
                    self._avg(base[sl[2]:], async_op=False)
                self._sliced = None
                return
            self._sliced = None
            self._avg(base, async_op=False)
            return
        # (gradients that do not share one buffer -- e.g. autograd cloned them: slices exchanged during the pass are final in
        # the clones, a second average of equal values changes nothing but round-off)
        self._sliced = None
        grads = []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                grads.append(None)
            else:
                grads.append(p.grad)
        src = [g for g in grads if g is not None]
        dst = [v for g, v in zip(grads, self.views) if g is not None]
        same = all(g.data_ptr() == v.data_ptr() for g, v in zip(src, dst))
        if not same:
            torch._foreach_copy_(dst, src)          # one multi-tensor copy into the flat buffer
        self._avg(self.flat, async_op=False)
        for p, v in zip(self.params, self.views):
            p.grad = v
