"""Host side of the network executor (csrc/net.hip, include/openscene_amd.h "network executor").

``MinkUNetBase.forward`` (the mirror of models/mink_unet.py:116-174) is compiled ONCE into a linear program of stages
-- conv [-> BN (+ residual) (+ ReLU)] [-> second store into a cat buffer] -- and every forward / backward pass is then
ONE C call that issues all its launches, instead of ~250 trips through Python, autograd and ctypes per training step.
Same kernels, same results as the module-by-module path (``minkowski.py`` / ``functional.py``), which stays the drop-in
surface for the reference's own ``models/mink_unet.py`` and the path of every configuration the executor does not take
(``OSN_EXECUTOR=0``, an arithmetic mode other than "tl", BN variants the reference never builds).

What Python still does per pass: build the coordinate manager's maps (``CoordinateManager.prebuild``), hand their
device pointers over, allocate the activation arena and the flat gradient buffer, refresh the weight images.
"""
import ctypes
import os
import weakref

import numpy as np
import torch
from torch.autograd import Function

from . import _lib, ops
from ._lib import check

ENABLED = os.environ.get("OSN_EXECUTOR", "1") != "0"
# a second stream for the work that is off the main dependency chain: every weight gradient of the backward pass (needed
# only at its end) and the BasicBlock shortcut stages of both passes.  Bitwise the same results; the GPU is the bottleneck
# since the executor took the host out of the way, and most launches leave compute units idle (measured: -0.86 ms / step)
SIDE_STREAM = os.environ.get("OSN_SIDE_STREAM", "1") != "0"
# the head's input / weight gradients on the supervised rows only when the loss says which rows of the output gradient are
# non-zero (losses.distill_loss does): 20 k of 101 k rows in the reference's configuration.  Exactly the same values.
ROW_SPARSE_HEAD = os.environ.get("OSN_ROW_SPARSE_HEAD", "1") != "0"
# A training step's first launch is the batched refresh of the weight images (~70 us, needs only the optimizer's output); the stem
# convolution that follows reads the fp32 kernel itself, no image.  The refresh is queued on the side stream and the forward pass is
# played as ops [0, 1) -> wait for the refresh -> ops [1, n): refresh and stem run side by side (round 5's A/B: -0.03 ms alone,
# -0.08 ms with the buffer-resource gathers, same loss bits; profiles/r05_s1_knobs_ab.txt).
_DRY_RUN = False        # tools/dryrun only: accept host tensors (a null HIP runtime logs the launches instead of running them)

_OP = np.dtype([(n, "<i4") for n in ("K", "cin", "cout", "lvl_in", "lvl_out", "map", "transposed", "src", "dst", "bn", "relu",
                                     "res", "copy_buf", "copy_col", "weight", "need_dgrad", "fine_unique", "reserved")])
_BUF = np.dtype([("level", "<i4"), ("channels", "<i4")])
_MAP = np.dtype([(n, "<u8") for n in ("nbr_fwd", "nbr_bwd", "tiles_fwd_rows", "tiles_fwd_tbl", "tiles_fwd_gmask",
                                      "tiles_bwd_rows", "tiles_bwd_tbl", "tiles_bwd_gmask", "counts", "tl_fwd", "tl_fwd_rows",
                                      "tl_bwd", "tl_bwd_rows", "pl_fwd")] +
                [(n, "<i4") for n in ("K", "flip", "tl_fwd_bm", "tl_bwd_bm")])
_WEIGHT = np.dtype([(n, "<u8") for n in ("W", "x6_fwd", "x6_dgrad", "tl_fwd", "tl_dgrad", "gW")])
_BN = np.dtype([(n, "<u8") for n in ("gamma", "beta", "running_mean", "running_var", "ggamma", "gbeta")] +
               [("eps", "<f4"), ("momentum", "<f4")])
assert _OP.itemsize == 72 and _MAP.itemsize == 128 and _WEIGHT.itemsize == 48 and _BN.itemsize == 56

IMG_X6_FWD, IMG_X6_DGRAD, IMG_TL_FWD, IMG_TL_DGRAD = 1, 2, 4, 8
K_NAMES = {0: "none", 1: "stem", 2: "tl", 3: "x6", 4: "wgrad_tl", 5: "wgrad", 6: "ws", 7: "ws_direct", 8: "wgrad_stem", 9: "dense", 10: "rg"}
K_WS = (6, 7)


class _Desc(ctypes.Structure):
    _fields_ = [("n_ops", ctypes.c_int32), ("n_bufs", ctypes.c_int32), ("n_bns", ctypes.c_int32), ("n_weights", ctypes.c_int32),
                ("n_maps", ctypes.c_int32), ("n_levels", ctypes.c_int32), ("tl_min_rows", ctypes.c_int32),
                ("tl_mid_rows", ctypes.c_int32), ("ws_max_rows", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("ops", ctypes.c_void_p), ("bufs", ctypes.c_void_p)]


class _Plan(ctypes.Structure):
    _fields_ = [("fwd_arena_bytes", ctypes.c_uint64), ("bwd_arena_bytes", ctypes.c_uint64), ("ws_bytes", ctypes.c_uint64),
                ("x_off", ctypes.c_void_p), ("stat_off", ctypes.c_void_p), ("y_off", ctypes.c_void_p),
                ("fwd_kernel", ctypes.c_void_p), ("dgrad_kernel", ctypes.c_void_p), ("wgrad_kernel", ctypes.c_void_p),
                ("images", ctypes.c_void_p)]


class _Run(ctypes.Structure):
    _fields_ = [("level_rows", ctypes.c_void_p), ("maps", ctypes.c_void_p), ("weights", ctypes.c_void_p), ("bns", ctypes.c_void_p),
                ("input", ctypes.c_void_p), ("output", ctypes.c_void_p), ("goutput", ctypes.c_void_p),
                ("fwd_arena", ctypes.c_void_p), ("fwd_arena_bytes", ctypes.c_uint64),
                ("bwd_arena", ctypes.c_void_p), ("bwd_arena_bytes", ctypes.c_uint64),
                ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_uint64),
                ("tl_counters", ctypes.c_void_p), ("training", ctypes.c_int32), ("first_op", ctypes.c_int32),
                ("end_op", ctypes.c_int32), ("flags", ctypes.c_int32), ("prof", ctypes.c_void_p),
                ("side_stream", ctypes.c_void_p), ("ws_side", ctypes.c_void_p), ("ws_side_bytes", ctypes.c_uint64),
                ("events", ctypes.c_void_p),
                ("grows_pos", ctypes.c_void_p), ("grows_idx", ctypes.c_void_p), ("goutput_rows", ctypes.c_void_p),
                ("n_grows", ctypes.c_int64)]


RUN_NO_JOIN = 1                                 # include/openscene_amd.h: OSN_NET_RUN_NO_JOIN
RUN_NO_BN_EPILOGUE = 2                          # include/openscene_amd.h: OSN_NET_RUN_NO_BN_EPILOGUE
# Inference: every stage's batch norm (+ residual) (+ ReLU) (+ cat store) runs in the epilogue of the kernel that finishes the stage's
# convolution (csrc/epilogue.h; bitwise the separate launch, 48 launches fewer per MinkUNet18A pass).  0: the separate launches (A/B, tests)
BN_EPILOGUE = os.environ.get("OSN_BN_EPILOGUE", "1") != "0"
# Streams the map chains of an INFERENCE pass are dealt to (ops.maps_build; training keeps ops.MAPS_STREAMS = 1: its maps are built
# beside the previous step on the prefetcher's stream, and more streams than hardware queues hurt there).  With the host-side
# marshalling cached (round 6) the pass is no longer waiting on the host between the map build and the forward pass, so the
# shorter map phase shows: maps_only 1.04 -> 0.79 ms (tools/maps_host_time.py)
INFER_MAPS_STREAMS = int(os.environ.get("OSN_INFER_MAPS_STREAMS", "3"))
TRAIN_MAPS_STREAMS = os.environ.get("OSN_TRAIN_MAPS_STREAMS", "1") != "0"     # the same for the maps a training step builds inside itself


def _ptr(a):
    return a.ctypes.data


class Program:
    """The stage program of one MinkUNet (pure data: built from the module tree, independent of any input)."""

    STRIDES = (1, 2, 4, 8, 16)

    def __init__(self, model):
        from . import mink_unet as mu
        from .minkowski import BasicBlock
        self.ops, self.bufs, self.convs, self.bns, self.map_keys = [], [], [], [], []
        self.relu_bufs = []                      # buffers behind a ReLU, in call order (test hook)
        P, L, E = model.PLANES, model.LAYERS, model.BLOCK.expansion
        if model.BLOCK is not BasicBlock:
            raise NotImplementedError("the executor compiles BasicBlock networks (every shipped MinkUNet variant)")
        init = model.INIT_DIM
        skip_width = (P[2] * E, P[1] * E, P[0] * E, init)
        # cat buffers of the four decoder stages: [up-branch | encoder skip] at levels 3, 2, 1, 0
        cat = [self._buf(3 - i, P[4 + i] + skip_width[i]) for i in range(4)]
        skip_copy = {3 - i: (cat[i], P[4 + i]) for i in range(4)}            # encoder level -> (cat buffer, first column)
        x = self._stage(model.conv0p1s1, model.bn0, -1, 0, 0, relu=True, need_dgrad=False, copy=skip_copy[0])
        for i in range(4):
            x = self._stage(getattr(model, mu._DOWN[i]), getattr(model, "bn%d" % (i + 1)), x, i, i + 1, relu=True)
            x = self._blocks(getattr(model, "block%d" % (i + 1)), x, i + 1, copy=skip_copy.get(i + 1))
        for i in range(4):
            lvl = 4 - i
            self._stage(getattr(model, mu._UP[i]), getattr(model, "bntr%d" % (4 + i)), x, lvl, lvl - 1, relu=True, transposed=True,
                        copy=(cat[i], 0))
            x = self._blocks(getattr(model, "block%d" % (5 + i)), cat[i], lvl - 1, copy=None)
        self.feature_buf = x
        self._stage(model.final, None, x, 0, 0, relu=False)
        self.n_levels = 5
        # maps that some tile-list-eligible convolution runs on get tile / pair lists (the 3-channel stem's 5^3 map does not)
        self.map_lists = [any(o["map"] == mi and ops.tl_eligible(o["K"], o["cin"], o["cout"]) for o in self.ops)
                          for mi in range(len(self.map_keys))]
        self.op_arr = np.zeros(len(self.ops), dtype=_OP)
        for i, o in enumerate(self.ops):
            self.op_arr[i] = tuple(o[n] for n in _OP.names)
        self.buf_arr = np.array(self.bufs, dtype=_BUF)
        self.params = [c.kernel for c in self.convs]
        for b in self.bns:
            self.params += [b.bn.weight, b.bn.bias]

    def _buf(self, level, channels):
        self.bufs.append((level, channels))
        return len(self.bufs) - 1

    def _map(self, conv, lvl_in, lvl_out, transposed):
        if conv.kernel_volume == 1:
            return -1
        s_in, s_out = self.STRIDES[lvl_in], self.STRIDES[lvl_out]
        key = (s_out, s_in, conv.kernel_size, conv.dilation) if transposed else (s_in, s_out, conv.kernel_size, conv.dilation)
        if key not in self.map_keys:
            self.map_keys.append(key)
        return self.map_keys.index(key)

    def _stage(self, conv, norm, src, lvl_in, lvl_out, relu, res=-1, transposed=False, copy=None, need_dgrad=True):
        from .minkowski import MinkowskiConvolutionTranspose
        if conv.bias is not None:
            raise NotImplementedError("the executor compiles bias-free convolutions (the reference builds no other)")
        if isinstance(conv, MinkowskiConvolutionTranspose) != bool(transposed) or conv.dilation != 1:
            raise NotImplementedError("unexpected convolution type / dilation in the MinkUNet tree")
        dst = self._buf(lvl_out, conv.out_channels) if norm is not None else -1
        self.convs.append(conv)
        bn = -1
        if norm is not None:
            self.bns.append(norm)
            bn = len(self.bns) - 1
        self.ops.append(dict(K=conv.kernel_volume, cin=conv.in_channels, cout=conv.out_channels, lvl_in=lvl_in, lvl_out=lvl_out,
                             map=self._map(conv, lvl_in, lvl_out, transposed), transposed=int(transposed), src=src, dst=dst, bn=bn,
                             relu=int(relu), res=res, copy_buf=copy[0] if copy else -1, copy_col=copy[1] if copy else 0,
                             weight=len(self.convs) - 1, need_dgrad=int(need_dgrad),
                             fine_unique=int(conv.kernel_size == 2 and conv.stride == 2 and conv.dilation == 1), reserved=0))
        if relu and dst >= 0:
            self.relu_bufs.append(dst)
        return dst

    def _blocks(self, stage, x, lvl, copy):
        from .minkowski import MinkowskiBatchNorm
        blocks = list(stage)
        for bi, blk in enumerate(blocks):
            y = self._stage(blk.conv1, blk.norm1, x, lvl, lvl, relu=True)
            res = x
            ds = blk.downsample
            if ds is not None:
                if not (isinstance(ds, torch.nn.Sequential) and len(ds) == 2 and isinstance(ds[1], MinkowskiBatchNorm)):
                    raise NotImplementedError("BasicBlock shortcut is not conv + batch norm")
                res = self._stage(ds[0], ds[1], x, lvl, lvl, relu=False)
            x = self._stage(blk.conv2, blk.norm2, y, lvl, lvl, relu=True, res=res,
                            copy=copy if bi == len(blocks) - 1 else None)
        return x


class _PassState:
    """What a forward pass leaves for its backward pass (kept alive by the autograd node)."""
    __slots__ = ("rows", "maps", "weights", "bns", "arena", "keep", "training", "feats", "plan_fwd_bytes", "cm")


class UNetExecutor:
    def __init__(self, model):
        self.program = Program(model)
        p = self.program
        self.desc = _Desc(len(p.ops), len(p.bufs), len(p.bns), len(p.convs), len(p.map_keys), p.n_levels, 0, 0, 0, 0,
                          _ptr(p.op_arr), _ptr(p.buf_arr))
        n = len(p.ops)
        self._x_off = np.zeros(n, np.uint64)
        self._stat_off = np.zeros(n, np.uint64)
        self._y_off = np.zeros(len(p.bufs), np.uint64)
        self._kf = np.zeros(n, np.int32)
        self._kd = np.zeros(n, np.int32)
        self._kw = np.zeros(n, np.int32)
        self._img = np.zeros(n, np.int32)
        self._plan = _Plan(0, 0, 0, _ptr(self._x_off), _ptr(self._stat_off), _ptr(self._y_off), _ptr(self._kf), _ptr(self._kd),
                           _ptr(self._kw), _ptr(self._img))
        self._rows = np.zeros(8, np.int64)
        self.prof = None                  # osn_prof_t* (bench.py), or None
        # gradient exchange (openscene_amd.distributed): hook(flat_gradients, first_float, end_float, last) is called after each
        # of `grad_segments` segments of the backward pass with the slice of convolution weight gradients that segment finished
        self.grad_ready_hook = None
        self.grad_segments = 4
        self._cuts = None
        self._events = {}                 # device index -> osn_events_t* (fork / join of the backward pass)
        # marshalling caches (round 6): the weight / batch-norm descriptor arrays of a pass are pure functions of the parameters'
        # addresses and versions -- rebuilt only when the fingerprint of a pass differs from the last one's (see _fingerprint)
        self._bn_mods = None
        self._w_cache = None              # (key, array, keep-alive list)
        self._bn_cache = None             # (key, array)
        self._flips = None
        # gradient layout: one flat fp32 buffer, every parameter's slice starts on a 16-byte boundary
        self.grad_off, off = [], 0
        for prm in p.params:
            self.grad_off.append(off)
            off += (prm.numel() + 3) // 4 * 4
        self.grad_total = off
        self.conv_grad_end = self.grad_off[len(p.convs)] if len(self.grad_off) > len(p.convs) else off   # end of the kernels' region

    # -------------------------------------------------------------------------------------------- eligibility
    def usable(self, x, model=None):
        from . import functional as F_
        if not ENABLED or F_.CONV_MODE != "tl":
            return False
        f = x.F
        if not ((f.is_cuda or _DRY_RUN) and f.dtype == torch.float32 and x.tensor_stride == 1):
            return False
        p = self.program
        if f.shape[1] != p.convs[0].in_channels:
            return False
        if f.requires_grad and torch.is_grad_enabled():
            return False                    # the node has no input-feature gradient (op 0 is planned without one): module path
        training = bool(model.training) if model is not None else None
        dev = f.device
        for m, b in zip(p.bns, self._bn_modules()):
            if training is not None and (bool(b.training) != training or bool(m.training) != training):
                return False                # per-BN flags (frozen statistics): the executor applies ONE flag to every BN
            if b.momentum is None or not b.affine or not b.track_running_stats or b._parameters["weight"].device != dev:
                return False
        f32 = torch.float32
        for w in p.params[:len(p.convs)]:
            if w.device != dev or w.dtype != f32:
                return False
        return True

    def _bn_modules(self):
        """The torch BatchNorm1d inside every MinkowskiBatchNorm, looked up once (nn.Module.__getattr__ is the slow path of every
        `m.bn` / `b.weight`: 480 calls per pass before round 6); a replaced sub-module is noticed by identity."""
        mods = self._bn_mods
        bns = self.program.bns
        if mods is None or any(m._modules.get("bn") is not b for m, b in zip(bns, mods)):
            mods = self._bn_mods = [m.bn for m in bns]
        return mods

    def _fingerprint(self):
        """What the descriptor arrays depend on: the address of every parameter and running statistic, the version of every kernel (its
        weight images), eps / momentum of every norm.  ~350 C-level reads (tens of microseconds) against rebuilding two numpy record
        arrays through ~500 module attribute look-ups."""
        p = self.program
        nc = len(p.convs)
        fp = [t.data_ptr() for t in p.params]
        fp += [t._version for t in p.params[:nc]]
        for b in self._bn_modules():
            d = b._buffers
            rm, rv = d["running_mean"], d["running_var"]
            fp += (rm.data_ptr() if rm is not None else 0, rv.data_ptr() if rv is not None else 0, b.eps, b.momentum)
        return tuple(fp)

    # -------------------------------------------------------------------------------------------- marshalling
    def _plan_query(self, lib, rows, training):
        from . import functional as F_
        self.desc.tl_min_rows = int(F_.TL_FWD_MIN_ROWS)
        self.desc.tl_mid_rows = int(F_.TL_MID_MIN_ROWS)
        self.desc.ws_max_rows = int(F_.WS_MAX_ROWS)
        self._rows[:len(rows)] = rows
        check(lib.osn_net_plan_query(ctypes.addressof(self.desc), _ptr(self._rows), int(training), ctypes.addressof(self._plan)),
              "osn_net_plan_query")

    def _maps(self, cm, training):
        p = self.program
        arr = np.zeros(max(len(p.map_keys), 1), dtype=_MAP)
        keep = []
        dp = lambda t: t.data_ptr() if t is not None else 0
        # pair arrays: every map with lists in training (weight gradients); in inference those a weight-stationary
        # forward launch reads (the plan of this pass is in self._kf)
        ws_maps = {int(p.op_arr[j]["map"]) for j in range(len(p.ops)) if int(self._kf[j]) in K_WS}
        for i, (s_in, s_out, k, dil) in enumerate(p.map_keys):
            fwd, bwd, flip = cm.kmap(s_in, s_out, k, dil)
            tf, tb = cm.kmap_tiles(s_in, s_out, k, dil)
            counts = cm.kmap_counts(s_in, s_out, k, dil)
            lf, lb = cm.kmap_lists(s_in, s_out, k, dil) if p.map_lists[i] else (None, None)
            a = arr[i]
            a["nbr_fwd"], a["nbr_bwd"], a["flip"], a["K"] = dp(fwd), dp(bwd), int(bool(flip)), k ** 3
            if tf is not None:
                a["tiles_fwd_rows"], a["tiles_fwd_tbl"], a["tiles_fwd_gmask"] = dp(tf[0]), dp(tf[1]), dp(tf[2])
            if tb is not None:
                a["tiles_bwd_rows"], a["tiles_bwd_tbl"], a["tiles_bwd_gmask"] = dp(tb[0]), dp(tb[1]), dp(tb[2])
            a["counts"] = dp(counts)
            if lf is not None:
                a["tl_fwd"], a["tl_fwd_rows"], a["tl_fwd_bm"] = dp(lf.buf), dp(lf.out_rows), lf.bm
                if training or i in ws_maps:
                    a["pl_fwd"] = dp(ops.pair_lists(lf))
            if lb is not None:
                a["tl_bwd"], a["tl_bwd_rows"], a["tl_bwd_bm"] = dp(lb.buf), dp(lb.out_rows), lb.bm
            keep.append((fwd, bwd, tf, tb, counts, lf, lb))
        return arr, keep

    def _weights(self, cm, need_images, fp=None, mutable=False):
        """fp (a _fingerprint): reuse the last pass's array when nothing it depends on has changed (mutable: a private copy -- the
        backward pass writes the gradient addresses into it)."""
        p = self.program
        key = None
        if fp is not None and ops.WEIGHT_CACHE:
            key = (fp, need_images.tobytes(), cm.device, ops.WEIGHT_CACHE_GENERATION)
            hit = self._w_cache
            if hit is not None and hit[0] == key:
                return (hit[1].copy() if mutable else hit[1]), hit[2]
        arr = np.zeros(len(p.convs), dtype=_WEIGHT)
        keep = []
        if self._flips is None:
            # (static per map key: the map of an odd stride-1 kernel is its own mirror, sparse.CoordinateManager.kmap)
            self._flips = {i: bool(cm.kmap(s_in, s_out, k, dil)[2]) for i, (s_in, s_out, k, dil) in enumerate(p.map_keys)}
        flips = self._flips
        for i, w in enumerate(p.params[:len(p.convs)]):
            a = arr[i]
            a["W"] = w.data_ptr()
            bits = int(need_images[i])
            flip = flips.get(int(p.op_arr[i]["map"]), False)
            if bits & IMG_X6_FWD:
                t = ops.weight_image(w, False, False, ops.PREP_X6); a["x6_fwd"] = t.data_ptr(); keep.append(t)
            if bits & IMG_TL_FWD:
                t = ops.weight_image(w, False, False, ops.PREP_TL); a["tl_fwd"] = t.data_ptr(); keep.append(t)
            if bits & IMG_X6_DGRAD:
                t = ops.weight_image(w, flip, True, ops.PREP_X6); a["x6_dgrad"] = t.data_ptr(); keep.append(t)
            if bits & IMG_TL_DGRAD:
                t = ops.weight_image(w, flip, True, ops.PREP_TL); a["tl_dgrad"] = t.data_ptr(); keep.append(t)
        if key is not None:
            self._w_cache = (key, arr, keep)
            return (arr.copy() if mutable else arr), keep
        return arr, keep

    def _bns(self, training, fp=None, mutable=False):
        p = self.program
        if fp is not None:
            hit = self._bn_cache
            if hit is not None and hit[0] == fp:
                return hit[1].copy() if mutable else hit[1]
        arr = np.zeros(max(len(p.bns), 1), dtype=_BN)
        for i, b in enumerate(self._bn_modules()):
            a = arr[i]
            a["gamma"], a["beta"] = b.weight.data_ptr(), b.bias.data_ptr()
            a["running_mean"], a["running_var"] = b.running_mean.data_ptr(), b.running_var.data_ptr()
            a["eps"], a["momentum"] = b.eps, b.momentum
        if fp is not None:
            self._bn_cache = (fp, arr)
            return arr.copy() if mutable else arr
        return arr

    # -------------------------------------------------------------------------------------------- passes
    def forward(self, model, x, features_only=False, rows=None):
        """-> the network output [N_0, out_channels] (or, features_only, the input of the final 1x1 conv).
        rows (int64 indices, distinct, in range): -> output[rows] only, [len(rows), out_channels] -- the final 1x1 convolution and, in
        training, both of its gradients run on those rows (run/distill.py:321-322 indexes the output with the supervision mask before
        anything reads it: the other rows of the [N, 768] matrix are never used)."""
        p = self.program
        grad = torch.is_grad_enabled() and (x.F.requires_grad or any(q.requires_grad for q in p.params))
        if grad and features_only:
            return None                                          # training through the folded head: module path
        if rows is not None:
            if rows.dtype != torch.int64 or rows.dim() != 1 or rows.device != x.F.device:
                raise ValueError("rows must be an int64 vector of row indices on the features' device")
            sparse = ROW_SPARSE_HEAD and 0 < rows.shape[0] < x.F.shape[0] and self._head_rows_ok()
            if not sparse:
                return self.forward(model, x).index_select(0, rows)
            if grad:
                return _UNetRowsFunction.apply(self, model, x, rows.contiguous(), *p.params)
            with torch.no_grad():
                feat, st = self._run_forward(model, x, True)
                return self._head_rows(st, feat, rows.contiguous())
        if grad:
            return _UNetFunction.apply(self, model, x, *p.params)
        with torch.no_grad():
            out, st = self._run_forward(model, x, features_only)
        return out

    def _head_rows_ok(self):
        """The final stage is a 1x1 convolution without a batch norm on the 1x1 kernel (every MinkUNet's `final`)."""
        o = self.program.ops[-1]
        return o["K"] == 1 and o["bn"] < 0 and o["dst"] < 0 and ops.dense_eligible(o["cin"], o["cout"]) and ops.dense_eligible(o["cout"], o["cin"])

    def _head_rows(self, st, feat, rows):
        """output[rows] = feat[rows] @ W_final: a row gather and the 1x1 kernel on len(rows) rows."""
        p = self.program
        dev = feat.device
        lib = ops._prep(dev)
        cin, cout = int(p.ops[-1]["cin"]), int(p.ops[-1]["cout"])
        n_sel = int(rows.shape[0])
        in_rows = torch.empty((n_sel, cin), dtype=torch.float32, device=dev)
        out = torch.empty((n_sel, cout), dtype=torch.float32, device=dev)
        image = int(st.weights[len(p.convs) - 1]["tl_fwd"])
        with ops._Dev(dev):
            check(lib.osn_rows_gather(feat.data_ptr(), rows.data_ptr(), n_sel, cin, in_rows.data_ptr(), ops._stream(dev)), "osn_rows_gather")
            check(lib.osn_dense_fwd(in_rows.data_ptr(), image, out.data_ptr(), n_sel, cin, cout, ops._stream(dev)), "osn_dense_fwd")
        return out

    def _run_forward(self, model, x, features_only=False, grad=False):
        p = self.program
        cm = x.coordinate_manager
        feats = ops._f32c(x.F, "features")
        dev = feats.device
        lib = ops._prep(dev)
        training = bool(model.training)
        # (maps that are built HERE are built inside the pass -- inference, or a training step whose call site prefetches nothing,
        #  run/distill.py:315-321 unchanged -- with nothing else of this pass in flight: the map chains on INFER_MAPS_STREAMS streams.
        #  Prefetched maps are in the manager's caches already and nothing is built.)
        cm.prebuild(pairs=True if grad else "ws", streams=INFER_MAPS_STREAMS if (not grad or TRAIN_MAPS_STREAMS) else None)
        rows = [cm.size(s) for s in p.STRIDES]
        if feats.shape[0] != rows[0]:
            raise ValueError("%d feature rows for %d voxels" % (feats.shape[0], rows[0]))
        # sizes for what the C side will lay out: osn_net_forward plans with run.training (batch statistics), which also reserves the
        # backward kernels' scratch -- a training-mode forward under no_grad (grad False, training True) asked for the inference
        # plan's smaller workspace and failed the run's size check whenever the pooled buffer had not already grown
        self._plan_query(lib, rows, grad or training)
        st = _PassState()
        st.rows, st.training, st.feats, st.cm = rows, training, feats, cm
        st.maps, keep_m = self._maps(cm, grad)
        prep_done = None
        if grad and SIDE_STREAM and not _DRY_RUN and K_NAMES[int(self._kf[0])] == "stem":
            main, side_t = torch.cuda.current_stream(dev), ops.side_stream(dev)
            side_t.wait_stream(main)                             # the optimizer step is in the main stream's past
            with ops.on_stream(side_t):
                st.weights, keep_w = self._weights(cm, self._img)
            prep_done = torch.cuda.Event()
            prep_done.record(side_t)
            st.bns = self._bns(training)
        elif grad:
            st.weights, keep_w = self._weights(cm, self._img)
            st.bns = self._bns(training)
        else:
            fp = self._fingerprint()                             # inference: both arrays from the last pass when nothing moved
            st.weights, keep_w = self._weights(cm, self._img, fp)
            st.bns = self._bns(training, fp)
        st.arena = torch.empty(int(self._plan.fwd_arena_bytes), dtype=torch.uint8, device=dev)
        st.plan_fwd_bytes = int(self._plan.fwd_arena_bytes)
        st.keep = (keep_m, keep_w)
        ws = ops._ws(int(self._plan.ws_bytes), dev)
        n_ops = len(p.ops)
        end = n_ops - 1 if features_only else n_ops
        out = None if features_only else torch.empty((rows[0], p.convs[-1].out_channels), dtype=torch.float32, device=dev)
        # (features_only with grad: the rows path -- the plan, the weight images and the arena are the full training pass's, the
        #  final stage is played by _head_rows on the selected rows)
        side, ws2, events = self._side(lib, dev)
        run = _Run(_ptr(self._rows), _ptr(st.maps), _ptr(st.weights), _ptr(st.bns), feats.data_ptr(),
                   out.data_ptr() if out is not None else None, None, st.arena.data_ptr(), st.arena.numel(), None, 0,
                   ws.data_ptr(), ws.numel(), ops.tl_counters(dev).data_ptr(), int(training), 0, end,
                   0 if (BN_EPILOGUE and not grad) else RUN_NO_BN_EPILOGUE, self.prof,      # (a pass with a backward pass keeps x)
                   side if events else None, ws2.data_ptr() if (events and ws2 is not None) else None,
                   ws2.numel() if (events and ws2 is not None) else 0, events, None, None, None, 0)
        with ops._Dev(dev):
            if prep_done is not None and end > 1:
                run.first_op, run.end_op = 0, 1                  # the stem (fp32 kernel, no image) beside the refresh ...
                check(lib.osn_net_forward(ctypes.addressof(self.desc), ctypes.addressof(run), ops._stream(dev)), "osn_net_forward")
                torch.cuda.current_stream(dev).wait_event(prep_done)
                run.first_op, run.end_op = 1, end                # ... everything else behind it
            check(lib.osn_net_forward(ctypes.addressof(self.desc), ctypes.addressof(run), ops._stream(dev)), "osn_net_forward")
        if training:
            torch._foreach_add_([m.bn.num_batches_tracked for m in p.bns], 1)
        from . import functional as F_
        if F_._relu_observer is not None:
            for b in p.relu_bufs:
                F_._relu_observer(self._view(st, b))
        if features_only:
            out = self._view(st, p.feature_buf)
        return out, st

    def _side(self, lib, dev):
        """(raw side stream, its scratch buffer, event pool) or (None, None, None)."""
        if not SIDE_STREAM:
            return None, None, None
        side = 0x51DE if _DRY_RUN else ops.side_stream(dev).cuda_stream       # (dry run: any handle but the main stream's)
        ws2 = ops.ws_on(int(self._plan.ws_bytes), dev, side)
        events = self._events.get(ops._idx(dev))
        if events is None:
            with ops._Dev(dev):
                events = self._events[ops._idx(dev)] = lib.osn_events_create(2 * len(self.program.ops) + 2)
        if not events:
            return None, None, None
        return side, ws2, events

    def _view(self, st, buf):
        p = self.program
        lvl, ch = p.bufs[buf]
        n = st.rows[lvl]
        off = int(self._y_off[buf])
        return st.arena[off:off + n * ch * 4].view(torch.float32).view(n, ch)

    def _run_backward(self, st, gout, rows_direct=None):
        """rows_direct = (pos int32 [N], idx int64 [n_sel], gradient rows float32 [n_sel, cout]): the output gradient exists only on
        those rows (the forward pass ran the head on them: forward(rows=...)); gout is then None."""
        p = self.program
        dev = gout.device if gout is not None else rows_direct[2].device
        lib = ops._prep(dev)
        gout_in = gout
        gout = ops._f32c(gout, "grad_output") if gout is not None else None
        self._plan_query(lib, st.rows, True)
        if int(self._plan.fwd_arena_bytes) != st.plan_fwd_bytes:
            raise RuntimeError("the executor's arena layout changed between a forward pass and its backward pass")
        grads = torch.empty(self.grad_total, dtype=torch.float32, device=dev)
        base = grads.data_ptr()
        nc = len(p.convs)
        st.weights["gW"] = base + 4 * np.asarray(self.grad_off[:nc], dtype=np.uint64)
        if p.bns:
            st.bns["ggamma"][:len(p.bns)] = base + 4 * np.asarray(self.grad_off[nc::2], dtype=np.uint64)
            st.bns["gbeta"][:len(p.bns)] = base + 4 * np.asarray(self.grad_off[nc + 1::2], dtype=np.uint64)
        barena = torch.empty(int(self._plan.bwd_arena_bytes), dtype=torch.uint8, device=dev)
        ws = ops._ws(int(self._plan.ws_bytes), dev)
        self._rows[:len(st.rows)] = st.rows
        side, ws2, events = self._side(lib, dev)
        # a gradient whose producer knows its non-zero rows (openscene_amd.losses.distill_loss: the loss sees `output[sel]`
        # only): the head's two gradients run on those rows.  The hint travels as an attribute of the gradient tensor and is
        # honoured only when it describes exactly this tensor IN THE STATE the loss left it (a tensor hook that edits the
        # gradient in place moves its version counter: the compacted rows would be stale, the dense gradient is used).
        rows_pos = rows_idx = rows_g = None
        n_rows = 0
        hint = getattr(gout_in, "_osn_rows", None) if gout_in is not None else None
        if rows_direct is not None:
            rows_pos, rows_idx, rows_g = rows_direct[0].data_ptr(), rows_direct[1].data_ptr(), rows_direct[2].data_ptr()
            n_rows = int(rows_direct[1].shape[0])
        elif (ROW_SPARSE_HEAD and hint is not None and hint["ptr"] == gout.data_ptr() and hint["shape"] == tuple(gout.shape)
                and hint.get("version") == gout_in._version and 0 < hint["idx"].shape[0] < gout.shape[0] and hint["rows"].device == dev):
            rows_pos, rows_idx, rows_g = hint["pos_ptr"], hint["idx"].data_ptr(), hint["rows"].data_ptr()
            n_rows = int(hint["idx"].shape[0])
            keep_hint = hint                      # (the tensors stay referenced until the launches are queued)
        run = _Run(_ptr(self._rows), _ptr(st.maps), _ptr(st.weights), _ptr(st.bns), st.feats.data_ptr(), None,
                   gout.data_ptr() if gout is not None else None,
                   st.arena.data_ptr(), st.arena.numel(), barena.data_ptr(), barena.numel(), ws.data_ptr(), ws.numel(),
                   ops.tl_counters(dev).data_ptr(), int(st.training), 0, len(p.ops), 0, self.prof,
                   side if events else None, ws2.data_ptr() if (events and ws2 is not None) else None,
                   ws2.numel() if (events and ws2 is not None) else 0, events, rows_pos, rows_idx, rows_g, n_rows)
        hook = self.grad_ready_hook
        with ops._Dev(dev):
            if hook is None:
                check(lib.osn_net_backward(ctypes.addressof(self.desc), ctypes.addressof(run), ops._stream(dev)), "osn_net_backward")
            else:
                # the pass in segments, highest ops first; after each, the weight-gradient slices of its convolutions are final
                # (every segment ends with the join of the weight-gradient stream) and the hook may start exchanging them
                # An inner segment does not join the weight-gradient stream into the main stream (RUN_NO_JOIN): its slices are
                # final in that stream's order, so its hook runs with THAT stream current -- a collective queues behind what the
                # current stream holds -- and the main stream goes on with the next segment.  The last segment joins everything.
                hi = len(p.ops)
                forked = bool(run.side_stream) and bool(run.events) and not _DRY_RUN
                for lo in self.backward_cuts():
                    run.first_op, run.end_op = lo, hi
                    inner = forked and lo > 0
                    run.flags = RUN_NO_JOIN if inner else 0
                    check(lib.osn_net_backward(ctypes.addressof(self.desc), ctypes.addressof(run), ops._stream(dev)), "osn_net_backward")
                    lo_f, hi_f = self.grad_off[lo], self.grad_off[hi] if hi < nc else self.conv_grad_end
                    if inner:
                        with torch.cuda.stream(ops.side_stream(dev)):
                            hook(grads, lo_f, hi_f, False)
                    else:
                        hook(grads, lo_f, hi_f, lo == 0)
                    hi = lo
                run.flags = 0
        return [grads[o:o + q.numel()].view_as(q) for o, q in zip(self.grad_off, p.params)]

    def backward_cuts(self):
        """First ops of the backward segments (descending, ending with 0): about equal shares of the convolution weights."""
        if self._cuts is None:
            p = self.program
            nc = len(p.convs)
            sizes = [c.kernel.numel() for c in p.convs]
            total, target = float(sum(sizes)), float(sum(sizes)) / max(1, self.grad_segments)
            cuts, acc = [], 0.0
            for i in range(nc - 1, 0, -1):
                acc += sizes[i]
                if acc >= target and len(cuts) < self.grad_segments - 1:
                    cuts.append(i)
                    acc = 0.0
            self._cuts = cuts + [0]
        return self._cuts

    def kernels(self, rows, training=True):
        """[(op index, forward kernel, input-gradient kernel, weight-gradient kernel)] names for the given level sizes."""
        self._plan_query(_lib.load(), rows, training)
        return [(i, K_NAMES[int(self._kf[i])], K_NAMES[int(self._kd[i])], K_NAMES[int(self._kw[i])]) for i in range(len(self.program.ops))]


class _UNetFunction(Function):
    """The whole U-Net as ONE autograd node: forward = osn_net_forward, backward = osn_net_backward."""

    @staticmethod
    def forward(ctx, ex, model, x, *params):
        out, st = ex._run_forward(model, x, grad=True)
        ctx.ex, ctx.st = ex, st
        ctx.save_for_backward(*params)           # autograd's version check guards the weight images kept for the backward
        return out

    @staticmethod
    def backward(ctx, gout):
        _ = ctx.saved_tensors                    # raises if a parameter was modified in place since the forward pass
        if ctx.st is None:
            raise RuntimeError("the network executor's node was backpropagated a second time: its activation arena is released "
                               "after the first backward pass (retain_graph is not supported here; set OSN_EXECUTOR=0 for the "
                               "module-by-module path)")
        grads = ctx.ex._run_backward(ctx.st, gout)
        ctx.st = None
        return (None, None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[3:]))


class _UNetRowsFunction(Function):
    """The U-Net with its final 1x1 convolution on SELECTED rows only: forward = osn_net_forward up to the head's input + the head on
    feat[rows]; backward = osn_net_backward with the row-compacted head gradients (in[rows]^T @ g, scatter(g @ W^T))."""

    @staticmethod
    def forward(ctx, ex, model, x, rows, *params):
        feat, st = ex._run_forward(model, x, True, grad=True)
        out = ex._head_rows(st, feat, rows)
        n = feat.shape[0]
        pos = torch.full((n,), -1, dtype=torch.int32, device=feat.device)
        pos[rows] = torch.arange(rows.shape[0], dtype=torch.int32, device=feat.device)
        ctx.ex, ctx.st, ctx.rows, ctx.pos = ex, st, rows, pos
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, gout_rows):
        _ = ctx.saved_tensors
        if ctx.st is None:
            raise RuntimeError("the network executor's node was backpropagated a second time (retain_graph is not supported here)")
        g = ops._f32c(gout_rows, "grad_output")
        grads = ctx.ex._run_backward(ctx.st, None, rows_direct=(ctx.pos, ctx.rows, g))
        ctx.st = None
        return (None, None, None, None) + tuple(gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[4:]))


_EXECUTORS = weakref.WeakKeyDictionary()       # model -> UNetExecutor | None.  NOT stored on the module: the executor holds ctypes
                                                # structures with pointer fields and raw event handles, which would break
                                                # copy.deepcopy(model), pickling and torch.save(model) (EMA copies, mp.spawn)


def for_model(model):
    """The (cached) executor of a MinkUNet module tree -- openscene_amd.mink_unet's or the reference's own
    models/mink_unet.py class built on the MinkowskiEngine alias -- or None when the tree is not compilable."""
    ex = _EXECUTORS.get(model, False)
    if ex is False:
        try:
            ex = UNetExecutor(model)
        except (NotImplementedError, AttributeError, TypeError):
            ex = None
        _EXECUTORS[model] = ex
    return ex
