"""Autograd glue: each Function's forward/backward is one or two C-ABI calls on
the current HIP stream (backward runs on autograd's thread; the library is
stateless so that is safe)."""
import os
import threading

import torch
from torch.autograd import Function

from . import ops

# Kernels / arithmetic of the forward and input-gradient convolutions:
#   "tl"     (default) second-generation kernels: forward / input gradient from per-tile compacted pair lists
#            (weights in registers, output tile in LDS, spconv_tl.hip) on the large maps, weight gradient from
#            per-offset pair arrays on the bf16 MFMA (wgrad_tl.hip) on every map; smaller maps, 1x1 convs and
#            the 3-channel stem take the "bf16x6" forward and the fp32-MFMA weight gradient
#   "bf16x6" output-stationary kernel over the dense neighbour table, three-way bf16 split of both operands,
#            six bf16 MFMAs per product block, fp32 accumulate: fp32-level accuracy
#   "fp32"   the same kernel on v_mfma_f32_32x32x2_f32 (exact fp32 products, 157 TF peak)
CONV_MODE = os.environ.get("OSN_CONV_MODE", "tl")
TL_FWD_MIN_ROWS = 65536
# ... and from this many rows on when both channel counts are at least 96 (measured with the per-width channel chunks of
# round 3, profiles/r03_s8: 48 k rows 96 -> 96 121 us against 134 us, 128 -> 96 143 / 164; 12.9 k rows 128 -> 128 63 / 78,
# 192 -> 128 88 / 104; narrower layers -- 64 -> 64: 39 / 34, 32 -> 32: 67 / 34 -- stay on the output-stationary kernel)
TL_MID_MIN_ROWS = 8192


# Weight-stationary kernel (spconv_ws.hip) for the launches that write at most this many rows (measured, profiles/r03_s9:
# 3 k rows 128 -> 128 34 us against 45, 256 -> 256 69 / 106-122; 700 rows 256 -> 256 30 / 43; a tie at 12.9 k rows; the limit
# 8192 against 4096 / 16384: L235k-34C step 25.5 / 26.2 / 25.6 ms, 8-scene batch 49.5 / 49.8 / 49.7, S100k 9.90 / 9.89 / 10.01), and --
# without a row limit -- for the launches that write the fine side of a 2^3 stride-2 map, where every row has exactly one
# pair and the result rows go straight to the output (100 k rows 96 -> 96: 32 us against 70; 48 k rows 128 -> 96: 20 / 62)
WS_MAX_ROWS = 8192


def ws_kernel(K, c_src, c_dst, n_src, n_dst, fine_unique, dst_fine):
    """"ws_direct" / "ws" / None: the weight-stationary kernel for a launch that gathers n_src rows of c_src channels and
    writes n_dst rows of c_dst channels (same rule as csrc/net.hip:ws_kernel)."""
    if K <= 1 or not ops.tl_eligible(K, c_src, c_dst, n_src):
        return None
    if fine_unique and dst_fine:
        return "ws_direct"
    if 0 < n_dst <= WS_MAX_ROWS:
        return "ws"
    return None


def tl_rows_ok(n_rows, c_a, c_b):
    """Is a table of n_rows rows big enough for the tile-list kernel on a conv between c_a and c_b channels?"""
    return n_rows >= TL_FWD_MIN_ROWS or (n_rows >= TL_MID_MIN_ROWS and min(c_a, c_b) >= 96)
# Backward of a convolution on a map of at most this many rows: the weight gradient (plan + kernel + reduce) runs on an
# auxiliary stream beside the input gradient (+ its reduce); the node forks after `gout` is ready and joins before it
# returns, so nothing outside sees the second stream.  OFF by default (0): measured on S100k (tools/ab_wall.sh, 2 rounds)
# 14.78 / 14.18 ms with 40000 against 14.31 / 14.29 ms without -- the deep levels are bound by the host's launch rate,
# not by the GPU, so there is nothing to overlap; 70000 and 200000 are slower (15.0 ms).  Results are bitwise identical.
WGRAD_OVERLAP_MAX_ROWS = 0


class SparseConvFunction(Function):
    """out[o] = sum_k feats[nbr_fwd[k, o]] @ kernel[k]  ([ME] MinkowskiConvolutionFunction /
    MinkowskiConvolutionTransposeFunction).  kernel: [K, cin, cout], or [cin, cout] when K == 1."""

    @staticmethod
    def forward(ctx, feats, kernel, nbr_fwd, nbr_bwd, flip, n_out, tiles_fwd=None, tiles_bwd=None, counts=None,
                lists_fwd=None, lists_bwd=None, transposed=False, fine_unique=False):
        ctx.save_for_backward(feats, kernel)
        # weight gradient: the pair arrays of the map (a transposed conv runs on the arrays of the strided conv it
        # mirrors = its own input-gradient lists, with the operand roles swapped)
        ctx.wg_lists = (lists_bwd, True) if transposed else (lists_fwd, False)
        ctx.maps = (nbr_fwd, nbr_bwd, bool(flip), tiles_bwd, counts)
        ctx.n_in = feats.shape[0]
        K = 1 if kernel.dim() == 2 else kernel.shape[0]
        cin, cout = kernel.shape[-2], kernel.shape[-1]
        tbl, rows, gm = (tiles_fwd[1], tiles_fwd[0], tiles_fwd[2]) if tiles_fwd is not None else (nbr_fwd, None, None)
        ctx.wp_dgrad = None
        ctx.tl_bwd = None
        ctx.ws_bwd = None
        ctx.dense_bwd = None
        ctx.rg_bwd = None
        # the cached weight images are shared and refreshed in place: remember which version of the kernel the image kept
        # for the backward pass belongs to (a weight changed through .data between forward and backward bypasses autograd's
        # own saved-tensor check)
        ctx.kver = (kernel._version, kernel.data_ptr())
        mode = "bf16x6" if CONV_MODE == "tl" else CONV_MODE
        ctx.stem = CONV_MODE != "fp32" and nbr_fwd is not None and ops.stem_eligible(K, cin, cout)
        if ctx.stem:
            return ops.stem_conv_fwd(feats, kernel, nbr_fwd, n_out)
        if CONV_MODE == "tl" and K == 1 and ops.dense_eligible(cin, cout):
            # 1x1 convs (head, shortcuts): the row-wise matrix-product kernel, forward and input gradient
            if ctx.needs_input_grad[0] and ops.dense_eligible(cout, cin):
                ctx.dense_bwd = ops.weight_image(kernel, False, True, ops.PREP_TL)
            return ops.dense_fwd(feats, ops.weight_image(kernel, False, False, ops.PREP_TL), cout)
        if CONV_MODE == "tl" and K > 1 and ops.tl_eligible(K, cin, cout, ctx.n_in):
            # weight-stationary kernel from the map's pair arrays (those of the strided / self direction: a transposed conv
            # walks them the other way): forward, and -- decided here, launched in backward -- the input gradient, which on
            # a self map (flip) keeps the direction and takes the mirrored weight image
            pl, swap_f = ctx.wg_lists
            ws_f = ws_kernel(K, cin, cout, ctx.n_in, n_out, fine_unique, transposed) if pl is not None else None
            ws_b = (ws_kernel(K, cout, cin, n_out, ctx.n_in, fine_unique, not transposed)
                    if pl is not None and ctx.needs_input_grad[0] and ops.tl_eligible(K, cout, cin, n_out) else None)
            # forward / input gradient: only on maps of at least TL_FWD_MIN_ROWS rows (measured: 20-30 % faster on the
            # 100 k-row maps, a tie at 48 k rows, slower below); 1x1 convs stay on the first-generation kernel
            # (a layer with 32 channels on one side goes to the register-gather kernel on ANY map size: rg_first, same rule as csrc/net.hip)
            rg_first_f = min(cin, cout) == 32 and ops.rg_eligible(K, cin, cout, ctx.n_in)
            rg_first_b = min(cin, cout) == 32 and ops.rg_eligible(K, cout, cin, n_out)
            fwd_ok = lists_fwd is not None and tl_rows_ok(n_out, cin, cout) and ws_f != "ws_direct" and not rg_first_f
            bwd_ok = (ctx.needs_input_grad[0] and ops.tl_eligible(K, cout, cin, n_out) and lists_bwd is not None
                      and tl_rows_ok(ctx.n_in, cin, cout) and ws_b != "ws_direct" and not rg_first_b)
            # narrow layers (32 / 64 channels on both sides): the register-gather kernel where neither the tile-list kernel nor a
            # direct weight-stationary launch applies (same rule as csrc/net.hip) -- forward here, input gradient decided here
            rg_f = not fwd_ok and ws_f != "ws_direct" and ops.rg_eligible(K, cin, cout, ctx.n_in)
            rg_b = (ctx.needs_input_grad[0] and not bwd_ok and ws_b != "ws_direct" and ops.rg_eligible(K, cout, cin, n_out))
            if rg_b:
                ctx.rg_bwd = ops.weight_image(kernel, flip, True, ops.PREP_TL)
                ws_b = None
            if rg_f:
                if bwd_ok:
                    ctx.wp_dgrad, ctx.tl_bwd = ops.weight_image(kernel, flip, True, ops.PREP_TL), lists_bwd
                elif ws_b is not None:
                    ctx.ws_bwd = (ops.weight_image(kernel, flip, True, ops.PREP_TL), pl, swap_f if flip else not swap_f, ws_b == "ws_direct")
                return ops.spconv_fwd_rg(feats, ops.weight_image(kernel, False, False, ops.PREP_TL), tbl, n_out, cout, out_rows=rows)
            if ws_b is not None and not bwd_ok:
                ctx.ws_bwd = (ops.weight_image(kernel, flip, True, ops.PREP_TL), pl, swap_f if flip else not swap_f,
                              ws_b == "ws_direct")
            if ws_f is not None and not fwd_ok:
                wf = ops.weight_image(kernel, False, False, ops.PREP_TL)
                if bwd_ok:
                    ctx.wp_dgrad, ctx.tl_bwd = ops.weight_image(kernel, flip, True, ops.PREP_TL), lists_bwd
                return ops.spconv_fwd_ws(feats, wf, pl, nbr_fwd, n_out, K, cout, swap=swap_f, direct=ws_f == "ws_direct")
            # weight images: parameters are served from ops' per-device cache (one launch per optimizer step for the
            # whole model); the input-gradient image is requested here too so that it is part of that launch
            wf = ops.weight_image(kernel, False, False, ops.PREP_TL) if fwd_ok else None
            if bwd_ok:
                ctx.wp_dgrad, ctx.tl_bwd = ops.weight_image(kernel, flip, True, ops.PREP_TL), lists_bwd
            if fwd_ok:
                return ops.spconv_fwd_tl(feats, wf, lists_fwd, n_out, K, cout)
        if mode == "bf16x6" and ops.x6_eligible(K, cin, cout, n_out):
            wp = ops.weight_image(kernel, False, False, ops.PREP_X6)
            if (ctx.needs_input_grad[0] and ctx.tl_bwd is None and ctx.ws_bwd is None and ctx.rg_bwd is None
                    and ops.x6_eligible(K, cout, cin, ctx.n_in)):
                ctx.wp_dgrad = ops.weight_image(kernel, flip, True, ops.PREP_X6)
            return ops.spconv_fwd_x6(feats, wp, tbl, n_out, out_rows=rows, gmask=gm)
        return ops.spconv_fwd(feats, kernel, tbl, n_out, out_rows=rows, gmask=gm)

    @staticmethod
    def backward(ctx, gout):
        feats, kernel = ctx.saved_tensors
        nbr_fwd, nbr_bwd, flip, tiles_bwd, counts = ctx.maps
        gout = gout.contiguous()
        gin = gk = None
        K = 1 if kernel.dim() == 2 else kernel.shape[0]
        if (ctx.wp_dgrad is not None or ctx.ws_bwd is not None or ctx.dense_bwd is not None or ctx.rg_bwd is not None) and ctx.kver != (kernel._version, kernel.data_ptr()):
            raise RuntimeError("a convolution kernel changed between its forward and its backward pass (version %d -> %d): "
                               "the input-gradient weight image kept from the forward is stale" % (ctx.kver[0], kernel._version))

        def weight_grad():
            cin, cout = kernel.shape[-2], kernel.shape[-1]
            tl, swap = ctx.wg_lists
            if ctx.stem:
                return ops.stem_conv_wgrad(feats, gout, nbr_fwd, K).reshape(kernel.shape)
            if CONV_MODE == "tl" and ops.tl_eligible(K, cin, cout, ctx.n_in):
                if K > 1 and tl is not None:
                    return ops.spconv_wgrad_tl(feats, gout, tl, K, swap=swap).reshape(kernel.shape)
                if K == 1 and cin <= 256 and cout <= 256:
                    # 1x1 shortcut convs: the pair-array kernel on the identity map (the 96 -> 768 head stays on the table
                    # kernel: 188 us against 215 us measured); same rule as the network executor (csrc/net.hip)
                    return ops.spconv_wgrad_tl(feats, gout, None, 1).reshape(kernel.shape)
            return ops.spconv_wgrad(feats, gout, nbr_fwd, K, counts).reshape(kernel.shape)

        side = None
        if ctx.needs_input_grad[1]:
            if (ctx.needs_input_grad[0] and gout.is_cuda
                    and max(ctx.n_in, gout.shape[0]) <= WGRAD_OVERLAP_MAX_ROWS):
                main = torch.cuda.current_stream(gout.device)
                side = ops.side_stream(gout.device)
                side.wait_stream(main)                    # fork: gout (and everything before it) is ready
                with ops.on_stream(side):
                    gk = weight_grad()
            else:
                gk = weight_grad()
        if ctx.needs_input_grad[0]:
            cin, cout = kernel.shape[-2], kernel.shape[-1]
            tbl, rows, gm = (tiles_bwd[1], tiles_bwd[0], tiles_bwd[2]) if tiles_bwd is not None else (nbr_bwd, None, None)
            mode = "bf16x6" if CONV_MODE == "tl" else CONV_MODE
            if ctx.dense_bwd is not None:
                gin = ops.dense_fwd(gout, ctx.dense_bwd, cin)
                ctx.dense_bwd = None
            elif ctx.rg_bwd is not None:
                gin = ops.spconv_fwd_rg(gout, ctx.rg_bwd, tbl, ctx.n_in, cin, out_rows=rows)
                ctx.rg_bwd = None
            elif ctx.tl_bwd is not None:
                gin = ops.spconv_fwd_tl(gout, ctx.wp_dgrad, ctx.tl_bwd, ctx.n_in, K, cin)
                ctx.wp_dgrad = ctx.tl_bwd = None
            elif ctx.ws_bwd is not None:
                wb, pl, swap_b, direct = ctx.ws_bwd
                gin = ops.spconv_fwd_ws(gout, wb, pl, nbr_bwd, ctx.n_in, K, cin, swap=swap_b, direct=direct)
                ctx.ws_bwd = None
            elif mode == "bf16x6" and ops.x6_eligible(K, cout, cin, ctx.n_in):
                # (reached only when the forward pass planned this kernel: rg_bwd / tl_bwd / ws_bwd are set otherwise)
                wp = ctx.wp_dgrad if ctx.wp_dgrad is not None else ops.weight_image(kernel, flip, True, ops.PREP_X6)
                ctx.wp_dgrad = None
                gin = ops.spconv_fwd_x6(gout, wp, tbl, ctx.n_in, out_rows=rows, gmask=gm)
            else:
                gin = ops.spconv_fwd(gout, ops.weight_transpose(kernel, flip), tbl, ctx.n_in, out_rows=rows, gmask=gm)
        if side is not None:
            main.wait_stream(side)                        # join: later work on this stream sees the weight gradient
        return gin, gk, None, None, None, None, None, None, None, None, None, None, None


class BatchNormActFunction(Function):
    """y = act(BN(x) [+ residual]) in one pass over x; batch statistics in training
    (running buffers updated in place like torch.nn.BatchNorm1d), running statistics in eval."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, relu):
        x = x.contiguous()
        if training:
            y, mean, var = ops.bn_forward_train(x, gamma, beta, eps, residual, relu, running_mean, running_var, momentum)
        else:
            mean, var = running_mean, running_var
            y = ops.bn_apply(x, mean, var, gamma, beta, eps, residual, relu)
        # bn -> relu without a residual: the backward pass recomputes the mask (y > 0) from x instead of reading y
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, mean, var, gamma, beta if relu else None)
        ctx.cfg = (bool(training), float(eps), bool(relu), residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, mean, var, gamma, beta = ctx.saved_tensors
        training, eps, relu, has_res = ctx.cfg
        want_gres = has_res and ctx.needs_input_grad[5]
        gx, gres, ggamma, gbeta = ops.bn_backward(x, y, gy.contiguous(), mean, var, gamma, eps, relu, training,
                                                  want_gres, beta=beta if (relu and not has_res) else None)
        return gx, ggamma, gbeta, None, None, gres, None, None, None, None


class ReluFunction(Function):
    """Stand-alone ME.MinkowskiReLU (models/mink_unet.py:114) of the un-fused module chain."""

    @staticmethod
    def forward(ctx, x):
        y = ops.relu_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ops.relu_bwd(y, gy)


class AddFunction(Function):
    """Row-aligned a + b of two feature matrices on one coordinate map (the un-fused residual `out += residual`)."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.add(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


class Cat2Function(Function):
    """ME.cat of two tensors (models/mink_unet.py:147,155,163,171): one launch forward, one launch backward."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.widths = (a.shape[1], b.shape[1])
        return ops.cat2(a, b)

    @staticmethod
    def backward(ctx, g):
        ca, cb = ctx.widths
        if not ctx.needs_input_grad[0] or not ctx.needs_input_grad[1]:
            g = g.contiguous()
            return (g[:, :ca].contiguous() if ctx.needs_input_grad[0] else None,
                    g[:, ca:].contiguous() if ctx.needs_input_grad[1] else None)
        return ops.cat2_bwd(g, ca, cb)


def _hip_eligible(*ts):
    """The elementwise HIP kernels take contiguous float32 device matrices on 16-byte boundaries.  Anything else ON A DEVICE
    (another dtype, a strided view, an odd storage offset) goes to the torch operator of the same name -- MinkowskiEngine
    accepts those too; host tensors are refused by the ops like everywhere else (no CPU fallback)."""
    for t in ts:
        if not t.is_cuda:
            return True                 # -> ops.* raises its usual error
        if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 != 0:
            return False
    return True


def relu(x):
    if not _hip_eligible(x):
        return torch.relu(x)
    return ReluFunction.apply(x)


def add(a, b):
    """Row-aligned sum of two feature matrices on ONE coordinate map (the BasicBlock residual).  Shapes must agree: a sparse add
    never broadcasts (a [N, C] + [1, C] would be a coordinate-map bug turned into numbers)."""
    if a.shape != b.shape:
        raise ValueError("add: feature matrices of shapes %s and %s (tensors on the same coordinate map have equal shapes)"
                         % (tuple(a.shape), tuple(b.shape)))
    if not _hip_eligible(a, b) and a.is_cuda and b.is_cuda:
        return a + b                    # dtype / stride / alignment the kernel does not take
    return AddFunction.apply(a, b)


def cat(ts):
    """Column concat of feature matrices on one coordinate map, left to right: one HIP launch per pair (the
    reference only ever concatenates two tensors).  Widths that are not multiples of 4, other dtypes and unaligned views
    take torch.cat on the device."""
    if all(t.is_cuda for t in ts) and (not _hip_eligible(*ts) or any(t.dim() != 2 or t.shape[1] % 4 or t.shape[1] < 4 for t in ts)):
        return torch.cat(list(ts), dim=1)
    out = ts[0]
    for t in ts[1:]:
        out = Cat2Function.apply(out, t)
    return out


def sparse_conv(feats, kernel, maps, n_out, tiles=None, counts=None, lists=None, transposed=False, fine_unique=False):
    """maps = CoordinateManager.kmap(...); tiles = .kmap_tiles(...) or None; counts = .kmap_counts(...) or None;
    lists = .kmap_lists(...) or None; transposed: the conv is a MinkowskiConvolutionTranspose; fine_unique: the map is a
    2^3 stride-2 map (every row of its fine side has exactly one pair)."""
    nbr_fwd, nbr_bwd, flip = maps
    tf, tb = tiles if tiles is not None else (None, None)
    lf, lb = lists if lists is not None else (None, None)
    return SparseConvFunction.apply(feats, kernel, nbr_fwd, nbr_bwd, flip, n_out, tf, tb, counts, lf, lb, bool(transposed),
                                    bool(fine_unique))


_tls = threading.local()


class deferred_bn_counters:
    """Context: collect the `num_batches_tracked += 1` of every BN in a forward pass and apply them as ONE
    multi-tensor add at exit (48 one-element launches -> 1).  Same final buffer values as nn.BatchNorm1d.
    The pending list is per thread (two forwards on two threads do not see each other's counters)."""

    @staticmethod
    def current():
        return getattr(_tls, "pending", None)

    def __enter__(self):
        self.prev = getattr(_tls, "pending", None)
        _tls.pending = self.pending = []
        return self

    def __exit__(self, *exc):
        _tls.pending = self.prev
        if self.pending:
            torch._foreach_add_(self.pending, 1)
        return False


_relu_observer = None


def set_relu_observer(fn):
    """Test hook: `fn(y)` is called with the output of every fused BN(+residual)+ReLU, in call order
    (None switches it off).  Parity tests record the run's activation pattern through it."""
    global _relu_observer
    _relu_observer = fn


def batch_norm_act(x, bn, residual=None, relu=False):
    """`bn` is a torch.nn.BatchNorm1d (the `.bn` of MinkowskiBatchNorm): same parameters,
    buffers and train/eval semantics, computed by the HIP kernels."""
    training = bn.training or (bn.running_mean is None)
    momentum = 0.1 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if bn.momentum is None:                       # cumulative average: needs the count now (host sync)
            bn.num_batches_tracked.add_(1)
            momentum = 1.0 / float(bn.num_batches_tracked)
        elif deferred_bn_counters.current() is not None:
            deferred_bn_counters.current().append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    if not training and rm is None:
        raise RuntimeError("eval-mode batch norm needs running statistics")
    gamma = bn.weight if bn.weight is not None else torch.ones(bn.num_features, device=x.device)
    beta = bn.bias if bn.bias is not None else torch.zeros(bn.num_features, device=x.device)
    y = BatchNormActFunction.apply(x, gamma, beta, rm, rv, residual, training, momentum, bn.eps, relu)
    if relu and _relu_observer is not None:
        _relu_observer(y)
    return y
