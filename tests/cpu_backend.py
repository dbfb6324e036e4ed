"""TEST INFRASTRUCTURE: a CPU restatement of every op in ``openscene_amd.ops`` built on
``oracle/`` + plain torch, with the same signatures and return conventions.

Two uses, both inside tests/ only:
  * the checker of the ``-m gpu`` parity tests (HIP op vs this, same inputs);
  * ``install(monkeypatch)`` swaps it in for ``openscene_amd.ops`` so the HOST logic
    (coordinate manager, autograd wiring, module tree, DDP) can be exercised without a
    GPU.  The product package never imports this file and has no hook for it.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import coords as oc
from oracle import query as oq
from oracle import sparse_ops as so
from oracle import voxelize as ov


class HashTable:
    def __init__(self, coords4):
        self.coords4 = np.asarray(coords4)


def _np(t):
    return t.detach().cpu().numpy()


def coords_unique(coords4, stride=1):
    c = _np(coords4).astype(np.int32)
    q = oc.floor_to_stride(c, stride) if stride > 1 else c
    if q.shape[0] == 0:
        z = torch.zeros(0, dtype=torch.int32)
        return torch.zeros((0, 4), dtype=torch.int32), z, z.clone(), HashTable(q)
    uniq, inv, first = oc.unique_first(q)
    return (torch.from_numpy(uniq.astype(np.int32)), torch.from_numpy(inv.astype(np.int32)),
            torch.from_numpy(first.astype(np.int32)), HashTable(uniq))


def coords_pyramid(coords4, strides=(1, 2, 4, 8, 16)):
    res, cur = [], coords4
    for s in strides:
        r = coords_unique(cur, s)
        res.append(r)
        cur = r[0]
    return res


def kmap_build(table, out_coords4, ksize, offset_scale, with_counts=False, self_map=False):
    # kernel_offsets(ksize, tensor_stride) scales by the tensor stride; dilation folded into offset_scale
    off = oc.kernel_offsets(ksize, offset_scale)
    nbr = torch.from_numpy(oc.kernel_map(table.coords4, _np(out_coords4), off))
    return (nbr, (nbr >= 0).sum(1).long()) if with_counts else nbr


def kmap_transpose(nbr, n_in):
    return torch.from_numpy(oc.transpose_table(_np(nbr), int(n_in)))


def kmap_sort(nbr, counts=None):
    """Spec of osn_kmap_sort: stable argsort of the occupancy key; key bit of offset k = its rank by pair
    count (most frequent = bit 0, rarest = bit K-1, ties: lower k first) or k itself without counts."""
    K = nbr.shape[0]
    bitpos = torch.arange(K)
    if counts is not None:
        c = [int(v) for v in counts]
        bitpos = torch.tensor([sum(1 for j in range(K) if j != k and (c[j] > c[k] or (c[j] == c[k] and j < k)))
                               for k in range(K)])
    key = ((nbr >= 0).long() << bitpos.reshape(K, 1)).sum(0)
    order = torch.from_numpy(np.argsort(key.numpy(), kind="stable").astype(np.int32))
    mask = ((nbr >= 0).long() << torch.arange(K).reshape(K, 1)).sum(0)
    ms = mask[order.long()]
    pad = (-ms.shape[0]) % 32
    g = torch.cat([ms, ms.new_zeros(pad)]).reshape(-1, 32)
    gmask = g[:, 0].clone()
    for j in range(1, 32):
        gmask |= g[:, j]
    return order, nbr[:, order.long()].contiguous(), gmask.int()


def kmap_count(nbr):
    return (nbr >= 0).sum(1).long()


def _w3(w):
    return w.unsqueeze(0) if w.dim() == 2 else w


def spconv_fwd(feats, weight, nbr, n_out, out_rows=None, gmask=None):
    w = _w3(weight)
    if nbr is None:
        nbr = torch.arange(n_out, dtype=torch.int32)[None]
    out = so.sparse_conv(feats, w, _np(nbr))
    if out_rows is not None:
        res = torch.zeros_like(out)
        res[out_rows.long()] = out
        out = res
    return out


def weight_prep_x6(weight, flip=False, for_dgrad=False):
    w = _w3(weight)
    if flip:
        w = torch.flip(w, dims=[0])
    return ("x6", w if not for_dgrad else w.transpose(1, 2).contiguous())


def spconv_fwd_x6(feats, wp, nbr, n_out, out_rows=None, gmask=None):
    return spconv_fwd(feats, wp[1], nbr, n_out, out_rows=out_rows)


class TileLists:
    """Spec of osn_tile_lists_build: per tile of `bm` consecutive table rows and per offset, the valid
    (input row, local output row) pairs in ascending local row."""

    def __init__(self, cnt, lst, bm, n_out, K, out_rows):
        self.cnt, self.lst, self.bm, self.n_out, self.K, self.out_rows = cnt, lst, bm, n_out, K, out_rows
        self.pairs = None

    @property
    def n_tiles(self):
        return -(-self.n_out // self.bm)

    def counts(self):
        return self.cnt

    def lists(self):
        return self.lst


def tile_rows(n_out):
    """Spec of osn_tile_rows: about two rounds of 768 workgroups (three per CU), 32 .. 64 rows, a multiple of 8."""
    bm = -(-max(int(n_out), 1) // 1536)
    bm = (bm + 7) // 8 * 8
    return max(32, min(64, bm))


def tile_lists(nbr, out_rows=None, bm=None):
    t = _np(nbr)
    K, n_out = t.shape
    bm = tile_rows(n_out) if bm is None else int(bm)
    nt = -(-n_out // bm)
    cnt = np.zeros((nt, K), np.int32)
    lst = np.full((nt, K, bm, 2), -7, np.int32)            # entries past cnt are undefined in the product
    for ti in range(nt):
        blk = t[:, ti * bm:(ti + 1) * bm]
        for k in range(K):
            j = np.nonzero(blk[k] >= 0)[0]
            cnt[ti, k] = j.size
            lst[ti, k, :j.size, 0] = blk[k, j]
            lst[ti, k, :j.size, 1] = j
    return TileLists(torch.from_numpy(cnt), torch.from_numpy(lst), bm, n_out, K, out_rows)


def tl_eligible(K, cin, cout, n_in=0):
    return cin % 4 == 0 and cin >= 8 and cout % 4 == 0 and K <= 128 and n_in <= (1 << 24)


def weight_prep_tl(weight, flip=False, want_fwd=True, want_dgrad=True):
    w = _w3(weight)
    wb = torch.flip(w, dims=[0]) if flip else w
    return (("tl", w) if want_fwd else None), (("tl", wb.transpose(1, 2).contiguous()) if want_dgrad else None)


def spconv_fwd_tl(feats, wp, tl, n_out, K, cout, bn_partial=None):
    """Evaluated FROM THE LISTS (not from the table), so host-logic tests exercise the list semantics."""
    w = wp[1]
    if tl is None:
        res = feats @ w[0]
        bm = tile_rows(n_out)
        table_rows = res
    else:
        cnt, lst, bm = tl.counts().numpy(), tl.lists().numpy(), tl.bm
        tab = feats.new_zeros((n_out, cout))
        for ti in range(tl.n_tiles):
            for k in range(K):
                c = int(cnt[ti, k])
                if c:
                    rows_in = torch.from_numpy(lst[ti, k, :c, 0].astype(np.int64))
                    rows_out = torch.from_numpy(lst[ti, k, :c, 1].astype(np.int64)) + ti * bm
                    tab[rows_out] += feats[rows_in] @ w[k]
        table_rows = tab
        if tl.out_rows is not None:
            res = torch.zeros_like(tab)
            res[tl.out_rows.long()] = tab
        else:
            res = tab
    if bn_partial is not None:
        for ti in range(-(-n_out // bm)):
            blk = table_rows[ti * bm:(ti + 1) * bm].double()
            bn_partial[ti, 0] = blk.sum(0)
            bn_partial[ti, 1] = (blk * blk).sum(0)
    return res


def pair_arrays(tl_or_table, out_rows=None):
    """Spec of osn_pair_lists_build: per offset k, the (input row, output TENSOR row) of every valid table entry in
    table-row order, offsets concatenated; poff[k] = first pair of offset k."""
    if isinstance(tl_or_table, TileLists):
        tl = tl_or_table
        cnt, lst = tl.counts().numpy(), tl.lists().numpy()
        pin, pout, poff = [], [], [0]
        for k in range(tl.K):
            for ti in range(tl.n_tiles):
                c = int(cnt[ti, k])
                rows = lst[ti, k, :c, 1].astype(np.int64) + ti * tl.bm
                pin.append(lst[ti, k, :c, 0])
                pout.append(_np(tl.out_rows)[rows] if tl.out_rows is not None else rows)
            poff.append(poff[-1] + int(cnt[:, k].sum()))
        return (torch.tensor(poff, dtype=torch.int32), torch.from_numpy(np.concatenate(pin).astype(np.int32)),
                torch.from_numpy(np.concatenate(pout).astype(np.int32)))
    raise TypeError("pass a TileLists")


def pair_lists(tl):
    return tl


def spconv_wgrad_tl(feats, gout, tl, K, swap=False):
    cin, cout = feats.shape[1], gout.shape[1]
    gw = feats.new_zeros((K, cin, cout))
    if tl is None:
        gw[0] = feats.t() @ gout
        return gw
    poff, pin, pout = pair_arrays(tl)
    for k in range(K):
        a, b = int(poff[k]), int(poff[k + 1])
        if b > a:
            i, o = pin[a:b].long(), pout[a:b].long()
            gw[k] = (feats[o].t() @ gout[i]) if swap else (feats[i].t() @ gout[o])
    return gw


def spconv_fwd_ws(feats, wp, tl, nbr_dst, n_dst, K, cout, swap=False, direct=False):
    """Spec of osn_spconv_fwd_ws, evaluated FROM THE PAIR ARRAYS: partial[k][dst] = feats[src] @ B[k] for every pair of
    offset k, then the sum over the offsets the destination table has at a row, ascending.  direct: every destination row
    must occur exactly once in the whole map."""
    w = wp[1]
    poff, pin, pout = pair_arrays(tl)
    src, dst = (pout, pin) if swap else (pin, pout)
    out = feats.new_zeros((n_dst, cout))
    seen = np.zeros(n_dst, dtype=np.int64)
    for k in range(K):
        a, b = int(poff[k]), int(poff[k + 1])
        if b > a:
            d = dst[a:b].long()
            assert torch.unique(d).numel() == d.numel(), "a destination row twice in one offset"
            if not direct:
                assert bool((nbr_dst[k][d] >= 0).all()), "a pair the destination table does not know"
            out[d] += feats[src[a:b].long()] @ w[k]
            seen[d.numpy()] += 1
        if not direct:
            assert int((nbr_dst[k] >= 0).sum()) == b - a, "the destination table has entries without a pair"
    if direct:
        assert (seen == 1).all(), "direct mode needs exactly one pair per destination row"
    return out


def rg_eligible(K, cin, cout, n_in):
    """Spec of osn_spconv_fwd_rg_ok."""
    return 1 < K <= 128 and cin in (32, 64) and cout in (32, 64) and 1 <= max(n_in, 1) < (1 << 24) and max(n_in, 1) * cin * 4 < (1 << 31)


def spconv_fwd_rg(feats, wp, nbr, n_out, cout, out_rows=None):
    """Spec of osn_spconv_fwd_rg: the table convolution (B given as a fragment-order image = the [K, c, n] matrix here)."""
    return spconv_fwd(feats, wp[1], nbr, n_out, out_rows=out_rows)


def stem_eligible(K, cin, cout):
    return cin <= 4 and cout == 32 and 1 < K <= 125


def stem_conv_fwd(feats, weight, nbr, n_out):
    return spconv_fwd(feats, weight, nbr, n_out)


def rows_argmax(scores, gather=None):
    labels = scores.argmax(1)
    return labels if gather is None else labels[gather]


def dense_eligible(cin, cout):
    return cin % 4 == 0 and cin >= 8 and cout % 4 == 0


def dense_fwd(feats, wp, cout):
    return feats @ wp[1][0]


def stem_conv_wgrad(feats, gout, nbr, K):
    return spconv_wgrad(feats, gout, nbr, K)


def x6_eligible(K, cin, cout, n_out):
    return cin % 4 == 0 and cin >= 8


def weight_transpose(weight, flip):
    w = _w3(weight)
    if flip:
        w = torch.flip(w, dims=[0])
    return w.transpose(1, 2).contiguous()


def spconv_wgrad(feats, gout, nbr, K, counts=None):
    n_out = gout.shape[0]
    if nbr is None:
        nbr = torch.arange(n_out, dtype=torch.int32)[None]
    nbr = nbr.long()
    gw = feats.new_zeros((K, feats.shape[1], gout.shape[1]))
    for k in range(K):
        o = torch.nonzero(nbr[k] >= 0).reshape(-1)
        if o.numel():
            gw[k] = feats[nbr[k][o]].t() @ gout[o]
    return gw


def bn_stats(x, running_mean=None, running_var=None, momentum=0.1):
    n = x.shape[0]
    mean = x.mean(0)
    var = x.var(0, unbiased=False)
    with torch.no_grad():
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean)
        if running_var is not None:
            unb = var * n / (n - 1) if n > 1 else var
            running_var.mul_(1 - momentum).add_(momentum * unb)
    return mean, var


def bn_forward_train(x, gamma, beta, eps, residual, relu, running_mean, running_var, momentum):
    mean, var = bn_stats(x, running_mean, running_var, momentum)
    return bn_apply(x, mean, var, gamma, beta, eps, residual, relu), mean, var


def bn_apply(x, mean, var, gamma, beta, eps, residual=None, relu=False):
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def bn_backward(x, y, gy, mean, var, gamma, eps, relu, training, want_gres, beta=None):
    if relu and y is None:          # mask recomputed from x (bn -> relu without a residual), as the HIP kernels do
        y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    g = gy * (y > 0).to(gy.dtype) if relu else gy
    invstd = torch.rsqrt(var + eps)
    xhat = (x - mean) * invstd
    sg, sgx = g.sum(0), (g * xhat).sum(0)
    n = x.shape[0]
    if training:
        gx = gamma * invstd * (g - sg / n - xhat * sgx / n)
    else:
        gx = gamma * invstd * g
    return gx, (g.clone() if want_gres else None), sgx, sg


def cosine_query(feats, text_half, gather=None, want_scores=True):
    scores, amax = oq.query(feats, text_half, gather)
    return (scores if want_scores else None), amax


def query_ensemble(feat_distill, feat_fusion, text_half, gather_distill=None, gather_fusion=None, want_scores=True):
    fd = feat_distill if gather_distill is None else feat_distill[gather_distill]
    ff = feat_fusion if gather_fusion is None else feat_fusion[gather_fusion]
    scores, amax, ens = oq.query_ensemble(fd, ff, text_half)
    pf = oq.half_matmul((ff / (ff.norm(dim=-1, keepdim=True) + 1e-5)).half(), text_half)
    pd = oq.half_matmul((fd / (fd.norm(dim=-1, keepdim=True) + 1e-5)).half(), text_half)
    sel = pd.max(dim=-1)[0] < pf.max(dim=-1)[0]
    return (scores if want_scores else None), amax, sel


def voxelize_fnv(xyz, T):
    T = np.asarray(T, dtype=np.float64)
    x = _np(xyz)
    homo = np.hstack((x, np.ones((x.shape[0], 1))))
    grid = np.floor(homo @ T.T[:, :3])
    grid = np.floor(grid - grid.min(0))
    inds, inverse = ov.quantize_first_occurrence(grid)
    return torch.from_numpy(grid), torch.from_numpy(inds), torch.from_numpy(inverse)


def fnv_hash(grid):
    return torch.from_numpy(ov.fnv_keys(_np(grid)).view(np.int64))


def ravel_hash(grid):
    return torch.from_numpy(ov.ravel_keys(_np(grid)).view(np.int64))


def feature_remap(mask_chunk, vox_ind):
    from oracle import loader as ol
    mv, src, ind = ol.remap(_np(mask_chunk).astype(bool), _np(vox_ind))
    return torch.from_numpy(mv), torch.from_numpy(src), torch.from_numpy(ind)


def batch_coords(xyz3, batch_index, out):
    out[:, 0] = batch_index
    out[:, 1:] = xyz3
    return out


def weight_prep_x6_pair(weight, flip=False):
    return weight_prep_x6(weight), weight_prep_x6(weight, flip=flip, for_dgrad=True)


def fusion_project(coords3, world_to_camera, intrinsic4, depth, image_hw, cut_bound, vis_thres):
    from oracle import fusion as of
    k = np.eye(4)
    k[0][0], k[1][1], k[0][2], k[1][2] = intrinsic4
    m = of.compute_mapping(None, _np(coords3), None if depth is None else _np(depth), k,
                           (image_hw[1], image_hw[0]), vis_thres, cut_bound, world_to_camera=world_to_camera)
    return torch.from_numpy(m)


def fusion_accumulate(feat2d, mapping, sum_features, counter, image_hw=None):
    from oracle import fusion as of
    of.accumulate(sum_features, counter.view(-1, 1), feat2d, mapping)


def fusion_finish(sum_features, counter):
    from oracle import fusion as of
    return of.finish(sum_features, counter.view(-1, 1))


def weight_image(weight, flip=False, for_dgrad=False, layout=0):
    """Spec of ops.weight_image (no cache on the CPU: the image is a function of the current weight)."""
    if layout == 1:
        wf, wb = weight_prep_tl(weight, flip, want_fwd=not for_dgrad, want_dgrad=for_dgrad)
        return wb if for_dgrad else wf
    return weight_prep_x6(weight, flip=flip, for_dgrad=for_dgrad)


def relu_fwd(x):
    return torch.relu(x)


def relu_bwd(y, gy):
    return gy * (y > 0).to(gy.dtype)


def add(a, b):
    return a + b


def cat2(a, b):
    return torch.cat([a, b], dim=1)


def cat2_bwd(gout, ca, cb):
    return gout[:, :ca].contiguous(), gout[:, ca:ca + cb].contiguous()


_NAMES = ["relu_fwd", "relu_bwd", "add", "cat2", "cat2_bwd", "fusion_project", "fusion_accumulate", "fusion_finish", "weight_image", "stem_eligible", "stem_conv_fwd", "stem_conv_wgrad", "dense_eligible", "dense_fwd", "rows_argmax", "TileLists", "tile_rows", "tile_lists", "pair_lists", "pair_arrays", "spconv_wgrad_tl", "spconv_fwd_ws", "tl_eligible", "weight_prep_tl", "spconv_fwd_tl", "HashTable", "coords_unique", "coords_pyramid", "kmap_build", "kmap_transpose", "kmap_sort", "kmap_count", "spconv_fwd", "weight_prep_x6", "weight_prep_x6_pair", "spconv_fwd_x6", "x6_eligible", "weight_transpose", "rg_eligible", "spconv_fwd_rg",
          "spconv_wgrad", "bn_stats", "bn_forward_train", "bn_apply", "bn_backward", "cosine_query", "query_ensemble", "voxelize_fnv",
          "fnv_hash", "ravel_hash", "feature_remap", "batch_coords"]


def install(monkeypatch):
    """Swap the CPU restatement in for openscene_amd.ops (host-logic tests only)."""
    import openscene_amd.ops as ops
    import sys
    me = sys.modules[__name__]
    for n in _NAMES:
        monkeypatch.setattr(ops, n, getattr(me, n))
