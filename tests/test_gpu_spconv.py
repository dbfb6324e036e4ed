"""HIP sparse convolution (forward, input gradient, weight gradient) vs the float64
oracle.  Tolerance (stated): fp32 products / fp32 accumulation against float64 ->
max |delta| <= 3e-5 * max |reference| per tensor (observed ~1e-6); kernel maps come
from the oracle so that a conv failure cannot hide behind a map failure."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import sparse_ops as so
from openscene_amd import synthetic as syn

import cpu_backend

pytestmark = pytest.mark.gpu
TOL = 3e-5


def dev():
    return torch.device("cuda", 0)


def close(got, ref, what, tol=TOL):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, "%s shape %s vs %s" % (what, tuple(got.shape), tuple(ref.shape))
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, "%s: max|d|=%.3e, max|ref|=%.3e, rel=%.3e" % (what, err, scale, err / scale)


_CLOUDS = {}


def cloud(kind):
    """-> oracle CoordinateManager over a cached cloud."""
    if kind not in _CLOUDS:
        if kind == "big":          # > 32768 rows: 128-row tiles
            v = syn.shuffled(syn.grid_voxels(syn.room_points(3, n_pts=60000), 0.02), 3)
        elif kind == "mid":        # 4096..32768 rows: 64-row tiles (+ offset split)
            v = syn.shuffled(syn.grid_voxels(syn.room_points(4, n_pts=9000), 0.03), 4)
        else:                      # < 4096 rows: 32-row tiles + offset split
            v = syn.shuffled(syn.grid_voxels(syn.room_points(5, n_pts=1500), 0.08), 5)
        _CLOUDS[kind] = oc.CoordinateManager(syn.batch_coords([v]))
    return _CLOUDS[kind]


CASES = [
    # cloud, (in_stride, out_stride, ksize), cin, cout
    ("big", (1, 1, 3), 32, 32),
    ("big", (1, 1, 3), 96, 96),
    ("big", (1, 1, 3), 128, 96),
    ("big", (1, 1, 3), 64, 128),
    ("big", (1, 1, 5), 3, 32),
    ("big", (1, 1, 1), 96, 768),
    ("big", (1, 1, 1), 96, 20),
    ("big", (1, 1, 1), 96, 21),      # cout % 4 != 0 : scalar weight staging
    ("big", (1, 1, 3), 6, 32),       # cin % 4 != 0, cin > 4 : scalar gather
    ("big", (1, 2, 2), 32, 32),
    ("big", (2, 1, 2), 96, 96),
    ("mid", (1, 1, 3), 64, 64),
    ("mid", (1, 1, 3), 192, 128),
    ("mid", (1, 1, 3), 128, 96),
    ("mid", (1, 2, 2), 64, 64),
    ("mid", (2, 1, 2), 128, 128),
    ("small", (1, 1, 3), 256, 256),
    ("small", (1, 1, 3), 128, 256),
    ("small", (1, 1, 3), 64, 64),
    ("small", (1, 1, 1), 128, 256),
    ("small", (2, 1, 2), 256, 128),
    ("small", (4, 4, 3), 256, 256),
    ("small", (1, 1, 5), 32, 64),    # 125 offsets on an MFMA kernel (the generic offset loops of the list / reduce kernels)
]


RG_CASES = [("big", (1, 1, 3), 32, 32), ("big", (1, 1, 3), 64, 32), ("big", (1, 2, 2), 32, 32), ("mid", (1, 1, 3), 64, 64), ("mid", (1, 1, 3), 32, 64),
            ("mid", (1, 2, 2), 64, 64), ("small", (1, 1, 3), 64, 64), ("small", (1, 1, 5), 32, 64), ("small", (4, 4, 3), 64, 32)]


@pytest.mark.parametrize("mode", ["fp32", "bf16x6", "tl", "ws", "rg"])
@pytest.mark.parametrize("kind,key,cin,cout", CASES + [c for c in RG_CASES if c not in CASES])
def test_forward_and_gradients(kind, key, cin, cout, mode, monkeypatch):
    """All kernel generations / arithmetic modes of the forward / input-gradient convolution meet the SAME
    tolerance: the split-bf16 arithmetic (three bf16 pieces per operand, six MFMAs per product block) is
    fp32-accurate.  "tl" = the tile-list kernel (spconv_tl.hip), fed with lists built from the oracle's table;
    "ws" = the weight-stationary kernel (spconv_ws.hip) on every map size, from the pair arrays of those lists, the
    2^3 stride-2 maps declared as such (=> direct mode for the launches that write their fine side).
    "rg" (round 6) = the register-gather kernel (spconv_rg.hip) on the narrow layers, forward and input gradient, straight from the
    neighbour table (the module path hands it the table it has: plain here)."""
    from openscene_amd import functional as F_
    from openscene_amd import ops
    ws = mode == "ws"
    rg = mode == "rg"
    monkeypatch.setattr(F_, "CONV_MODE", "tl" if (ws or rg) else mode)
    monkeypatch.setattr(F_, "TL_FWD_MIN_ROWS", (1 << 30) if (ws or rg) else 0)   # "tl": every map size goes through the tile-list kernel (split launches on small maps)
    if ws or rg:
        monkeypatch.setattr(F_, "TL_MID_MIN_ROWS", 1 << 30)
        monkeypatch.setattr(F_, "WS_MAX_ROWS", (1 << 30) if ws else 0)
    if ws or mode == "tl":
        monkeypatch.setattr(ops, "rg_eligible", lambda *a: False)        # (the narrow layers too: these modes are about spconv_ws.hip / spconv_tl.hip)
    if rg and not ((kind, key, cin, cout) in RG_CASES and ops.rg_eligible(key[2] ** 3, cin, cout, 1) and ops.rg_eligible(key[2] ** 3, cout, cin, 1)):
        pytest.skip("shape outside the register-gather kernel")
    if mode in ("tl", "ws") and not ops.tl_eligible(key[2] ** 3, cin, cout):
        pytest.skip("shape outside the tile-list kernel (takes the bf16x6 path, tested above)")
    if ws and key[2] == 1:
        pytest.skip("1x1 convs have no pair arrays (identity map)")
    cm = cloud(kind)
    si, so_, k = key
    K = k ** 3
    n_in, n_out = cm.level(si).shape[0], cm.level(so_).shape[0]
    nbr_np = cm.kmap(si, so_, k) if K > 1 else None
    g = torch.Generator().manual_seed(cin * 1000 + cout + K)
    feats = torch.randn(n_in, cin, generator=g)
    w = torch.randn((K, cin, cout) if K > 1 else (cin, cout), generator=g) * (1.0 / np.sqrt(cin * K))
    gout = torch.randn(n_out, cout, generator=g)

    # float64 reference with torch autograd
    f64 = feats.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    ref = so.sparse_conv(f64, w64, nbr_np if nbr_np is not None else np.arange(n_out, dtype=np.int32)[None])
    ref.backward(gout.double())

    d = dev()
    if nbr_np is None:
        maps = (None, None, False)
    else:
        nbr = torch.from_numpy(nbr_np).to(d)
        if si == so_ and k % 2 == 1:
            maps = (nbr, nbr, True)
        else:
            maps = (nbr, torch.from_numpy(oc.transpose_table(nbr_np, n_in)).to(d), False)
    fg = feats.to(d).requires_grad_(True)
    wg = w.to(d).requires_grad_(True)
    lists = None
    rg_calls = []
    if rg:
        real_rg = ops.spconv_fwd_rg
        monkeypatch.setattr(ops, "spconv_fwd_rg", lambda *a, **kw: (rg_calls.append(1), real_rg(*a, **kw))[1])
    if mode in ("tl", "ws", "rg") and K > 1:
        lists = (ops.tile_lists(maps[0]), ops.tile_lists(maps[1]) if maps[1] is not maps[0] else None)
        if lists[1] is None:
            lists = (lists[0], lists[0])
    launches = []
    if ws:
        real = ops.spconv_fwd_ws
        monkeypatch.setattr(ops, "spconv_fwd_ws", lambda *a, **kw: (launches.append(bool(kw.get("direct"))), real(*a, **kw))[1])
    # a conv from the coarse to the fine level runs as the transposed conv of the strided map, as the modules call it
    out = F_.sparse_conv(fg, wg, maps, n_out, lists=lists, transposed=ws and si > so_, fine_unique=ws and k == 2)
    close(out, ref, "forward")
    out.backward(gout.to(d))
    close(fg.grad, f64.grad, "input gradient")
    close(wg.grad, w64.grad, "weight gradient")
    if ws:      # forward and input gradient both ran on the weight-stationary kernel; direct where the fine side is written
        assert launches == ([si > so_, si < so_] if k == 2 else [False, False]), launches
    if rg:      # forward and input gradient both ran on the register-gather kernel
        assert len(rg_calls) == 2, rg_calls


@pytest.mark.parametrize("n_in,n_out,K,cin,cout", [(1, 1, 8, 32, 32), (63, 63, 27, 32, 64), (65, 130, 8, 64, 32), (1000, 257, 27, 64, 64),
                                                   (300, 300, 125, 32, 32), (5000, 4097, 27, 32, 32)])
def test_register_gather_kernel_edge_cases(n_in, n_out, K, cin, cout):
    """spconv_rg.hip on hand-made tables: row counts that are not multiples of the 64-row workgroup, more / fewer input than output rows,
    offsets whose count is not a multiple of the four waves, rows without any neighbour (zeros), an arbitrary row permutation through
    out_rows, bitwise reproducibility; against a float64 product through the same table (3e-5 of the tensor max: the bound of every
    convolution kernel here)."""
    from openscene_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(n_in * 7 + n_out + K)
    nbr = torch.randint(0, n_in, (K, n_out), generator=g, dtype=torch.int32)
    nbr[torch.rand(K, n_out, generator=g) < 0.7] = -1                # ~30 % occupancy
    if n_out > 3:
        nbr[:, 2] = -1                                                # a row without any neighbour
    feats = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) / np.sqrt(cin * K * 0.3)
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        on = nbr[k] >= 0
        ref[on] += feats[nbr[k][on].long()].double() @ w[k].double()
    assert ops.rg_eligible(K, cin, cout, n_in) and not ops.rg_eligible(K, 96, cout, n_in) and not ops.rg_eligible(1, cin, cout, n_in)
    wf, _ = ops.weight_prep_tl(w.to(d), want_dgrad=False)
    out = ops.spconv_fwd_rg(feats.to(d), wf, nbr.to(d), n_out, cout)
    scale = float(ref.abs().max())
    assert float((out.cpu().double() - ref).abs().max()) <= 3e-5 * scale
    if n_out > 3:
        assert float(out[2].abs().max()) == 0.0
    assert torch.equal(out, ops.spconv_fwd_rg(feats.to(d), wf, nbr.to(d), n_out, cout)), "not bitwise reproducible"
    perm = torch.randperm(n_out, generator=g).int()
    out_p = ops.spconv_fwd_rg(feats.to(d), wf, nbr[:, perm.long()].contiguous().to(d), n_out, cout, out_rows=perm.to(d))
    assert torch.equal(out, out_p), "the table's row order changed the result"


@pytest.mark.parametrize("n_in,n_out,K,cin", [(50000, 40001, 125, 3), (33000, 32768, 27, 4), (70000, 70017, 125, 1), (40000, 33333, 8, 2)])
def test_stem_convolution_on_the_matrix_cores(n_in, n_out, K, cin):
    """Round 6: from 32 768 output rows on, the stem convolution (models/mink_unet.py:47-50: 5^3, 3 -> 32) is a matrix product on the MFMA
    units (stem_mfma_fwd_kernel: eight contraction elements per lane = two table entries x four channels, B fragments split from the fp32
    weight in the kernel, split-bf16 arithmetic like every other convolution here).  Against a float64 product through the same table,
    3e-5 of the tensor max (the file's bound; observed ~1e-6); odd offset counts (the padded half of the last offset pair), 1 - 4 input
    channels, a ragged last workgroup, rows without a neighbour (exact zeros), bitwise reproducibility, and agreement with the exact-fp32
    kernel of the smaller maps on the first 20 000 rows of the same table."""
    from openscene_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(n_out + K)
    nbr = torch.randint(0, n_in, (K, n_out), generator=g, dtype=torch.int32)
    nbr[torch.rand(K, n_out, generator=g) < 0.88] = -1                # ~12 % occupancy, the 5^3 map's
    nbr[:, 5] = -1
    feats = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, 32, generator=g) / np.sqrt(cin * K * 0.12)
    ref = torch.zeros(n_out, 32, dtype=torch.float64)
    for k in range(K):
        on = nbr[k] >= 0
        ref[on] += feats[nbr[k][on].long()].double() @ w[k].double()
    assert ops.stem_eligible(K, cin, 32)
    out = ops.stem_conv_fwd(feats.to(d), w.to(d), nbr.to(d), n_out)
    scale = float(ref.abs().max())
    err = float((out.cpu().double() - ref).abs().max())
    assert err <= 3e-5 * scale, "max |d| %.3e of %.3e" % (err, scale)
    assert float(out[5].abs().max()) == 0.0
    assert torch.equal(out, ops.stem_conv_fwd(feats.to(d), w.to(d), nbr.to(d), n_out)), "not bitwise reproducible"
    head = ops.stem_conv_fwd(feats.to(d), w.to(d), nbr[:, :20000].contiguous().to(d), 20000)      # 4096 <= rows < 32768: exact fp32 products
    assert float((out[:20000] - head).abs().max()) <= 3e-5 * scale
    # the weight gradient of the same map (stem_mfma_wgrad_kernel: contraction over the rows, 256 partial gradients in workgroup order)
    gout = torch.randn(n_out, 32, generator=g)
    gw_ref = torch.zeros(K, cin, 32, dtype=torch.float64)
    for k in range(K):
        on = nbr[k] >= 0
        gw_ref[k] = feats[nbr[k][on].long()].double().t() @ gout[on].double()
    gw = ops.stem_conv_wgrad(feats.to(d), gout.to(d), nbr.to(d), K)
    assert gw.shape == (K, cin, 32)
    gerr = float((gw.cpu().double() - gw_ref).abs().max())
    assert gerr <= 3e-5 * float(gw_ref.abs().max()), "weight gradient: max |d| %.3e of %.3e" % (gerr, float(gw_ref.abs().max()))
    assert torch.equal(gw, ops.stem_conv_wgrad(feats.to(d), gout.to(d), nbr.to(d), K)), "weight gradient not bitwise reproducible"


def test_out_rows_indirection_and_determinism():
    from openscene_amd import ops
    cm = cloud("mid")
    nbr_np = cm.kmap(1, 1, 3)
    n = nbr_np.shape[1]
    d = dev()
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(n, 64, generator=g).to(d)
    w = (torch.randn(27, 64, 96, generator=g) * 0.05).to(d)
    nbr = torch.from_numpy(nbr_np).to(d)
    a = ops.spconv_fwd(feats, w, nbr, n)
    b = ops.spconv_fwd(feats, w, nbr, n)
    assert torch.equal(a, b), "forward is not bitwise reproducible"
    perm = torch.randperm(n, generator=g).int().to(d)
    # tile slot j computes the output of row perm[j] and stores it at row perm[j]
    c = ops.spconv_fwd(feats, w, nbr[:, perm.long()].contiguous(), n, out_rows=perm)
    assert torch.equal(a, c), "row order of the tiles changed the result"
    ga = ops.spconv_wgrad(feats, a, nbr, 27)
    gb = ops.spconv_wgrad(feats, a, nbr, 27)
    assert torch.equal(ga, gb), "weight gradient is not bitwise reproducible"
    cnt = ops.kmap_count(nbr)
    gc = ops.spconv_wgrad(feats, a, nbr, 27, cnt)            # pair-count-balanced work items
    gd = ops.spconv_wgrad(feats, a, nbr, 27, cnt)
    assert torch.equal(gc, gd), "balanced weight gradient is not bitwise reproducible"
    assert (gc - ga).abs().max().item() <= 1e-5 * ga.abs().max().item()


@pytest.mark.parametrize("kind,key,bm", [("big", (1, 1, 3), None), ("big", (1, 2, 2), 88), ("mid", (1, 1, 3), 36),
                                         ("small", (2, 1, 2), 32), ("mid", (1, 1, 5), 80)])
def test_tile_lists_match_the_spec(kind, key, bm):
    """osn_tile_lists_build is bit-exact against its numpy specification (tests/cpu_backend.tile_lists),
    also on a tile-ordered table."""
    from openscene_amd import ops
    cm = cloud(kind)
    nbr_np = cm.kmap(*key)
    d = dev()
    nbr = torch.from_numpy(nbr_np).to(d)
    for table in (nbr, ops.kmap_sort(nbr, ops.kmap_count(nbr))[1] if nbr.shape[0] <= 32 else None):
        if table is None:
            continue
        got = ops.tile_lists(table, bm=bm)
        ref = cpu_backend.tile_lists(table.cpu(), bm=bm)
        assert got.bm == ref.bm and got.n_tiles == ref.n_tiles
        cnt, lst = got.counts().cpu(), got.lists().cpu()
        assert torch.equal(cnt, ref.counts())
        valid = torch.arange(got.bm).reshape(1, 1, -1) < cnt.unsqueeze(-1)
        assert torch.equal(lst[valid], ref.lists()[valid])
        assert int(cnt.sum()) == int((table >= 0).sum())


def test_tile_list_kernel_rows_partials_and_determinism():
    """The tile-list kernel on a tile-ordered S100k-class map: features come back in tensor row order through
    `out_rows`, the per-tile batch-norm partial sums add up to the column sums of the output, the result is
    bitwise reproducible, and it agrees with the first-generation kernel to fp32 round-off."""
    from openscene_amd import ops
    cm = cloud("big")
    nbr_np = cm.kmap(1, 1, 3)
    n = nbr_np.shape[1]
    d = dev()
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(n, 128, generator=g).to(d)
    w = (torch.randn(27, 128, 96, generator=g) * 0.05).to(d)
    nbr = torch.from_numpy(nbr_np).to(d)
    order, tbl, gm = ops.kmap_sort(nbr, ops.kmap_count(nbr))
    tl = ops.tile_lists(tbl, out_rows=order)
    wf, wb = ops.weight_prep_tl(w, flip=True)
    part = torch.zeros(tl.n_tiles, 2, 96, dtype=torch.float64, device=d)
    a = ops.spconv_fwd_tl(feats, wf, tl, n, 27, 96, bn_partial=part)
    b = ops.spconv_fwd_tl(feats, wf, tl, n, 27, 96)
    assert torch.equal(a, b), "tile-list forward is not bitwise reproducible"
    ref = ops.spconv_fwd_x6(feats, ops.weight_prep_x6(w), tbl, n, out_rows=order, gmask=gm)
    assert (a - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    plain = ops.spconv_fwd_tl(feats, wf, ops.tile_lists(nbr), n, 27, 96)           # unordered table: other tiles, same sums
    assert (a - plain).abs().max().item() <= 2e-6 * ref.abs().max().item()
    s1, s2 = part[:, 0].sum(0), part[:, 1].sum(0)
    assert (s1 - a.double().sum(0)).abs().max().item() <= 1e-9 * a.double().abs().sum(0).max().item()
    assert (s2 - (a.double() ** 2).sum(0)).abs().max().item() <= 1e-9 * (a.double() ** 2).sum(0).max().item()
    # input gradient through the same lists (odd stride-1 kernel: the map is its own mirror, weights flipped)
    gout = torch.randn(n, 96, generator=g).to(d)
    gi = ops.spconv_fwd_tl(gout, wb, tl, n, 27, 128)
    gi_ref = ops.spconv_fwd_x6(gout, ops.weight_prep_x6(w, flip=True, for_dgrad=True), tbl, n, out_rows=order, gmask=gm)
    assert (gi - gi_ref).abs().max().item() <= 2e-6 * gi_ref.abs().max().item()


WG_CASES = [("big", (1, 1, 3), 96, 96), ("big", (1, 1, 3), 128, 96), ("big", (1, 1, 3), 32, 64), ("mid", (1, 1, 3), 192, 128),
            ("small", (1, 1, 3), 256, 256), ("big", (1, 2, 2), 32, 32), ("big", (2, 1, 2), 96, 96), ("big", (1, 1, 1), 96, 768),
            ("mid", (1, 1, 3), 64, 20)]


@pytest.mark.parametrize("kind,key,cin,cout", WG_CASES)
def test_wgrad_tile_list_kernel_vs_oracle(kind, key, cin, cout):
    """Second-generation weight gradient (pair arrays, split-bf16 MFMA with LDS transpose reads) vs the float64
    oracle: max |delta| <= 3e-5 max |ref| (same bound as the fp32-MFMA kernel); pair arrays bit-exact vs their
    specification; bitwise reproducible.  Transposed convs run on the strided conv's arrays with swapped roles."""
    from openscene_amd import ops
    cm = cloud(kind)
    si, so_, k = key
    K = k ** 3
    n_in, n_out = cm.level(si).shape[0], cm.level(so_).shape[0]
    g = torch.Generator().manual_seed(cin + 7 * cout + K)
    feats = torch.randn(n_in, cin, generator=g)
    gout = torch.randn(n_out, cout, generator=g)
    d = dev()
    if K == 1:
        ref = feats.double().t() @ gout.double()
        got = ops.spconv_wgrad_tl(feats.to(d), gout.to(d), None, 1)
        close(got[0], ref, "weight gradient (identity)")
        return
    nbr_np = cm.kmap(si, so_, k)
    ref = torch.zeros(K, cin, cout, dtype=torch.float64)
    for kk in range(K):
        o = np.nonzero(nbr_np[kk] >= 0)[0]
        ref[kk] = feats.double()[nbr_np[kk, o]].t() @ gout.double()[o]
    swap = so_ < si
    if swap:      # transposed conv: lists of the strided conv it mirrors (table rows = coarse rows = this conv's INPUT rows)
        table = torch.from_numpy(cm.kmap(so_, si, k)).to(d)
    else:
        table = torch.from_numpy(nbr_np).to(d)
    for sort in (False, True):
        if sort and table.shape[1] < 64:
            continue
        if sort:
            order, tbl, _ = ops.kmap_sort(table, ops.kmap_count(table))
            tl = ops.tile_lists(tbl, out_rows=order)
        else:
            tl = ops.tile_lists(table)
        poff, pin, pout = ops.pair_arrays(tl)
        spec = cpu_backend.pair_arrays(cpu_backend.tile_lists(tl_table_cpu(table, tl), out_rows=tl.out_rows.cpu() if tl.out_rows is not None else None, bm=tl.bm))
        assert torch.equal(poff.cpu(), spec[0]) and torch.equal(pin.cpu(), spec[1]) and torch.equal(pout.cpu(), spec[2])
        got = ops.spconv_wgrad_tl(feats.to(d), gout.to(d), tl, K, swap=swap)
        again = ops.spconv_wgrad_tl(feats.to(d), gout.to(d), tl, K, swap=swap)
        assert torch.equal(got, again), "weight gradient is not bitwise reproducible"
        close(got, ref, "weight gradient (sorted=%s)" % sort)


def tl_table_cpu(table, tl):
    """The table the lists were built from (tile-ordered if the lists carry a permutation)."""
    t = table.cpu()
    return t[:, tl.out_rows.cpu().long()] if tl.out_rows is not None else t


def test_cached_work_items_of_a_strided_conv_and_its_transpose():
    """A transposed conv shares the pair counts of the strided conv it mirrors but has another row count:
    the cached weight-gradient work items must not be shared between the two."""
    from openscene_amd import ops
    from openscene_amd.sparse import CoordinateManager
    d = dev()
    v = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)          # S100k: both directions plan 512 items
    cm = CoordinateManager(torch.from_numpy(syn.batch_coords([v])).to(d))
    down = cm.kmap(1, 2, 2)[0]
    up = cm.kmap(2, 1, 2)[0]
    cnt_d, cnt_u = cm.kmap_counts(1, 2, 2), cm.kmap_counts(2, 1, 2)
    assert cnt_d is cnt_u
    g = torch.Generator().manual_seed(5)
    fine = torch.randn(cm.size(1), 32, generator=g).to(d)
    coarse = torch.randn(cm.size(2), 32, generator=g).to(d)
    for _ in range(2):                                                      # second round hits the cache
        a = ops.spconv_wgrad(fine, coarse, down, 8, cnt_d)
        b = ops.spconv_wgrad(coarse, fine, up, 8, cnt_u)
        assert (a - ops.spconv_wgrad(fine, coarse, down, 8)).abs().max().item() <= 1e-5 * a.abs().max().item()
        assert (b - ops.spconv_wgrad(coarse, fine, up, 8)).abs().max().item() <= 1e-5 * b.abs().max().item()
        # the two weight gradients are each other's transposes
        assert (a - b.transpose(1, 2)).abs().max().item() <= 1e-5 * a.abs().max().item()


def test_known_answer_cases():
    """The hand-derivable cases of tests/test_oracle_kat.py, through the HIP kernels."""
    from openscene_amd import ops
    from openscene_amd.sparse import CoordinateManager
    d = dev()
    c = torch.tensor([[0, 5, 6, 7]], dtype=torch.int32, device=d)
    cm = CoordinateManager(c)
    nbr, _, _ = cm.kmap(1, 1, 3)
    assert int((nbr >= 0).sum()) == 1 and int(nbr[13, 0]) == 0
    W = torch.stack([torch.eye(4) * k for k in range(27)]).to(d)
    x = (torch.arange(4.0).reshape(1, 4) + 1).to(d)
    assert torch.equal(ops.spconv_fwd(x, W, nbr, 1), 13 * x)
    # 2x2x2 block -> one coarse voxel; transposed conv returns each child in * W[octant]
    g = np.stack(np.meshgrid(*[np.arange(2)] * 3, indexing="ij"), -1).reshape(-1, 3) + 4
    c = torch.from_numpy(np.concatenate([np.zeros((8, 1), np.int32), g.astype(np.int32)], 1)).to(d)
    cm = CoordinateManager(c)
    down, up, _ = cm.kmap(1, 2, 2)
    assert cm.size(2) == 1 and sorted(down[:, 0].tolist()) == list(range(8))
    W = (torch.arange(8.0).reshape(8, 1, 1) + 1).to(d)
    out = ops.spconv_fwd(torch.tensor([[2.0]], device=d), W, up, 8).cpu()
    for r in range(8):
        k = (g[r, 0] - 4) + 2 * (g[r, 1] - 4) + 4 * (g[r, 2] - 4)
        assert out[r, 0] == 2.0 * (k + 1)


def test_empty_and_bad_arguments():
    from openscene_amd import ops, _lib
    d = dev()
    out = ops.spconv_fwd(torch.zeros((0, 32), device=d), torch.zeros((27, 32, 32), device=d),
                         torch.zeros((27, 0), dtype=torch.int32, device=d), 0)
    assert out.shape == (0, 32)
    with pytest.raises(ValueError):
        ops.spconv_fwd(torch.zeros((4, 16), device=d), torch.zeros((27, 32, 32), device=d),
                       torch.zeros((27, 4), dtype=torch.int32, device=d), 4)
    with pytest.raises(ValueError):
        ops.spconv_fwd(torch.zeros((4, 32), device=d), torch.zeros((27, 32, 32), device=d), None, 4)


def test_weight_prep_x6_is_an_exact_three_way_split():
    """hi + mid + lo reproduces the fp32 weight to <= 2^-24 relative, in both layouts."""
    from openscene_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(3)
    w = torch.randn(27, 40, 24, generator=g).to(d)
    for flip, dgrad in ((False, False), (True, True), (False, True)):
        wp = ops.weight_prep_x6(w, flip=flip, for_dgrad=dgrad).float()
        assert wp.shape == (3, 27, 40 if dgrad else 24, 32 if dgrad else 64)
        rec = wp.sum(0)
        ref = torch.flip(w, dims=[0]) if flip else w
        ref = ref if dgrad else ref.transpose(1, 2)
        nc = ref.shape[2]
        assert torch.all(rec[:, :, nc:] == 0)
        err = (rec[:, :, :nc] - ref).abs().max().item()
        assert err <= 2 ** -23 * ref.abs().max().item()


def test_weight_image_cache_batches_and_follows_the_parameter_version():
    """ops.weight_image: (i) the batched launch writes bit-identical images to the per-weight entry points, for both
    layouts, forward / input gradient, flipped or not, K = 1 [cin, cout] weights included; (ii) an in-place update of
    the parameter (what an optimizer step is) refreshes every image of the device in one go; (iii) an unchanged
    parameter is served without a launch; (iv) the per-step wgrad / dgrad overlap stream does not change results."""
    from openscene_amd import ops
    ops.clear_weight_cache()
    g = torch.Generator().manual_seed(11)
    shapes = [(27, 96, 96), (8, 32, 64), (1, 96, 768), (27, 20, 36), (125, 4, 32)]
    params = []
    for K, cin, cout in shapes:
        w = torch.randn((K, cin, cout) if K > 1 else (cin, cout), generator=g) * 0.1
        params.append(torch.nn.Parameter(w.to(dev())))

    def direct(p, flip, dg, layout):
        if layout == ops.PREP_TL:
            wf, wb = ops.weight_prep_tl(p.detach(), flip, want_fwd=not dg, want_dgrad=dg)
            return wb if dg else wf
        return ops.weight_prep_x6(p.detach(), flip=flip, for_dgrad=dg)

    combos = [(False, False), (False, True), (True, True)]
    for rnd in range(3):
        for p in params:
            for layout in (ops.PREP_X6, ops.PREP_TL):
                for flip, dg in combos:
                    img = ops.weight_image(p, flip, dg, layout)
                    ref = direct(p, flip, dg, layout)
                    assert img.shape == ref.shape and img.dtype == ref.dtype
                    assert torch.equal(img.view(torch.uint8).flatten(), ref.view(torch.uint8).flatten()), (rnd, tuple(p.shape), layout, flip, dg)
        first = ops.weight_image(params[0], False, False, ops.PREP_X6)
        assert ops.weight_image(params[0], False, False, ops.PREP_X6) is first        # served from the cache
        with torch.no_grad():                                                         # "optimizer step"
            for p in params:
                p.add_(0.01 * torch.randn(p.shape, generator=g).to(dev()))
    ops.clear_weight_cache()


def test_tile_list_conv_leaves_its_persistent_counters_zero():
    """osn_spconv_fwd_tl_pc: caller-owned tile counters, zero before the first call and put back to zero by the last
    workgroup of every launch (no memset per call): many launches of different shapes in a row stay correct and
    bitwise repeatable, and the counters read zero afterwards."""
    from openscene_amd import ops
    cm = cloud("big")
    nbr = torch.from_numpy(cm.kmap(1, 1, 3)).to(dev())
    n = nbr.shape[1]
    tl = ops.tile_lists(nbr)
    g = torch.Generator().manual_seed(2)
    outs = []
    for rnd in range(3):
        for cin, cout in ((32, 32), (96, 128), (64, 20)):
            x = torch.randn(n, cin, generator=torch.Generator().manual_seed(cin)).to(dev())
            w = (torch.randn(27, cin, cout, generator=torch.Generator().manual_seed(cout)) * 0.1).to(dev())
            wf, _ = ops.weight_prep_tl(w, want_dgrad=False)
            out = ops.spconv_fwd_tl(x, wf, tl, n, 27, cout)
            if rnd == 0:
                ref = ops.spconv_fwd_x6(x, ops.weight_prep_x6(w), nbr, n)
                close(out, ref, "tile-list conv %d->%d" % (cin, cout), tol=2e-6)
                outs.append(out)
            else:
                assert torch.equal(out, outs[[(32, 32), (96, 128), (64, 20)].index((cin, cout))])
    torch.cuda.synchronize()
    assert ops._tl_counters and all(int(c.abs().sum()) == 0 for c in ops._tl_counters.values())
