"""HIP coordinate / kernel maps vs the oracle -- bit-exact (integer work)."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from openscene_amd import synthetic as syn

import cpu_backend

pytestmark = pytest.mark.gpu

S100K_LEVELS = [100999, 47618, 12868, 3052, 700]
S100K_PAIRS27 = [535643, 439572, 134680, 34704, 8630]
S100K_PAIRS125 = 1407639


def dev():
    return torch.device("cuda", 0)


def random_cloud(seed, n, extent, batch=1, lo=0):
    rng = np.random.default_rng(seed)
    rows = []
    for b in range(batch):
        g = np.unique(rng.integers(lo, lo + extent, (n, 3)), axis=0)
        g = g[rng.permutation(g.shape[0])]
        rows.append(np.concatenate([np.full((g.shape[0], 1), b), g], 1))
    return np.concatenate(rows, 0).astype(np.int32)


def s100k():
    v = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    return syn.batch_coords([v])


def test_unique_rows_keep_caller_order():
    from openscene_amd import ops
    c = random_cloud(0, 5000, 40, batch=2, lo=-7)
    out, inv, first, table = ops.coords_unique(torch.from_numpy(c).to(dev()), 1)
    assert out.shape[0] == c.shape[0]
    assert np.array_equal(out.cpu().numpy(), c)
    assert np.array_equal(inv.cpu().numpy(), np.arange(c.shape[0]))
    assert np.array_equal(first.cpu().numpy(), np.arange(c.shape[0]))


def test_duplicates_first_occurrence_order():
    from openscene_amd import ops
    rng = np.random.default_rng(1)
    c = random_cloud(1, 3000, 12, batch=2)
    c = c[rng.integers(0, c.shape[0], 9000)]                      # heavy duplication, random order
    t = torch.from_numpy(c)
    want = cpu_backend.coords_unique(t, 1)
    got = ops.coords_unique(t.to(dev()), 1)
    for w, g in zip(want[:3], got[:3]):
        assert np.array_equal(w.numpy(), g.cpu().numpy())


@pytest.mark.parametrize("case", ["random_neg", "room"])
def test_stride_pyramid_and_kernel_maps(case):
    from openscene_amd.sparse import CoordinateManager
    if case == "random_neg":
        c = random_cloud(2, 6000, 48, batch=2, lo=-20)
    else:
        c = syn.batch_coords([syn.shuffled(syn.grid_voxels(syn.room_points(1, n_pts=30000), 0.05), 1),
                              syn.shuffled(syn.grid_voxels(syn.room_points(2, n_pts=20000), 0.05), 2)])
    ref = oc.CoordinateManager(c)
    cm = CoordinateManager(torch.from_numpy(c).to(dev()))
    for s in (2, 4, 8, 16):
        assert np.array_equal(cm.coords(s).cpu().numpy(), ref.level(s)), "coords stride %d" % s
        assert np.array_equal(cm.parent(s).cpu().numpy(), ref.parent[s]), "parent stride %d" % s
    for (si, so_, k) in [(1, 1, 3), (1, 1, 5), (2, 2, 3), (4, 4, 3), (8, 8, 3), (16, 16, 3),
                         (1, 2, 2), (2, 4, 2), (4, 8, 2), (8, 16, 2), (2, 1, 2), (16, 8, 2)]:
        fwd, bwd, flip = cm.kmap(si, so_, k)
        want = ref.kmap(si, so_, k)
        assert np.array_equal(fwd.cpu().numpy(), want), "kmap %s" % ((si, so_, k),)
        # the table used for the input gradient is the transposed operator's map
        want_t = oc.transpose_table(want, ref.level(si).shape[0])
        got_t = bwd.cpu().numpy()[::-1] if flip else bwd.cpu().numpy()
        assert np.array_equal(got_t, want_t), "transposed kmap %s" % ((si, so_, k),)
    assert cm.kmap(1, 1, 1) == (None, None, False)


def test_s100k_full_size_properties():
    """Full BASELINE size: level sizes / pair counts of the canonical scene + structural properties."""
    from openscene_amd import ops
    from openscene_amd.sparse import CoordinateManager
    c = s100k()
    cm = CoordinateManager(torch.from_numpy(c).to(dev()))
    sizes = [cm.size(s) for s in (1, 2, 4, 8, 16)]
    assert sizes == S100K_LEVELS
    for lvl, s in enumerate((1, 2, 4, 8, 16)):
        nbr, _, _ = cm.kmap(s, s, 3)
        cnt = ops.kmap_count(nbr).cpu().numpy()
        assert np.array_equal(cm.kmap_counts(s, s, 3).cpu().numpy(), cnt)   # fused count of kmap_build
        assert int(cnt.sum()) == S100K_PAIRS27[lvl]
        assert int(cnt[13]) == sizes[lvl]                              # centre offset: every voxel sees itself
        assert np.array_equal(cnt, cnt[::-1])                          # (i,o) in map_k <=> (o,i) in map_{K-1-k}
        t = ops.kmap_transpose(nbr, sizes[lvl])
        assert torch.equal(t, torch.flip(nbr, dims=[0]))
        assert torch.equal(nbr[13], torch.arange(sizes[lvl], dtype=torch.int32, device=nbr.device))
    nbr5, _, _ = cm.kmap(1, 1, 5)
    assert int(ops.kmap_count(nbr5).sum()) == S100K_PAIRS125
    for s in (1, 2, 4, 8):                                             # k2s2: exactly one parent per fine voxel
        down, up, _ = cm.kmap(s, 2 * s, 2)
        assert int(ops.kmap_count(down).sum()) == cm.size(s)
        assert torch.equal(cm.kmap_counts(s, 2 * s, 2), ops.kmap_count(down))
        assert torch.equal(cm.kmap_counts(2 * s, s, 2), ops.kmap_count(up))
        assert torch.all((up >= 0).sum(0) == 1)
        par = cm.parent(2 * s).long()
        assert torch.equal(up.max(0)[0].long(), par)
    # bit-exact against the oracle at full size for the two biggest tables
    ref = oc.CoordinateManager(c)
    assert np.array_equal(cm.kmap(1, 1, 3)[0].cpu().numpy(), ref.kmap(1, 1, 3))
    assert np.array_equal(cm.coords(2).cpu().numpy(), ref.level(2))


def test_determinism_and_empty_and_range():
    from openscene_amd import ops, _lib
    c = torch.from_numpy(random_cloud(3, 20000, 64, batch=3)).to(dev())
    a = ops.coords_unique(c, 4)
    b = ops.coords_unique(c, 4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    e = ops.coords_unique(torch.zeros((0, 4), dtype=torch.int32, device=dev()), 1)
    assert e[0].shape == (0, 4) and e[1].shape == (0,)
    bad = torch.tensor([[0, 1, 2, 3], [0, 40000, 0, 0]], dtype=torch.int32, device=dev())
    with pytest.raises(_lib.OpenSceneAmdError):
        ops.coords_unique(bad, 1)


def test_kmap_sort_bit_exact_and_skip_efficiency():
    """osn_kmap_sort == stable argsort of the occupancy mask; on S100k it brings the active offsets
    per 128-row tile from 27 (hash order) to <= 16."""
    from openscene_amd import ops
    from openscene_amd.sparse import CoordinateManager
    c = s100k()
    cm = CoordinateManager(torch.from_numpy(c).to(dev()))
    for (si, so_, k) in [(1, 1, 3), (2, 2, 3), (1, 2, 2), (2, 1, 2)]:
        nbr = cm.kmap(si, so_, k)[0]
        for cnt in (None, cm.kmap_counts(si, so_, k)):
            order, tbl, gm = ops.kmap_sort(nbr, cnt)
            want_order, want_tbl, want_gm = cpu_backend.kmap_sort(nbr.cpu(), None if cnt is None else cnt.cpu())
            assert torch.equal(order.cpu(), want_order) and torch.equal(tbl.cpu(), want_tbl) and torch.equal(gm.cpu(), want_gm)
    nbr = cm.kmap(1, 1, 3)[0]
    order, tbl, _ = ops.kmap_sort(nbr)
    _, tbl_r, _ = ops.kmap_sort(nbr, cm.kmap_counts(1, 1, 3))

    def active(t):
        v = (t >= 0).cpu().numpy()
        n = v.shape[1] // 128 * 128
        return v[:, :n].reshape(27, -1, 128).any(2).sum(0).mean()
    assert active(nbr) > 26.5 and active(tbl) <= 16.0
    assert active(tbl_r) <= 0.93 * active(tbl)            # rarity-ordered key bits: another -10 %
    up = cm.kmap(2, 1, 2)[0]                      # transposed k2s2 table: one offset per row -> one per tile
    _, tup, _ = ops.kmap_sort(up)
    v = (tup >= 0).cpu().numpy()
    n = v.shape[1] // 128 * 128
    assert v[:, :n].reshape(8, -1, 128).any(2).sum(0).mean() < 1.2


@pytest.mark.parametrize("ksize", [1, 3, 5])
def test_half_probe_self_map_equals_the_full_probe(ksize):
    """osn_kmap_build_self (offsets below the centre probed, hits mirrored) against osn_kmap_build on the same table:
    tables and per-offset counts bit-identical, at a scaled offset too (a deeper level's stride)."""
    from openscene_amd import ops
    for seed, scale in ((3, 1), (4, 2)):
        c = random_cloud(seed, 9000, 40, batch=2, lo=-15)
        c[:, 1:] *= scale
        uniq, _, _, table = ops.coords_unique(torch.from_numpy(c).to(dev()))
        a, ca = ops.kmap_build(table, uniq, ksize, scale, with_counts=True, self_map=False)
        b, cb = ops.kmap_build(table, uniq, ksize, scale, with_counts=True, self_map=True)
        assert torch.equal(a, b) and torch.equal(ca, cb)
        assert torch.equal(ops.kmap_build(table, uniq, ksize, scale, self_map=True), a)      # without counts


def test_one_sync_pyramid_equals_the_level_by_level_build():
    """ops.coords_pyramid (all levels queued with device-side counts, one read-back) against coords_unique level by
    level: coordinates, parent maps, first rows bit-identical; the (larger) hash tables answer every probe the same way;
    duplicates and negative coordinates included; an unpackable coordinate is reported after the one read-back."""
    from openscene_amd import ops
    from openscene_amd._lib import OpenSceneAmdError
    c = random_cloud(9, 12000, 60, batch=3, lo=-25)
    c = np.concatenate([c, c[:500]])                       # duplicates at stride 1 too
    x = torch.from_numpy(c).to(dev())
    strides = (1, 2, 4, 8, 16)
    pyr = ops.coords_pyramid(x, strides)
    cur = x
    for s, (pc, pi, pf, pt) in zip(strides, pyr):
        rc, ri, rf, rt = ops.coords_unique(cur, s)
        assert torch.equal(pc, rc) and torch.equal(pi, ri) and torch.equal(pf, rf), "stride %d" % s
        scale = s
        assert torch.equal(ops.kmap_build(pt, pc, 3, scale), ops.kmap_build(rt, rc, 3, scale))
        cur = rc
    bad = c.copy()
    bad[7, 1] = 40000
    with pytest.raises(OpenSceneAmdError):
        ops.coords_pyramid(torch.from_numpy(bad).to(dev()), strides)
    one = ops.coords_pyramid(x[:1], strides)                # a single voxel: every level has one row
    assert [t[0].shape[0] for t in one] == [1] * 5


@pytest.mark.parametrize("streams", [1, 3])
def test_all_maps_from_one_call_equal_the_per_map_builds(streams, monkeypatch):
    """CoordinateManager.prebuild on the device = ONE osn_maps_build call that deals the levels' maps to several streams:
    every product (neighbour tables, transposed tables, pair counts, tile orders + group masks, tile lists, pair arrays)
    bit-identical to the per-map entry points, on one stream and on three; twice in a row (fork / join reuse)."""
    from openscene_amd import ops
    from openscene_amd.sparse import CoordinateManager
    monkeypatch.setattr(ops, "MAPS_STREAMS", streams)
    v = syn.shuffled(syn.grid_voxels(syn.room_points(12, n_pts=60000), 0.025), 12)
    coords = torch.from_numpy(syn.batch_coords([v, v[: len(v) // 2] + 3])).to(dev())
    ref = CoordinateManager(coords)
    monkeypatch.setattr(CoordinateManager, "_prebuild_fast", lambda self, *a, **k: None)
    ref.prebuild()
    monkeypatch.undo()
    monkeypatch.setattr(ops, "MAPS_STREAMS", streams)
    for _ in range(2):
        calls = []
        real = ops.maps_build
        monkeypatch.setattr(ops, "maps_build", lambda *a, **k: (calls.append(len(a[1])), real(*a, **k))[1])
        cm = CoordinateManager(coords)
        cm.prebuild(pairs=True)
        monkeypatch.setattr(ops, "maps_build", real)
        assert calls == [10]                                    # 5^3 + five 3^3 + four 2^3 maps, one call
        assert ref.kmap_tiles(1, 1, 3)[0] is not None and ref.kmap_tiles(16, 16, 3)[0] is None      # ordered and plain tables
        keys = [(1, 1, 5)] + [(s, s, 3) for s in (1, 2, 4, 8, 16)] + [(s, 2 * s, 2) for s in (1, 2, 4, 8)] + \
               [(2 * s, s, 2) for s in (1, 2, 4, 8)]
        for key in keys:
            a, b = cm.kmap(*key), ref.kmap(*key)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2], key
            assert torch.equal(cm.kmap_counts(*key), ref.kmap_counts(*key)), key
            for ta, tb in zip(cm.kmap_tiles(*key), ref.kmap_tiles(*key)):
                assert (ta is None) == (tb is None), key
                if ta is not None:
                    assert all(torch.equal(x, y) for x, y in zip(ta, tb)), key
            if key[2] == 5:
                continue
            for la, lb in zip(cm.kmap_lists(*key), ref.kmap_lists(*key)):
                assert la.bm == lb.bm and la.n_out == lb.n_out and (la.out_rows is None) == (lb.out_rows is None), key
                ca, cb = la.counts(), lb.counts()
                assert torch.equal(ca, cb), key
                # list entries are defined up to the count of their (tile, offset)
                valid = (torch.arange(la.bm, device=ca.device)[None, None, :] < ca[:, :, None])[..., None].expand(-1, -1, -1, 2)
                assert torch.equal(la.lists()[valid], lb.lists()[valid]), key
            if key[0] <= key[1]:                                # pair arrays of the forward lists (weight gradient)
                assert cm.kmap_lists(*key)[0].pairs is not None
                for x, y in zip(ops.pair_arrays(cm.kmap_lists(*key)[0]), ops.pair_arrays(ref.kmap_lists(*key)[0])):
                    assert torch.equal(x, y), key
