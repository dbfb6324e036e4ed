"""SparseTensor and its per-tensor coordinate manager (GPU resident).

Mirrors the slice of MinkowskiEngine's API the reference touches
(``from MinkowskiEngine import SparseTensor`` -- run/distill.py:18,316-317;
run/evaluate.py:18,284; models/mink_unet.py:116-174 use ``.F``, ``ME.cat``
and ``+=``).  Semantics restated in SURVEY.md appendix C items 1-4:

  * a new CoordinateManager per SparseTensor (maps are rebuilt every forward);
  * unique coordinates keep the caller's row order;
  * stride-2 convs create floor(c / 2s) * 2s coarse maps (cached);
  * transposed convs land on the cached finer map, so decoder outputs are
    row-aligned with encoder skips.
"""
import torch

from . import ops


def _F_mod():
    from . import functional       # late: functional imports ops, which this module imports too
    return functional


class CoordinateManager:
    """Caches, per tensor stride, the coordinate rows + hash table, and per
    (in stride, out stride, kernel size, dilation) the k-major neighbour tables."""

    PYRAMID = (2, 4, 8, 16)       # coarse levels built together with the stride-1 map (one host sync for all of them)

    def __init__(self, coordinates, pyramid=None):
        pyramid = self.PYRAMID if pyramid is None else tuple(pyramid)
        levels = ops.coords_pyramid(coordinates, (1,) + pyramid)
        coords, inverse, first, table = levels[0]
        self.n_input = coordinates.shape[0]
        self.unique_index = None          # rows kept if the caller passed duplicates
        self.inverse_mapping = None
        if coords.shape[0] != coordinates.shape[0]:
            self.unique_index = first.long()
            self.inverse_mapping = inverse.long()
        else:
            coords = coordinates.contiguous()     # identical rows, caller's own storage order
        self._coords = {1: coords}
        self._tables = {1: table}
        self._parent = {}
        for s, (c, parent, _first, t) in zip(pyramid, levels[1:]):
            self._coords[s], self._tables[s], self._parent[s] = c, t, parent
        self._kmaps = {}
        self.device = coordinates.device

    # -- coordinate maps ----------------------------------------------------
    def coords(self, stride):
        if stride not in self._coords:
            if stride < 2 or stride % 2:
                raise ValueError("tensor stride %r was never created on this manager" % (stride,))
            fine = self.coords(stride // 2)
            c, parent, _first, table = ops.coords_unique(fine, stride)
            self._coords[stride] = c
            self._tables[stride] = table
            self._parent[stride] = parent
        return self._coords[stride]

    def size(self, stride):
        return self.coords(stride).shape[0]

    def parent(self, stride):
        """int32 [N_{stride/2}] : row of the stride-`stride` voxel containing each finer voxel."""
        self.coords(stride)
        return self._parent[stride]

    def has(self, stride):
        return stride in self._coords

    # -- kernel maps ---------------------------------------------------------
    def kmap(self, in_stride, out_stride, ksize, dilation=1):
        """(nbr_fwd, nbr_bwd, flip).  nbr_fwd [K, N_out] feeds the forward conv;
        the input gradient is a forward conv of grad_out over nbr_bwd with the
        kernel transposed (and mirrored in k if flip)."""
        key = (in_stride, out_stride, ksize, dilation)
        hit = self._kmaps.get(key)
        if hit is not None:
            return hit
        if ksize == 1 and in_stride == out_stride:
            res = (None, None, False)
        elif in_stride == out_stride:
            self.coords(in_stride)
            nbr, cnt = ops.kmap_build(self._tables[in_stride], self._coords[in_stride], ksize, dilation * in_stride,
                                      with_counts=True, self_map=True)
            self._kmaps[("counts",) + key] = cnt
            if ksize % 2 == 1:
                res = (nbr, nbr, True)            # the map of an odd stride-1 kernel is its own mirror
            else:
                res = (nbr, ops.kmap_transpose(nbr, self.size(in_stride)), False)
        elif out_stride > in_stride:              # strided conv: fine -> coarse
            out_c = self.coords(out_stride)
            nbr, cnt = ops.kmap_build(self._tables[in_stride], out_c, ksize, dilation * in_stride, with_counts=True)
            self._kmaps[("counts",) + key] = cnt
            res = (nbr, ops.kmap_transpose(nbr, self.size(in_stride)), False)
        else:                                     # transposed conv: swap the fine -> coarse map
            down_fwd, down_bwd, _ = self.kmap(out_stride, in_stride, ksize, dilation)
            # the swapped table holds the same pairs per offset
            self._kmaps[("counts",) + key] = self._kmaps.get(("counts", out_stride, in_stride, ksize, dilation))
            res = (down_bwd, down_fwd, False)
        self._kmaps[key] = res
        return res

    def kmap_counts(self, in_stride, out_stride, ksize, dilation=1):
        """int64 [K] pair count per offset of the forward table (device), or None for K == 1;
        lets the weight-gradient kernel balance its work items."""
        key = ("counts", in_stride, out_stride, ksize, dilation)
        if key not in self._kmaps:
            fwd = self.kmap(in_stride, out_stride, ksize, dilation)[0]       # kmap_build fills the count as it goes
            if key not in self._kmaps:
                self._kmaps[key] = ops.kmap_count(fwd) if fwd is not None else None
        return self._kmaps[key]

    def prebuild(self, strides=(2, 4, 8, 16), kernel_sizes=(3,), stem_kernel=5, pairs=False, streams=None):
        """Build every kernel map (and any level of the pyramid the constructor did not create) up front, before any
        heavy kernel is queued: the host then runs ahead of the GPU for the rest of the forward/backward pass instead of
        stopping at every first use of a map.  (The pyramid's levels 1 ... 16 were built by the constructor with one
        host synchronisation; a level beyond them costs one read-back of its unique count here.)
        On the device all maps come from ONE C call that deals the levels' independent chains of launches to several
        streams (ops.maps_build); pairs: True = also build the pair arrays of every map (training: the weight gradients),
        "ws" = only those a weight-stationary convolution reads in inference (2^3 stride-2 maps, maps of the deep levels)."""
        for s in strides:
            self.coords(s)
        levels = (1,) + tuple(strides)
        if self.device.type == "cuda" and all(self.size(s) > 0 for s in levels):
            self._prebuild_fast(levels, kernel_sizes, stem_kernel, pairs, streams)
        if stem_kernel:
            self.kmap(1, 1, stem_kernel)
            self.kmap_tiles(1, 1, stem_kernel)
        for s in levels:
            for k in kernel_sizes:
                self.kmap(s, s, k)
                self.kmap_tiles(s, s, k)
                self.kmap_lists(s, s, k)
        for s in levels[:-1]:
            self.kmap(s, 2 * s, 2)
            self.kmap_tiles(s, 2 * s, 2)
            self.kmap_lists(s, 2 * s, 2)
            self.kmap(2 * s, s, 2)
            self.kmap_tiles(2 * s, s, 2)
            self.kmap_lists(2 * s, s, 2)

    def _want_sort(self, K, rows):
        """kmap_tiles' rule: which tables get a tile-ordered copy."""
        return K <= 32 and rows >= self.SORT_MIN_ROWS and not (K <= 8 and rows < self.SORT_MIN_ROWS_K8)

    def _prebuild_fast(self, levels, kernel_sizes, stem_kernel, pairs, streams=None):
        """The maps prebuild() asks for, not cached yet, as jobs of one ops.maps_build call; results land in the caches
        kmap() / kmap_counts() / kmap_tiles() / kmap_lists() read -- the same tensors the per-map path would create.
        The GPU is idle when this runs (the pyramid's size read-back has just returned), so the launches go out first:
        ONE allocation, addresses by arithmetic, the C call -- and only then the tensor views over the allocation."""
        dev = self.device
        index = {s: i for i, s in enumerate(levels)}
        specs = ([(1, 1, stem_kernel)] if stem_kernel else []) + [(s, s, k) for s in levels for k in kernel_sizes] + \
                [(s, 2 * s, 2) for s in levels[:-1]]
        # streams: the two longest chains (the 5^3 map; the level-0 3^3 map with its tile order and lists) get one each
        lane = {(1, 1, stem_kernel): 1, (1, 1, 3): 2, (4, 4, 3): 2, (1, 2, 2): 2}
        plan, total, sort_rows = [], 0, 0

        def take(nbytes):
            nonlocal total
            off = total
            total += (int(nbytes) + 255) // 256 * 256
            return off, int(nbytes)
        for si, so, k in specs:
            key = (si, so, k, 1)
            if key in self._kmaps or (si == so and k % 2 == 0):
                continue
            K, n_in, n_out = k ** 3, self.size(si), self.size(so)
            own = si == so
            e = dict(key=key, K=K, n_in=n_in, n_out=n_out, own=own, lane=lane.get((si, so, k), 0), lvl_in=index[si], lvl_out=index[so],
                     ksize=k, scale=si)
            e["nbr_fwd"] = take(4 * K * n_out)
            e["counts"] = take(8 * K)
            if not own:
                e["nbr_bwd"] = take(4 * K * n_in)
            for side, rows, on in (("fwd", n_out, True), ("bwd", n_in, not own)):
                if on and self._want_sort(K, rows):
                    e["order_" + side], e["sorted_" + side] = take(4 * rows), take(4 * K * rows)
                    e["gmask_" + side] = take(4 * ((rows + 31) // 32))
                    sort_rows = max(sort_rows, rows)
                if on and K <= 32:                       # tile lists: every 3^3 / 2^3 map (not the 3-channel stem's 5^3 map)
                    bm = ops.tile_rows(rows)
                    e["bm_" + side] = bm
                    e["tl_" + side] = take(ops._cached("osn_tile_lists_bytes", rows, K, bm))
            if "tl_fwd" in e and (pairs is True or (pairs == "ws" and ((k == 2 and so == 2 * si) or n_out <= _F_mod().WS_MAX_ROWS))):
                e["pl_fwd"] = take(ops._cached("osn_pair_lists_bytes", n_out, K, e["bm_fwd"]))
            plan.append(e)
        if not plan:
            return
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        base = arena.data_ptr()
        ptr_fields = ("nbr_fwd", "nbr_bwd", "counts", "order_fwd", "sorted_fwd", "gmask_fwd", "order_bwd", "sorted_bwd", "gmask_bwd",
                      "tl_fwd", "tl_bwd", "pl_fwd")
        jobs = []
        for e in plan:
            q = dict(lvl_in=e["lvl_in"], lvl_out=e["lvl_out"], ksize=e["ksize"], scale=e["scale"], self_map=int(e["own"]), stream=e["lane"],
                     bm_fwd=e.get("bm_fwd", 0), bm_bwd=e.get("bm_bwd", 0))
            for f in ptr_fields:
                if f in e:
                    q[f] = base + e[f][0]
            jobs.append(q)
        ops.maps_build([(self._coords[s], self._tables[s], self.size(s)) for s in levels], jobs, dev, sort_rows, streams)

        # ---- the launches are queued; now the views the rest of the library works with
        def view(e, f, dtype, shape):
            off, nb = e[f]
            return arena[off:off + nb].view(dtype).view(shape)
        for e in plan:
            K, n_in, n_out, key = e["K"], e["n_in"], e["n_out"], e["key"]
            nbr = view(e, "nbr_fwd", torch.int32, (K, n_out))
            counts = view(e, "counts", torch.int64, (K,))
            bwd = nbr if e["own"] else view(e, "nbr_bwd", torch.int32, (K, n_in))
            tiles = {}
            for side, rows in (("fwd", n_out), ("bwd", n_in)):
                tiles[side] = (view(e, "order_" + side, torch.int32, (rows,)), view(e, "sorted_" + side, torch.int32, (K, rows)),
                               view(e, "gmask_" + side, torch.int32, ((rows + 31) // 32,))) if ("order_" + side) in e else None
            if e["own"]:
                tiles["bwd"] = tiles["fwd"]
            self._kmaps[key] = (nbr, bwd, bool(e["own"]))
            self._kmaps[("counts",) + key] = counts
            self._kmaps[("tiles",) + key] = (tiles["fwd"], tiles["bwd"])
            if "tl_fwd" in e:
                lf = ops.TileLists(view(e, "tl_fwd", torch.uint8, (e["tl_fwd"][1],)), e["bm_fwd"], n_out, K,
                                   tiles["fwd"][0] if tiles["fwd"] is not None else None)
                lb = lf if e["own"] else ops.TileLists(view(e, "tl_bwd", torch.uint8, (e["tl_bwd"][1],)), e["bm_bwd"], n_in, K,
                                                       tiles["bwd"][0] if tiles["bwd"] is not None else None)
                if "pl_fwd" in e:
                    lf.pairs = view(e, "pl_fwd", torch.uint8, (e["pl_fwd"][1],))
                self._kmaps[("lists",) + key] = (lf, lb)

    def tensors(self):
        """Every device tensor this manager owns (coordinates, hash tables, parent maps, kernel maps, tile
        orders, pair counts)."""
        seen = []

        def walk(v):
            if isinstance(v, torch.Tensor):
                seen.append(v)
            elif isinstance(v, (tuple, list)):
                for u in v:
                    walk(u)
            elif isinstance(v, ops.HashTable):
                walk((v.keys, v.vals))
            elif isinstance(v, ops.TileLists):
                walk((v.buf, v.out_rows))
        for d in (self._coords, self._tables, self._parent, self._kmaps):
            for v in d.values():
                walk(v)
        return seen

    def record_stream(self, stream):
        """Tell the caching allocator that `stream` uses this manager's tensors (built on another stream)."""
        for t in self.tensors():
            t.record_stream(stream)

    SORT_MIN_ROWS = 8192   # below this the sort does not pay
    # 2^3 (strided / transposed) maps are never tile-ordered: measured per-step kernel time 14.81 ms with, 14.42 ms without
    # (tools/ab_kernel_time.sh): their eight sorts cost more than the convs of those maps gain
    SORT_MIN_ROWS_K8 = 10 ** 9

    def kmap_tiles(self, in_stride, out_stride, ksize, dilation=1):
        """Tile-ordered variants of kmap(): ((order, table) for the forward conv or None,
        (order, table) for the input gradient or None).  Rows are ordered by their
        offset-occupancy mask (ops.kmap_sort) so the conv kernel's per-tile offset skip bites;
        features keep the caller's row order (the permutation rides along as `out_rows`)."""
        key = ("tiles", in_stride, out_stride, ksize, dilation)
        hit = self._kmaps.get(key)
        if hit is not None:
            return hit
        fwd, bwd, flip = self.kmap(in_stride, out_stride, ksize, dilation)
        counts = self.kmap_counts(in_stride, out_stride, ksize, dilation)     # same per offset for fwd and bwd tables

        def tiles(tbl):
            if tbl is None or tbl.shape[0] > 32 or tbl.shape[1] < self.SORT_MIN_ROWS:
                return None
            if tbl.shape[0] <= 8 and tbl.shape[1] < self.SORT_MIN_ROWS_K8:
                return None
            return ops.kmap_sort(tbl, counts)

        tf = tiles(fwd)
        tb = tf if (bwd is fwd) else tiles(bwd)
        res = (tf, tb)
        self._kmaps[key] = res
        return res



    def kmap_lists(self, in_stride, out_stride, ksize, dilation=1):
        """Per-tile compacted pair lists (ops.TileLists) of the forward table and of the input-gradient
        table of a map: (tl_fwd or None, tl_bwd or None).  Built from the tile-ordered tables where those
        exist, so a tile's rows share their offsets; None for K == 1.  (The weight gradient uses them on every map;
        the forward / input gradient only on large maps -- functional.TL_FWD_MIN_ROWS.)"""
        key = ("lists", in_stride, out_stride, ksize, dilation)
        hit = self._kmaps.get(key)
        if hit is not None:
            return hit
        if out_stride < in_stride:                # transposed conv: the strided conv's lists, roles swapped
            lf, lb = self.kmap_lists(out_stride, in_stride, ksize, dilation)
            res = self._kmaps[key] = (lb, lf)
            return res
        fwd, bwd, flip = self.kmap(in_stride, out_stride, ksize, dilation)
        tf, tb = self.kmap_tiles(in_stride, out_stride, ksize, dilation)

        def lists(tbl, tiles):
            if tbl is None or tbl.shape[0] > 128 or tbl.shape[1] == 0:
                return None
            if tiles is not None:
                return ops.tile_lists(tiles[1], out_rows=tiles[0])
            return ops.tile_lists(tbl)

        lf = lists(fwd, tf)
        lb = lf if (bwd is fwd) else lists(bwd, tb)
        res = (lf, lb)
        self._kmaps[key] = res
        return res


class MapPrefetcher:
    """Builds the coordinate pyramid and every kernel map of the NEXT batch on a high-priority side
    stream while the current step's kernels run on the main stream.  The maps depend on coordinates
    only, which a loader knows one batch ahead; their ~250 launches are latency-bound and leave the
    chip mostly idle when they run alone (1.7 ms of an 18 ms step on S100k).

        pf = MapPrefetcher(device)
        handle = pf.submit(coords_next)          # after enqueueing the current step
        ...
        x = SparseTensor(feats, coordinate_manager=pf.take(handle))

    submit() blocks the host only on the side stream (the size read-back of the pyramid).

    Measured on MI355X / S100k (one scene per step).  Round 2 (per-module host path, ~90 C calls per map set): 18.3 ms per
    step with the prefetcher against 17.6 ms without -- the step was host-bound and the prefetcher's read-backs stalled it.
    Round 3 (network executor, every map from one C call on one stream): 10.57 ms with pyramid + maps ahead, 11.10 ms with
    the pyramid only (`bench.py --no-prefetch-maps`), 11.9 ms with nothing ahead -- the maps' ~130 latency-bound launches
    run beside the previous step's backward pass instead of in front of the forward pass."""

    def __init__(self, device, threaded=False, pyramid_only=False, **prebuild_args):
        """threaded: build on a worker thread, so that the caller never blocks on the pyramid's size read-backs (the
        C calls release the GIL while they wait): submit() returns at once, take() joins the build.
        pyramid_only: queue only the coordinate pyramid ahead (five short launch chains and the one size read-back -- the
        part of a step during which nothing else can run); the kernel maps are then built inside the step, from one C call
        (ops.maps_build), when the forward pass asks for them."""
        self.device = torch.device(device)
        self.stream = ops.aux_stream(self.device)     # the process's third stream (one per device, high priority; shared by every prefetcher)
        self.prebuild_args = prebuild_args
        self.pyramid_only = bool(pyramid_only)
        self.threaded = bool(threaded)
        self._pool = None
        if self.threaded:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="osn-maps")

    def _build(self, coordinates, ready):
        torch.cuda.set_device(self.device)
        self.stream.wait_event(ready)
        coordinates.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            cm = CoordinateManager(coordinates)
            if not self.pyramid_only:
                cm.prebuild(**self.prebuild_args)
            done = torch.cuda.Event()
            done.record(self.stream)
        return cm, done

    def submit(self, coordinates):
        """`coordinates` int32 [N,4] must be complete on the CURRENT stream when this is called."""
        if coordinates.dtype != torch.int32:
            coordinates = coordinates.int()
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        if self._pool is not None:
            return self._pool.submit(self._build, coordinates, ready)
        return self._build(coordinates, ready)

    def take(self, handle):
        cm, done = handle.result() if hasattr(handle, "result") else handle
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        cm.record_stream(main)
        return cm


class SparseTensor:
    """features float32 [N, C] + int32 coordinates [N, 4] (batch, x, y, z) on one device."""

    def __init__(self, features, coordinates=None, tensor_stride=1, coordinate_manager=None, **unused):
        if coordinate_manager is None:
            if coordinates is None:
                raise ValueError("SparseTensor needs coordinates or a coordinate_manager")
            if coordinates.dtype != torch.int32:
                coordinates = coordinates.int()
            if coordinates.device != features.device:
                raise ValueError("features on %s but coordinates on %s" % (features.device, coordinates.device))
            if coordinates.shape[0] != features.shape[0]:
                raise ValueError("%d feature rows vs %d coordinate rows" % (features.shape[0], coordinates.shape[0]))
            coordinate_manager = CoordinateManager(coordinates)
            if coordinate_manager.unique_index is not None:
                features = features[coordinate_manager.unique_index]
        elif (tensor_stride == 1 and coordinate_manager.unique_index is not None
              and features.shape[0] == coordinate_manager.n_input):
            # a prefetched manager built from the raw (duplicated) coordinates of these feature rows
            features = features[coordinate_manager.unique_index]
        self._F = features
        self.tensor_stride = tensor_stride
        self.coordinate_manager = coordinate_manager

    # ME names
    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return self.coordinate_manager.coords(self.tensor_stride)

    features = F
    coordinates = C

    @property
    def device(self):
        return self._F.device

    @property
    def shape(self):
        return self._F.shape

    @property
    def dtype(self):
        return self._F.dtype

    def size(self, *a):
        return self._F.size(*a)

    def __len__(self):
        return self._F.shape[0]

    def _like(self, feats):
        return SparseTensor(feats, tensor_stride=self.tensor_stride, coordinate_manager=self.coordinate_manager)

    def _same_map(self, other):
        if other.coordinate_manager is not self.coordinate_manager or other.tensor_stride != self.tensor_stride:
            raise ValueError("SparseTensors live on different coordinate maps")

    def __add__(self, other):
        if isinstance(other, SparseTensor):
            self._same_map(other)
            return self._like(_F_mod().add(self._F, other._F))
        return self._like(self._F + other)

    def __iadd__(self, other):
        # the reference's residual `out += residual`; out-of-place on the feature
        # matrix so autograd never sees an in-place edit of a saved tensor
        if isinstance(other, SparseTensor):
            self._same_map(other)
            self._F = _F_mod().add(self._F, other._F)
        else:
            self._F = self._F + other
        return self

    def __repr__(self):
        return "SparseTensor(N=%d, C=%d, tensor_stride=%d, device=%s)" % (
            self._F.shape[0], self._F.shape[1], self.tensor_stride, self._F.device)


def cat(*tensors):
    """ME.cat: column concat of tensors on the same coordinate map (models/mink_unet.py:147,155,163,171)."""
    if len(tensors) == 1 and isinstance(tensors[0], (list, tuple)):
        tensors = tuple(tensors[0])
    first = tensors[0]
    for t in tensors[1:]:
        first._same_map(t)
    return first._like(_F_mod().cat([t.F for t in tensors]))
