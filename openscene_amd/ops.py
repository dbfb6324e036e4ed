"""Tensor-level wrappers over the C ABI: allocate outputs / scratch with torch,
pass raw device pointers and the current HIP stream.  No autograd here (see
functional.py) and no CPU fallback."""
import ctypes
import os
import threading
import weakref

import torch

from . import _lib
from ._lib import check

_vp = ctypes.c_void_p

# Optional launch profiler (bench.py): an object with start(kind, dev, **meta) -> token and
# stop(token); brackets the C-ABI call with HIP events on the stream the kernels run on.
_profiler = None


def set_profiler(p):
    global _profiler
    _profiler = p


# Host-side cost matters: a training step is ~1200 launches and the deep U-Net levels run kernels of a few
# microseconds, so every microsecond spent here is a microsecond the GPU may sit idle
# (tools/host_profile.py measures this layer against a mock library).  Hence raw ints instead of
# ctypes / torch.cuda.Stream objects and one raw-stream query per helper.
_raw_stream_of = torch._C._cuda_getCurrentRawStream      # device index -> current hipStream_t as int
_current_device = torch._C._cuda_getDevice


def _prof_start(kind, dev, **meta):
    # the profiler brackets torch's current stream: launches redirected by on_stream() are not bracketed
    if _profiler is None or getattr(_tls, "stream", None) is not None:
        return None
    return _profiler.start(kind, dev, **meta)


def _p(t):
    return t.data_ptr() if t is not None else None       # ctypes converts the int to void* (argtypes are set)


def _idx(dev):
    i = dev.index
    return i if i is not None else _current_device()


_tls = threading.local()


def _stream(dev):
    s = getattr(_tls, "stream", None)
    return s if s is not None else _raw_stream_of(_idx(dev))


class on_stream:
    """Context: the C-ABI calls of this thread launch on `stream` (a torch.cuda.Stream) instead of torch's current
    stream -- without touching torch's stream state, so tensors are still allocated from the current stream's pool.
    The caller orders the two streams (functional.py forks / joins around the weight gradient)."""
    __slots__ = ("raw", "prev", "torch_stream", "prev_torch")

    def __init__(self, stream):
        self.raw = stream.cuda_stream
        self.torch_stream = stream

    def __enter__(self):
        self.prev = getattr(_tls, "stream", None)
        self.prev_torch = getattr(_tls, "torch_stream", None)
        _tls.stream = self.raw
        _tls.torch_stream = self.torch_stream

    def __exit__(self, *a):
        _tls.stream = self.prev
        _tls.torch_stream = self.prev_torch


_side_streams = {}


def side_stream(dev):
    """One auxiliary HIP stream per device (created on first use), default priority.  (The weight gradients queued here crowd
    the main stream's memory-bound kernels; a LOW-priority stream was measured twice -- round 3, and round 5 with the executor,
    profiles/r05_s1_knobs_ab.txt -- as no change: the hardware's priorities do not pre-empt resident workgroups.)"""
    i = _idx(dev)
    s = _side_streams.get(i)
    if s is None:
        s = torch.cuda.Stream(device=i)
        _side_streams[i] = s
    return s


_aux_streams = {}


def aux_stream(dev):
    """The THIRD stream of the process (per device, high priority, created on first use): the map prefetcher's stream, and the third
    lane of an inference pass's map build.  The library never creates a fourth: the process is held to three hardware queues
    (openscene_amd.configure_hw_queues), and a stream beyond them shares a queue with one that matters (round 6: two extra map
    streams moved a LATER prefetcher onto the main stream's queue and the scene-stream inference went from 2.85 to 4.85 ms per scene)."""
    i = _idx(dev)
    s = _aux_streams.get(i)
    if s is None:
        _lo, hi = torch.cuda.Stream.priority_range()
        s = _aux_streams[i] = torch.cuda.Stream(device=i, priority=hi)
    return s


# Scratch for the C-ABI calls: ONE growing buffer per (device, stream).  Every call's scratch is only
# live while that call's kernels run, and calls on one stream execute in order (autograd's backward
# thread launches on the same stream), so sharing is safe and saves a torch.empty per launch.
_ws_pool = {}


def _ws(nbytes, dev):
    nbytes = max(int(nbytes), 16)
    i = _idx(dev)
    key = (i, _stream(dev))
    buf = _ws_pool.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _ws_pool[key] = buf
    return buf


def ws_on(nbytes, dev, raw_stream):
    """The growing scratch buffer of (device, the given raw stream) -- for launches a C call places on another stream."""
    nbytes = max(int(nbytes), 16)
    key = (_idx(dev), raw_stream)
    buf = _ws_pool.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _ws_pool[key] = buf
    return buf


_size_cache = {}


def _cached(fn_name, *args):
    """ws-size / plan helpers are pure functions of their integer arguments: memoise the ctypes call."""
    key = (fn_name,) + args
    v = _size_cache.get(key)
    if v is None:
        v = getattr(_lib.load(), fn_name)(*args)
        _size_cache[key] = v
    return v


_dev_ok = {}


def _prep(dev):
    if dev not in _dev_ok:
        _lib.require_device(dev)          # raises for CPU tensors / other architectures
        _dev_ok[dev] = True
    return _lib.load()


class _Dev:
    """Make `dev` the current HIP device for the duration of a call if it is not already."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index
        self.ctx = None if (idx is None or idx == _current_device()) else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _query_feats(t, name):
    """Query features: float32 (network output) or float16 (fused 2-D features as the reference stores them,
    scripts/feature_fusion/fusion_util.py:87).  The kernel reads fp32 rows; an fp16 matrix is widened once here
    (half -> float -> half is the identity, so `feats.half() @ text` is unchanged; costs one pass over it)."""
    if t.dtype == torch.float16:
        t = t.float()
    return _f32c(t, name)


# ----------------------------------------------------------------- coordinates
class HashTable:
    """Open-addressing table keys[cap] u64 / vals[cap] i32 living in HBM."""
    __slots__ = ("keys", "vals", "cap")

    def __init__(self, n, dev):
        self.cap = int(_lib.load().osn_hash_capacity(int(n)))
        self.keys = torch.empty(self.cap, dtype=torch.int64, device=dev)
        self.vals = torch.empty(self.cap, dtype=torch.int32, device=dev)


def coords_unique(coords4, stride=1):
    """-> (unique coords [U,4] int32 in first-occurrence order, inverse [N] int32,
    first [U] int32, HashTable mapping packed key -> unique row)."""
    if coords4.dtype != torch.int32 or coords4.dim() != 2 or coords4.shape[1] != 4:
        raise TypeError("coordinates must be int32 [N, 4] rows (batch, x, y, z)")
    dev = coords4.device
    lib = _prep(dev)
    coords4 = coords4.contiguous()
    n = coords4.shape[0]
    table = HashTable(n, dev)
    out = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    inverse = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    first = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    wsb = _cached("osn_coords_unique_ws_bytes", n)
    ws = _ws(wsb, dev)
    nu = ctypes.c_int64(0)
    with _Dev(dev):
        check(lib.osn_coords_unique(_p(coords4), n, int(stride), _p(table.keys), _p(table.vals), table.cap,
                                    _p(out), _p(inverse), _p(first), ctypes.byref(nu), _p(ws), ws.numel(),
                                    _stream(dev)), "osn_coords_unique")
    u = int(nu.value)
    return out[:u], inverse[:n], first[:u], table


def coords_pyramid(coords4, strides=(1, 2, 4, 8, 16)):
    """coords_unique for a chain of strides (level i is the unique set of level i-1 quantised to strides[i]) with ONE
    host synchronisation for the whole chain instead of one per level: every level is queued with its row count still
    in device memory (osn_coords_unique_async) and sized for the input's row count; the counts come back together.
    -> [(coords [U_i, 4], inverse [U_{i-1}] (parent map), first [U_i], HashTable)] -- the tuples coords_unique returns."""
    if coords4.dtype != torch.int32 or coords4.dim() != 2 or coords4.shape[1] != 4:
        raise TypeError("coordinates must be int32 [N, 4] rows (batch, x, y, z)")
    n0 = coords4.shape[0]
    if n0 == 0 or len(strides) == 1:
        res, cur = [], coords4
        for s in strides:
            r = coords_unique(cur, s)
            res.append(r)
            cur = r[0]
        return res
    dev = coords4.device
    lib = _prep(dev)
    coords4 = coords4.contiguous()
    L = len(strides)
    counts = torch.empty(L + 1, dtype=torch.int32, device=dev)        # [0 .. L-1] unique rows per level, [L] range error (zeroed by the call)
    ws = _ws(_cached("osn_coords_unique_ws_bytes", n0), dev)
    st = _stream(dev)
    bufs = []
    for _ in strides:
        bufs.append((torch.empty((n0, 4), dtype=torch.int32, device=dev), torch.empty(n0, dtype=torch.int32, device=dev),
                     torch.empty(n0, dtype=torch.int32, device=dev), HashTable(n0, dev)))
    arr = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])
    with _Dev(dev):
        # one call, one preset launch for every table (round 6; was one osn_coords_unique_async per level)
        check(lib.osn_coords_pyramid_async(_p(coords4), n0, (ctypes.c_int32 * L)(*[int(s) for s in strides]), L,
                                           arr([b[3].keys for b in bufs]), arr([b[3].vals for b in bufs]), bufs[0][3].cap,
                                           arr([b[0] for b in bufs]), arr([b[1] for b in bufs]), arr([b[2] for b in bufs]),
                                           _p(counts), _p(ws), ws.numel(), st), "osn_coords_pyramid_async")
    host = counts.tolist()                                             # the one synchronisation
    if host[L]:
        raise _lib.OpenSceneAmdError("osn_coords_pyramid_async: coordinate outside the packable range "
                                     "(|x|,|y|,|z| < 32767, 0 <= batch < 65535)")
    res, u_prev = [], n0
    for (out, inverse, first, table), u in zip(bufs, host[:L]):
        res.append((out[:u], inverse[:u_prev], first[:u], table))
        u_prev = u
    return res


def kmap_build(table, out_coords4, ksize, offset_scale, with_counts=False, self_map=False):
    """nbr int32 [K, n_out] (and, with_counts, int64 [K] pairs per offset from the same pass).  self_map: out_coords4 are
    the rows the table was built from, in row order, and ksize is odd (stride-1 convolution): the half-probe builder."""
    dev = out_coords4.device
    lib = _prep(dev)
    out_coords4 = out_coords4.contiguous()
    n_out = out_coords4.shape[0]
    K = ksize ** 3
    nbr = torch.empty((K, n_out), dtype=torch.int32, device=dev)
    counts = torch.empty(K, dtype=torch.int64, device=dev) if with_counts else None
    fn, what = (lib.osn_kmap_build_self, "osn_kmap_build_self") if (self_map and ksize % 2 == 1) else \
        (lib.osn_kmap_build, "osn_kmap_build")
    with _Dev(dev):
        check(fn(_p(table.keys), _p(table.vals), table.cap, _p(out_coords4), n_out, int(ksize),
                 int(offset_scale), _p(nbr), _p(counts), _stream(dev)), what)
    return (nbr, counts) if with_counts else nbr


def kmap_transpose(nbr, n_in):
    dev = nbr.device
    lib = _prep(dev)
    K, n_out = nbr.shape
    tbl = torch.empty((K, int(n_in)), dtype=torch.int32, device=dev)
    with _Dev(dev):
        check(lib.osn_kmap_transpose(_p(nbr), n_out, K, int(n_in), _p(tbl), _stream(dev)), "osn_kmap_transpose")
    return tbl


def kmap_sort(nbr, counts=None):
    """-> (order int32 [n_out], nbr_sorted int32 [K, n_out], gmask int32 [ceil(n_out/32)]): rows ordered by
    offset-occupancy mask (key bits ordered by rarity when the per-offset pair `counts` are given);
    gmask = OR of the masks (bit k = offset k) of each 32-row group of the sorted table."""
    dev = nbr.device
    lib = _prep(dev)
    nbr = nbr.contiguous()
    K, n_out = nbr.shape
    order = torch.empty(n_out, dtype=torch.int32, device=dev)
    out = torch.empty_like(nbr)
    gmask = torch.empty((n_out + 31) // 32, dtype=torch.int32, device=dev)
    with _Dev(dev):
        wsb = _cached("osn_kmap_sort_ws_bytes", n_out)
        ws = _ws(wsb, dev)
        check(lib.osn_kmap_sort(_p(nbr), n_out, K, _p(counts), _p(order), _p(out), _p(gmask), _p(ws), ws.numel(),
                                _stream(dev)), "osn_kmap_sort")
    return order, out, gmask


def kmap_count(nbr):
    dev = nbr.device
    lib = _prep(dev)
    K, n_out = nbr.shape
    counts = torch.empty(K, dtype=torch.int64, device=dev)
    with _Dev(dev):
        check(lib.osn_kmap_count(_p(nbr), n_out, K, _p(counts), _stream(dev)), "osn_kmap_count")
    return counts


# ------------------------------------------------- every kernel map of a scene from one call, on several streams
_MAP_LEVEL = None
_MAP_JOB = None
_map_events = {}
# streams the jobs of one maps_build call are dealt to.  Measured on MI355X (profiles/r03_s5..s7): 2 - 4 streams do not
# shorten the step (12.5 - 12.7 ms either way: the 5^3 stem map is the long pole and the first thing the forward pass
# needs), and every extra stream counts against the runtime's four hardware queues -- with the executor's side stream
# and the input prefetcher's stream in flight as well, streams start to share queues and the step DOUBLES (26 ms).
MAPS_STREAMS = int(os.environ.get("OSN_MAPS_STREAMS", "1"))        # 1 = everything on the caller's stream


def _map_dtypes():
    global _MAP_LEVEL, _MAP_JOB
    if _MAP_LEVEL is None:
        import numpy as np
        _MAP_LEVEL = np.dtype([("coords4", "<u8"), ("keys", "<u8"), ("vals", "<u8"), ("cap", "<i8"), ("rows", "<i8")])
        _MAP_JOB = np.dtype([(n, "<i4") for n in ("lvl_in", "lvl_out", "ksize", "scale", "self_map", "stream", "bm_fwd", "bm_bwd")] +
                            [(n, "<u8") for n in ("nbr_fwd", "nbr_bwd", "counts", "order_fwd", "sorted_fwd", "gmask_fwd", "order_bwd",
                                                  "sorted_bwd", "gmask_bwd", "tl_fwd", "tl_bwd", "pl_fwd")])
        assert _MAP_LEVEL.itemsize == 40 and _MAP_JOB.itemsize == 128
    return _MAP_LEVEL, _MAP_JOB


def _addr(v):
    return 0 if v is None else (int(v) if isinstance(v, int) else v.data_ptr())


def maps_build(levels, jobs, dev, sort_rows, streams=None):
    """levels: [(coords4, HashTable, rows)]; jobs: list of dicts with the fields of osn_map_job (tensors, device addresses
    or None for the pointers, `stream` = 0 .. MAPS_STREAMS - 1); sort_rows: rows of the largest table that gets tile-ordered (scratch
    size).  One C call; the maps of different levels run side by side on MAPS_STREAMS streams (fork / join inside)."""
    import numpy as np
    lib = _prep(dev)
    ldt, jdt = _map_dtypes()
    la = np.zeros(len(levels), dtype=ldt)
    for i, (c, t, rows) in enumerate(levels):
        la[i] = (c.data_ptr(), t.keys.data_ptr(), t.vals.data_ptr(), t.cap, rows)
    ja = np.zeros(max(len(jobs), 1), dtype=jdt)
    for i, q in enumerate(jobs):
        ja[i] = tuple((_addr(q.get(n)) if jdt[n].kind == "u" else int(q.get(n, 0))) for n in jdt.names)
    ns = max(1, min(int(streams) if streams else MAPS_STREAMS, 3))      # (streams: the caller's choice -- the executor's inference pass)
    idx = _idx(dev)
    main = _stream(dev)
    raws = [main]
    if ns > 1:
        # lanes 1, 2: the two auxiliary streams the process has anyway (the executor's side stream, the prefetcher's) -- never new ones
        pool = [s for s in (side_stream(dev), aux_stream(dev)) if s.cuda_stream != main]
        ns = min(ns, 1 + len(pool))
        raws += [s.cuda_stream for s in pool[:ns - 1]]
        ev = _map_events.get(idx)
        if ev is None:
            with _Dev(dev):
                ev = _map_events[idx] = lib.osn_events_create(8)
        if not ev:
            ns, raws, ev = 1, [main], None
    else:
        ev = None
    wsb = int(_cached("osn_kmap_sort_ws_bytes", int(sort_rows))) if sort_rows else 256
    wss = [_ws(wsb, dev)] + [ws_on(wsb, dev, r) for r in raws[1:]]
    st_arr = (ctypes.c_void_p * ns)(*raws)
    ws_arr = (ctypes.c_void_p * ns)(*[w.data_ptr() for w in wss])
    wb_arr = (ctypes.c_uint64 * ns)(*[w.numel() for w in wss])
    with _Dev(dev):
        check(lib.osn_maps_build(la.ctypes.data, len(levels), ja.ctypes.data, len(jobs), st_arr, ws_arr, wb_arr, ns, ev),
              "osn_maps_build")


# ----------------------------------------------------------------- convolution
def _w3(weight):
    return weight.unsqueeze(0) if weight.dim() == 2 else weight


def spconv_fwd(feats, weight, nbr, n_out, out_rows=None, gmask=None):
    """out[o] = sum_k feats[nbr[k,o]] @ weight[k].  nbr None <=> K == 1 identity map.
    (out_rows, gmask) come with a tile-ordered table from kmap_sort."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    if feats.shape[1] != cin:
        raise ValueError("features have %d channels, kernel expects %d" % (feats.shape[1], cin))
    if nbr is not None:
        if nbr.dtype != torch.int32 or nbr.shape != (K, n_out):
            raise ValueError("nbr must be int32 [%d, %d], got %s %s" % (K, n_out, nbr.dtype, tuple(nbr.shape)))
        nbr = nbr.contiguous()
    elif K != 1 or feats.shape[0] != n_out:
        raise ValueError("nbr=None is the identity map and needs K == 1 and n_in == n_out")
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    wsb = _cached("osn_spconv_fwd_ws_bytes", n_out, K, cin, cout)
    ws = _ws(wsb, dev) if wsb else None
    tok = _prof_start("spconv_fwd", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_spconv_fwd(_p(feats), _p(w), _p(nbr), _p(out_rows), _p(gmask), _p(out), n_out, K, cin, cout,
                                 _p(ws), int(wsb), _stream(dev)), "osn_spconv_fwd")
    if tok is not None:
        _profiler.stop(tok)
    return out


def weight_prep_x6(weight, flip=False, for_dgrad=False):
    """bf16 [3, K, n, c_pad] pre-split weights for spconv_fwd_x6 (n = output channels of the conv that will run)."""
    dev = weight.device
    lib = _prep(dev)
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    nn, nc = (cin, cout) if for_dgrad else (cout, cin)
    cp = (nc + 31) // 32 * 32
    wp = torch.empty((3, K, nn, cp), dtype=torch.bfloat16, device=dev)
    with _Dev(dev):
        check(lib.osn_weight_prep_x6(_p(w), K, cin, cout, int(bool(flip)), int(bool(for_dgrad)), _p(wp), _stream(dev)),
              "osn_weight_prep_x6")
    return wp


def weight_prep_x6_pair(weight, flip=False):
    """(forward planes, input-gradient planes) of one weight from a single launch."""
    dev = weight.device
    lib = _prep(dev)
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    wf = torch.empty((3, K, cout, (cin + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
    wb = torch.empty((3, K, cin, (cout + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)
    with _Dev(dev):
        check(lib.osn_weight_prep_x6_pair(_p(w), K, cin, cout, int(bool(flip)), _p(wf), _p(wb), _stream(dev)),
              "osn_weight_prep_x6_pair")
    return wf, wb


def spconv_fwd_x6(feats, wp, nbr, n_out, out_rows=None, gmask=None):
    """Split-bf16 convolution: out[o] = sum_k feats[nbr[k,o]] @ B[k], B given as weight_prep_x6 planes."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    _three, K, cout, cp = wp.shape
    cin = feats.shape[1]
    if (cin + 31) // 32 * 32 != cp:
        raise ValueError("features have %d channels, prepared weights expect <= %d" % (cin, cp))
    if nbr is not None:
        if nbr.dtype != torch.int32 or nbr.shape != (K, n_out):
            raise ValueError("nbr must be int32 [%d, %d], got %s %s" % (K, n_out, nbr.dtype, tuple(nbr.shape)))
        nbr = nbr.contiguous()
    elif K != 1 or feats.shape[0] != n_out:
        raise ValueError("nbr=None is the identity map and needs K == 1 and n_in == n_out")
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    wsb = _cached("osn_spconv_fwd_ws_bytes", n_out, K, cin, cout)
    ws = _ws(wsb, dev) if wsb else None
    tok = _prof_start("spconv_fwd_x6", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_spconv_fwd_x6(_p(feats), _p(wp), _p(nbr), _p(out_rows), _p(gmask), _p(out), n_out, K, cin, cout,
                                    _p(ws), int(wsb), _stream(dev)), "osn_spconv_fwd_x6")
    if tok is not None:
        _profiler.stop(tok)
    return out


# ------------------------------------------------- second-generation convolution (tile lists)
class TileLists:
    """Per-tile compacted pair lists of one neighbour table (osn_tile_lists_build): `buf` holds
    cnt int32 [n_tiles, K] and lst int2 [n_tiles, K, bm]; `out_rows` is the row permutation of a
    tile-ordered table (None = table rows are tensor rows)."""
    __slots__ = ("buf", "bm", "n_out", "K", "out_rows", "pairs")

    def __init__(self, buf, bm, n_out, K, out_rows):
        self.buf, self.bm, self.n_out, self.K, self.out_rows = buf, bm, n_out, K, out_rows
        self.pairs = None                  # per-offset pair arrays for the weight gradient, built on first use

    @property
    def n_tiles(self):
        return -(-self.n_out // self.bm)

    def counts(self):
        """int32 [n_tiles, K] view of the pair counts."""
        return self.buf[:self.n_tiles * self.K * 4].view(torch.int32).view(self.n_tiles, self.K)

    def lists(self):
        """int32 [n_tiles, K, bm, 2] view: (input row, local output row); only the first cnt entries are defined."""
        off = (self.n_tiles * self.K * 4 + 255) // 256 * 256
        return self.buf[off:off + self.n_tiles * self.K * self.bm * 8].view(torch.int32).view(self.n_tiles, self.K, self.bm, 2)


def tile_rows(n_out):
    return int(_cached("osn_tile_rows", int(n_out)))


def tile_lists(nbr, out_rows=None, bm=None):
    """TileLists of an int32 [K, n_out] table (rows possibly tile-ordered; then pass its `out_rows`)."""
    dev = nbr.device
    lib = _prep(dev)
    nbr = nbr.contiguous()
    K, n_out = nbr.shape
    bm = tile_rows(n_out) if bm is None else int(bm)
    nbytes = int(_cached("osn_tile_lists_bytes", n_out, K, bm))
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with _Dev(dev):
        check(lib.osn_tile_lists_build(_p(nbr), n_out, K, bm, _p(buf), _stream(dev)), "osn_tile_lists_build")
    return TileLists(buf, bm, n_out, K, out_rows)


def tl_eligible(K, cin, cout, n_in=0):
    """Shapes the tile-list kernel takes (everything of the U-Net but the 3-channel stem; up to 2^24 input rows; up to 512
    input channels = four 128-channel chunks in the kernel's step table: wider 1x1 convs are the dense kernel's)."""
    return cin % 4 == 0 and 8 <= cin <= 512 and cout % 4 == 0 and K <= 128 and n_in <= (1 << 24)


def weight_prep_tl(weight, flip=False, want_fwd=True, want_dgrad=True):
    """(forward image, input-gradient image) of one weight, MFMA-fragment layout, one launch."""
    dev = weight.device
    lib = _prep(dev)
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    wf = torch.empty(_cached("osn_weight_prep_tl_bytes", K, cin, cout, 0), dtype=torch.uint8, device=dev) if want_fwd else None
    wb = torch.empty(_cached("osn_weight_prep_tl_bytes", K, cin, cout, 1), dtype=torch.uint8, device=dev) if want_dgrad else None
    with _Dev(dev):
        check(lib.osn_weight_prep_tl(_p(w), K, cin, cout, int(bool(flip)), _p(wf), _p(wb), _stream(dev)),
              "osn_weight_prep_tl")
    return wf, wb


# ------------------------------------------------- weight images of model parameters: one launch per optimizer step
# A convolution weight that is an nn.Parameter keeps its images (forward / input gradient, plane or fragment layout)
# between calls, keyed on the parameter's version counter: an optimizer step bumps every version, the first
# convolution of the next forward then refreshes ALL images of the device in one osn_weight_prep_batch launch (the job
# table lives in device memory and is reused while the set of stale images is the same); in eval mode nothing is
# launched.  Anything that changes a parameter without bumping its version (writes through `.data` / raw pointers)
# must call clear_weight_cache().  OSN_WEIGHT_CACHE=0 switches the cache off (one prep launch per convolution).
WEIGHT_CACHE = os.environ.get("OSN_WEIGHT_CACHE", "1") != "0"
PREP_X6, PREP_TL = 0, 1


class _Image:
    __slots__ = ("ref", "ptr", "version", "image", "K", "cin", "cout", "flip", "for_dgrad", "layout", "blocks")


class _WeightImages:
    def __init__(self):
        self.entries = {}          # (id(param), flip, for_dgrad, layout) -> _Image
        self.table = None          # (tuple of entry keys, job table on the device, total blocks)
        self.keep = []             # recent job tables (their launches may still be queued)
        self.lock = threading.Lock()


_weight_images = {}


WEIGHT_CACHE_GENERATION = 0          # moves when the cache is emptied (the executor's descriptor cache keys on it)


def clear_weight_cache():
    global WEIGHT_CACHE_GENERATION
    WEIGHT_CACHE_GENERATION += 1
    _weight_images.clear()


def _alloc_image(K, cin, cout, for_dgrad, layout, dev):
    if layout == PREP_TL:
        return torch.empty(_cached("osn_weight_prep_tl_bytes", K, cin, cout, int(for_dgrad)), dtype=torch.uint8, device=dev)
    nn, nc = (cin, cout) if for_dgrad else (cout, cin)
    return torch.empty((3, K, nn, (nc + 31) // 32 * 32), dtype=torch.bfloat16, device=dev)


def _refresh_images(imgs, lib, dev):
    import numpy as np
    stale, keys = [], []
    for key, e in list(imgs.entries.items()):
        p = e.ref()
        if p is None:
            del imgs.entries[key]
            continue
        if e.version != p._version or e.ptr != p.data_ptr():
            stale.append((e, p))
            keys.append(key)
    if not stale:
        return
    keys = tuple(keys)
    tbl = imgs.table
    if tbl is None or tbl[0] != keys or any(e.ptr != p.data_ptr() for e, p in stale):
        jobs = np.zeros(len(stale), dtype=[("W", "<u8"), ("out", "<u8"), ("first", "<i8"), ("K", "<i4"), ("cin", "<i4"),
                                           ("cout", "<i4"), ("flip", "<i4"), ("dg", "<i4"), ("layout", "<i4")])
        first = 0
        for i, (e, p) in enumerate(stale):
            e.ptr = p.data_ptr()
            jobs[i] = (e.ptr, e.image.data_ptr(), first, e.K, e.cin, e.cout, e.flip, e.for_dgrad, e.layout)
            first += e.blocks
        # (under ops.on_stream the launch below goes to that stream: the table's upload has to be in ITS past, not in torch's
        # current stream's)
        ts = getattr(_tls, "torch_stream", None)
        if ts is not None:
            with torch.cuda.stream(ts):
                tbl = (keys, torch.from_numpy(jobs.view(np.uint8)).to(dev), first)
        else:
            tbl = (keys, torch.from_numpy(jobs.view(np.uint8)).to(dev), first)
        imgs.table = tbl
        imgs.keep = (imgs.keep + [tbl[1]])[-8:]
    with _Dev(dev):
        check(lib.osn_weight_prep_batch(_p(tbl[1]), len(stale), tbl[2], _stream(dev)), "osn_weight_prep_batch")
    for e, p in stale:
        e.version = p._version


def weight_image(weight, flip=False, for_dgrad=False, layout=PREP_X6):
    """Image of `weight` for spconv_fwd_x6 (layout PREP_X6) or spconv_fwd_tl (PREP_TL); for_dgrad: the image the input
    gradient multiplies with (transposed, offsets mirrored if flip).  Parameters are served from the per-device cache
    (see above), other tensors are prepared on the spot."""
    flip, for_dgrad = bool(flip), bool(for_dgrad)
    cacheable = (WEIGHT_CACHE and isinstance(weight, torch.nn.Parameter) and weight.dtype == torch.float32
                 and weight.is_contiguous())
    if not cacheable:
        if layout == PREP_TL:
            wf, wb = weight_prep_tl(weight, flip, want_fwd=not for_dgrad, want_dgrad=for_dgrad)
            return wb if for_dgrad else wf
        return weight_prep_x6(weight, flip=flip, for_dgrad=for_dgrad)
    dev = weight.device
    lib = _prep(dev)
    imgs = _weight_images.get(dev)
    if imgs is None:
        imgs = _weight_images[dev] = _WeightImages()
    key = (id(weight), flip, for_dgrad, layout)
    with imgs.lock:
        e = imgs.entries.get(key)
        if e is None or e.ref() is not weight:
            K, cin, cout = _w3(weight).shape
            e = _Image()
            e.ref, e.ptr, e.version = weakref.ref(weight), 0, -1
            e.K, e.cin, e.cout, e.flip, e.for_dgrad, e.layout = K, cin, cout, int(flip), int(for_dgrad), layout
            e.blocks = _cached("osn_weight_prep_job_blocks", K, cin, cout, int(for_dgrad), layout)
            e.image = _alloc_image(K, cin, cout, for_dgrad, layout, dev)
            imgs.entries[key] = e
        if e.version != weight._version or e.ptr != weight.data_ptr():
            _refresh_images(imgs, lib, dev)
        return e.image


_tl_counters = {}


def tl_counters(dev):
    """The 128 persistent tile counters of (device, current stream): zero once, the tile-list kernel leaves them zero."""
    ck = (_idx(dev), _stream(dev))
    c = _tl_counters.get(ck)
    if c is None:
        c = _tl_counters[ck] = torch.zeros(128, dtype=torch.int32, device=dev)
    return c


def spconv_fwd_tl(feats, wp, tl, n_out, K, cout, bn_partial=None):
    """out[o] = sum_k feats[list rows] @ B[k] with B given as a weight_prep_tl image; tl None <=> K == 1 identity.
    bn_partial: optional float64 [n_tiles, 2, cout] receiving per-tile column sums / sums of squares."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    cin = feats.shape[1]
    if tl is not None:
        if tl.n_out != n_out or tl.K != K:
            raise ValueError("tile lists are for a [%d, %d] table, conv wants [%d, %d]" % (tl.K, tl.n_out, K, n_out))
        bm, buf, rows = tl.bm, tl.buf, tl.out_rows
    else:
        if K != 1 or feats.shape[0] != n_out:
            raise ValueError("tl=None is the identity map and needs K == 1 and n_in == n_out")
        bm, buf, rows = tile_rows(n_out), None, None
    need = _cached("osn_weight_prep_tl_bytes", K, cin, cout, 0)
    if wp.numel() != need:
        raise ValueError("prepared weight has %d bytes, a [%d, %d, %d] conv needs %d" % (wp.numel(), K, cin, cout, need))
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    ws = _ws(_cached("osn_spconv_fwd_tl_ws_bytes", n_out, K if tl is not None else 1, cout, bm), dev)
    tok = _prof_start("spconv_fwd_tl", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    st = _stream(dev)
    ck = (_idx(dev), st)
    counters = _tl_counters.get(ck)
    if counters is None:                # 128 tile counters per (device, stream), zero once: the kernel leaves them zero
        counters = _tl_counters[ck] = torch.zeros(128, dtype=torch.int32, device=dev)
    with _Dev(dev):
        try:
            check(lib.osn_spconv_fwd_tl_pc(_p(feats), feats.shape[0], _p(wp), _p(buf), _p(rows), _p(out), _p(bn_partial), n_out,
                                           K, cin, cout, bm, _p(ws), ws.numel(), _p(counters), st), "osn_spconv_fwd_tl_pc")
        except Exception:
            _tl_counters.pop(ck, None)      # a failed launch leaves the counters undefined: start from a fresh zero buffer
            raise
    if tok is not None:
        _profiler.stop(tok)
    return out


def pair_lists(tl):
    """Per-offset pair arrays + weight-gradient work items of a TileLists (built once, cached on it)."""
    if tl.pairs is None:
        dev = tl.buf.device
        lib = _prep(dev)
        nbytes = int(_cached("osn_pair_lists_bytes", tl.n_out, tl.K, tl.bm))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with _Dev(dev):
            check(lib.osn_pair_lists_build(_p(tl.buf), _p(tl.out_rows), tl.n_out, tl.K, tl.bm, _p(buf), _stream(dev)),
                  "osn_pair_lists_build")
        tl.pairs = buf
    return tl.pairs


def pair_arrays(tl):
    """(poff int32 [K+1], pin int32 [P], pout int32 [P]) host-readable views of pair_lists(tl) (tests / tools)."""
    buf = pair_lists(tl)
    poff = buf[:(tl.K + 1) * 4].view(torch.int32)
    cap = tl.K * max(tl.n_out, 1)
    P = int(poff[tl.K])
    pin = buf[16384:16384 + cap * 4].view(torch.int32)[:P]
    pout = buf[16384 + cap * 4:16384 + 2 * cap * 4].view(torch.int32)[:P]
    return poff, pin, pout


def spconv_wgrad_tl(feats, gout, tl, K, swap=False):
    """gW [K, cin, cout] from the pair arrays of `tl` (None <=> K == 1 identity).  swap: `tl` belongs to the strided
    convolution that this transposed convolution mirrors."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    gout = _f32c(gout, "grad_output")
    n_in, cin = feats.shape
    n_out, cout = gout.shape
    if tl is not None:
        table_rows = n_in if swap else n_out
        if tl.K != K or tl.n_out != table_rows:
            raise ValueError("pair lists are for a [%d, %d] table, the weight gradient wants [%d, %d]" % (
                tl.K, tl.n_out, K, table_rows))
        pl = pair_lists(tl)
    else:
        if K != 1 or n_in != n_out:
            raise ValueError("tl=None is the identity map and needs K == 1 and n_in == n_out")
        pl = None
    gw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    wsb = _cached("osn_spconv_wgrad_tl_ws_bytes", K, cin, cout)
    ws = _ws(wsb, dev)
    tok = _prof_start("spconv_wgrad_tl", dev, n_in=n_in, n_out=n_out, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_spconv_wgrad_tl(_p(feats), _p(gout), _p(pl), int(bool(swap)), _p(gw), n_in, n_out, K, cin, cout,
                                      _p(ws), ws.numel(), _stream(dev)), "osn_spconv_wgrad_tl")
    if tok is not None:
        _profiler.stop(tok)
    return gw


def rows_argmax(scores, gather=None):
    """int64 labels [len(gather) or N] = argmax over the columns of float32 `scores` [N, c] (a column slice of a wider
    contiguous matrix is fine), row gather[p] for point p: argmax + point -> voxel gather in one launch."""
    dev = scores.device
    lib = _prep(dev)
    if scores.dtype != torch.float32 or scores.dim() != 2 or scores.stride(1) != 1:
        raise ValueError("scores must be a float32 matrix with unit column stride")
    n, c = scores.shape
    ld = scores.stride(0) if n > 1 else max(c, scores.stride(0))
    if gather is not None:
        if gather.dtype != torch.int64 or gather.device != dev:
            raise ValueError("gather must be an int64 vector on the scores' device")
        gather = gather.contiguous()
    n_pts = n if gather is None else gather.shape[0]
    labels = torch.empty(n_pts, dtype=torch.int64, device=dev)
    with _Dev(dev):
        check(lib.osn_rows_argmax(_p(scores), ld, c, _p(gather), n_pts, n, _p(labels), _stream(dev)), "osn_rows_argmax")
    return labels


def dense_eligible(cin, cout):
    """Shapes the 1x1-convolution kernel takes (everything of the U-Net family; odd widths stay on the generic kernel)."""
    return cin % 4 == 0 and cin >= 8 and cout % 4 == 0


def dense_fwd(feats, wp, cout):
    """out = feats @ B, B a weight_prep_tl image of a K = 1 weight (forward image, or the input-gradient image)."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    n, cin = feats.shape
    need = _cached("osn_weight_prep_tl_bytes", 1, cin, cout, 0)
    if wp.numel() != need:
        raise ValueError("prepared weight has %d bytes, a [%d, %d] 1x1 conv needs %d" % (wp.numel(), cin, cout, need))
    out = torch.empty((n, cout), dtype=torch.float32, device=dev)
    tok = _prof_start("dense_fwd", dev, n_in=n, n_out=n, K=1, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_dense_fwd(_p(feats), _p(wp), _p(out), n, cin, cout, _stream(dev)), "osn_dense_fwd")
    if tok is not None:
        _profiler.stop(tok)
    return out


def stem_conv_wgrad(feats, gout, nbr, K):
    """gW [K, cin, 32] of the stem convolution (stem_eligible shapes) from the plain neighbour table."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    gout = _f32c(gout, "grad_output")
    cin, (n_out, cout) = feats.shape[1], gout.shape
    if nbr.dtype != torch.int32 or tuple(nbr.shape) != (K, n_out):
        raise ValueError("nbr must be int32 [%d, %d], got %s %s" % (K, n_out, nbr.dtype, tuple(nbr.shape)))
    gw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    ws = _ws(_cached("osn_stem_conv_wgrad_ws_bytes", K, cin), dev)
    tok = _prof_start("stem_wgrad", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_stem_conv_wgrad(_p(feats), _p(gout), _p(nbr.contiguous()), _p(gw), n_out, K, cin, cout, _p(ws), ws.numel(),
                                      _stream(dev)), "osn_stem_conv_wgrad")
    if tok is not None:
        _profiler.stop(tok)
    return gw


def spconv_fwd_ws(feats, wp, tl, nbr_dst, n_dst, K, cout, swap=False, direct=False):
    """Convolution of a small map from the pair arrays of `tl` (weight-stationary workgroups + ordered sum over the
    offsets): out[r] = sum_k feats[src of (k, r)] @ B[k], B a weight_prep_tl image.  nbr_dst int32 [K, n_dst]: the
    neighbour table of the destination side, plain row order.  swap: `tl` belongs to the map's transposed direction.
    direct: every destination row has exactly one pair in the map (fine side of a 2^3 stride-2 map): rows are written
    straight to the output, nbr_dst may be None."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    n_in, cin = feats.shape
    table_rows = n_in if swap else n_dst
    if tl.K != K or tl.n_out != table_rows:
        raise ValueError("pair lists are for a [%d, %d] table, the convolution wants [%d, %d]" % (tl.K, tl.n_out, K, table_rows))
    if not (direct and nbr_dst is None) and (nbr_dst.dtype != torch.int32 or tuple(nbr_dst.shape) != (K, n_dst)):
        raise ValueError("nbr_dst must be int32 [%d, %d], got %s %s" % (K, n_dst, nbr_dst.dtype, tuple(nbr_dst.shape)))
    need = _cached("osn_weight_prep_tl_bytes", K, cin, cout, 0)
    if wp.numel() != need:
        raise ValueError("prepared weight has %d bytes, a [%d, %d, %d] conv needs %d" % (wp.numel(), K, cin, cout, need))
    pl = pair_lists(tl)
    out = torch.empty((n_dst, cout), dtype=torch.float32, device=dev)
    ws = _ws(_cached("osn_spconv_fwd_ws_ws_bytes", n_dst, K, cout, int(bool(direct))), dev)
    tok = _prof_start("spconv_fwd_ws", dev, n_in=n_in, n_out=n_dst, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_spconv_fwd_ws(_p(feats), n_in, _p(wp), _p(pl), table_rows, int(bool(swap)), int(bool(direct)),
                                    _p(nbr_dst.contiguous() if nbr_dst is not None else None),
                                    _p(out), n_dst, K, cin, cout, _p(ws), ws.numel(), _stream(dev)), "osn_spconv_fwd_ws")
    if tok is not None:
        _profiler.stop(tok)
    return out


def rg_eligible(K, cin, cout, n_in):
    """Shapes the register-gather kernel takes (csrc/spconv_rg.hip): 32 / 64 channels on both sides, K > 1."""
    return bool(_cached("osn_spconv_fwd_rg_ok", int(max(n_in, 1)), int(K), int(cin), int(cout)))


def spconv_fwd_rg(feats, wp, nbr, n_out, cout, out_rows=None):
    """out[o] = sum_k feats[nbr[k, o]] @ B[k], B given as a weight_prep_tl image (forward or input-gradient image); nbr plain or
    tile-ordered (then out_rows = its permutation)."""
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    nbr = nbr.contiguous()
    K = nbr.shape[0]
    if nbr.shape[1] != n_out:
        raise ValueError("table is [%d, %d], conv wants %d output rows" % (K, nbr.shape[1], n_out))
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    with _Dev(dev):
        check(lib.osn_spconv_fwd_rg(_p(feats), feats.shape[0], _p(wp), _p(nbr), _p(out_rows), _p(out), n_out, K, feats.shape[1], cout,
                                    _stream(dev)), "osn_spconv_fwd_rg")
    return out


def stem_eligible(K, cin, cout):
    """The dedicated kernels of the U-Net's 3-channel stem conv (any conv with <= 4 input and 32 output channels)."""
    return cin <= 4 and cout == 32 and 1 < K <= 125


def stem_conv_fwd(feats, weight, nbr, n_out):
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    if nbr.dtype != torch.int32 or nbr.shape != (K, n_out):
        raise ValueError("nbr must be int32 [%d, %d], got %s %s" % (K, n_out, nbr.dtype, tuple(nbr.shape)))
    out = torch.empty((n_out, cout), dtype=torch.float32, device=dev)
    tok = _prof_start("stem_fwd", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    with _Dev(dev):
        check(lib.osn_stem_conv_fwd(_p(feats), _p(w), _p(nbr.contiguous()), _p(out), n_out, K, cin, cout, _stream(dev)),
              "osn_stem_conv_fwd")
    if tok is not None:
        _profiler.stop(tok)
    return out


def x6_eligible(K, cin, cout, n_out):
    """The split-bf16 kernel handles every conv of the U-Net except the 3-channel stem."""
    if cin % 4 or cin < 8:
        return False
    if 3 * K * cout * ((cin + 31) // 32 * 32) >= 1 << 30:        # the kernel indexes the prepared weight with 32 bits
        return False
    plan = spconv_fwd_plan(n_out, K, cin, cout)
    S = plan[4]
    return -(-K // S) <= 32


def weight_transpose(weight, flip):
    dev = weight.device
    lib = _prep(dev)
    w = _f32c(_w3(weight), "weight")
    K, cin, cout = w.shape
    wt = torch.empty((K, cout, cin), dtype=torch.float32, device=dev)
    with _Dev(dev):
        check(lib.osn_weight_transpose(_p(w), K, cin, cout, int(bool(flip)), _p(wt), _stream(dev)),
              "osn_weight_transpose")
    return wt


_wgrad_items = {}      # id(counts tensor of a map) -> (weakref to it, {items_bytes: work-item table})


def _wgrad_plan_items(lib, counts, n_out, K, cin, cout, dev):
    """Work-item table of a map, built once per (map, table size) and shared by every conv on that map."""
    nbytes = _cached("osn_spconv_wgrad_items_bytes", n_out, K, cin, cout)
    key = id(counts)
    hit = _wgrad_items.get(key)
    if hit is None or hit[0]() is not counts:
        # entry dies with the map's counts tensor (tensors compare elementwise, so no WeakKeyDictionary)
        hit = _wgrad_items[key] = (weakref.ref(counts, lambda _r, k=key: _wgrad_items.pop(k, None)), {})
    per_map = hit[1]
    # a transposed conv shares its counts tensor with the strided conv it mirrors but has another n_out
    sub = (int(n_out), int(K), int(cin), int(cout))         # the plan depends on the channel tiling, not only on its size
    items = per_map.get(sub)
    if items is None:
        items = per_map[sub] = torch.empty(nbytes // 4, dtype=torch.int32, device=dev)
        with _Dev(dev):
            check(lib.osn_spconv_wgrad_plan(_p(counts), n_out, K, cin, cout, _p(items), _stream(dev)),
                  "osn_spconv_wgrad_plan")
    return items


def spconv_wgrad(feats, gout, nbr, K, counts=None):
    dev = feats.device
    lib = _prep(dev)
    feats = _f32c(feats, "features")
    gout = _f32c(gout, "grad_output")
    n_out, cout = gout.shape
    cin = feats.shape[1]
    if nbr is not None:
        nbr = nbr.contiguous()
        if nbr.shape != (K, n_out):
            raise ValueError("nbr shape %s does not match (K=%d, n_out=%d)" % (tuple(nbr.shape), K, n_out))
    gw = torch.empty((K, cin, cout), dtype=torch.float32, device=dev)
    wsb = _cached("osn_spconv_wgrad_ws_bytes", n_out, K, cin, cout)
    ws = _ws(wsb, dev) if wsb else None
    tok = _prof_start("spconv_wgrad", dev, n_in=feats.shape[0], n_out=n_out, K=K, cin=cin, cout=cout)
    items = _wgrad_plan_items(lib, counts, n_out, K, cin, cout, dev) if (counts is not None and n_out > 0) else None
    with _Dev(dev):
        check(lib.osn_spconv_wgrad(_p(feats), _p(gout), _p(nbr), _p(counts), _p(items), _p(gw), n_out, K, cin, cout,
                                   _p(ws), int(wsb), _stream(dev)), "osn_spconv_wgrad")
    if tok is not None:
        _profiler.stop(tok)
    return gw


_plan_cache = {}


def spconv_fwd_plan(n_out, K, cin, cout):
    """(WM, WN, TN, BK, S, workgroups) of the kernel instance osn_spconv_fwd picks."""
    key = (int(n_out), int(K), int(cin), int(cout))
    v = _plan_cache.get(key)
    if v is None:
        plan = (ctypes.c_int32 * 6)()
        check(_lib.load().osn_spconv_fwd_plan(key[0], key[1], key[2], key[3], plan), "osn_spconv_fwd_plan")
        v = _plan_cache[key] = tuple(plan)
    return v


# ------------------------------------------------------------------ batch norm
def bn_stats(x, running_mean=None, running_var=None, momentum=0.1):
    dev = x.device
    lib = _prep(dev)
    x = _f32c(x, "x")
    n, c = x.shape
    mean = torch.empty(c, dtype=torch.float32, device=dev)
    var = torch.empty(c, dtype=torch.float32, device=dev)
    wsb = _cached("osn_bn_ws_bytes", n, c)
    ws = _ws(wsb, dev)
    with _Dev(dev):
        check(lib.osn_bn_stats(_p(x), n, c, _p(mean), _p(var), _p(running_mean), _p(running_var), float(momentum),
                               _p(ws), ws.numel(), _stream(dev)), "osn_bn_stats")
    return mean, var


def bn_apply(x, mean, var, gamma, beta, eps, residual=None, relu=False):
    dev = x.device
    lib = _prep(dev)
    x = _f32c(x, "x")
    n, c = x.shape
    if residual is not None:
        residual = _f32c(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("residual shape %s != %s" % (tuple(residual.shape), tuple(x.shape)))
    y = torch.empty_like(x)
    with _Dev(dev):
        check(lib.osn_bn_apply(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), float(eps), _p(residual), int(bool(relu)),
                               _p(y), n, c, _stream(dev)), "osn_bn_apply")
    return y


def bn_forward_train(x, gamma, beta, eps, residual, relu, running_mean, running_var, momentum):
    """(y, mean, var): batch statistics (running buffers updated in place) + normalise (+ residual) (+ ReLU), one C call."""
    dev = x.device
    lib = _prep(dev)
    x = _f32c(x, "x")
    n, c = x.shape
    if residual is not None:
        residual = _f32c(residual, "residual")
        if residual.shape != x.shape:
            raise ValueError("residual shape %s != %s" % (tuple(residual.shape), tuple(x.shape)))
    mv = torch.empty((2, c), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    ws = _ws(_cached("osn_bn_ws_bytes", n, c), dev)
    with _Dev(dev):
        check(lib.osn_bn_forward_train(_p(x), n, c, _p(gamma), _p(beta), float(eps), _p(residual), int(bool(relu)),
                                       float(momentum), _p(mv[0]), _p(mv[1]), _p(running_mean), _p(running_var), _p(y),
                                       _p(ws), ws.numel(), _stream(dev)), "osn_bn_forward_train")
    return y, mv[0], mv[1]


def bn_backward(x, y, gy, mean, var, gamma, eps, relu, training, want_gres, beta=None):
    """beta given and y None (batch norm + ReLU without a residual): the ReLU mask is recomputed from x, y is not read."""
    return bn_backward_multi(x, y, [_f32c(gy, "grad_output")], mean, var, gamma, eps, relu, training, want_gres, beta=beta)


def _row_view(t, c, name):
    """(pointer, row stride in floats) of a float32 [n, c] matrix that may be a column window of a wider one."""
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != c or (t.shape[0] > 1 and t.stride(1) != 1):
        raise TypeError("%s must be a float32 [n, %d] matrix with unit column stride" % (name, c))
    ld = t.stride(0) if t.shape[0] > 1 else max(c, t.stride(0))
    if ld % 4 or t.data_ptr() % 16:
        raise ValueError("%s: row stride and first element must be 16-byte aligned" % name)
    return t.data_ptr(), ld


def bn_forward_train2(x, gamma, beta, eps, residual, relu, running_mean, running_var, momentum, y2):
    """bn_forward_train that ALSO stores the result into `y2`, a [n, c] column window of a wider matrix (ME.cat written in
    place by its producer).  -> (y, mean, var)."""
    dev = x.device
    lib = _prep(dev)
    x = _f32c(x, "x")
    n, c = x.shape
    if residual is not None:
        residual = _f32c(residual, "residual")
    p2, ld2 = _row_view(y2, c, "y2")
    mv = torch.empty((2, c), dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    ws = _ws(_cached("osn_bn_ws_bytes", n, c), dev)
    with _Dev(dev):
        check(lib.osn_bn_forward_train2(_p(x), n, c, _p(gamma), _p(beta), float(eps), _p(residual), int(bool(relu)),
                                        float(momentum), _p(mv[0]), _p(mv[1]), _p(running_mean), _p(running_var), _p(y), p2, ld2,
                                        _p(ws), ws.numel(), _stream(dev)), "osn_bn_forward_train2")
    return y, mv[0], mv[1]


def bn_backward_multi(x, y, gys, mean, var, gamma, eps, relu, training, want_gres, beta=None):
    """bn_backward whose incoming gradient is the SUM of the matrices in `gys` (1 .. 3, each possibly a column window of a
    wider matrix): the sum is formed while reading.  y None with relu needs beta (mask recomputed from x; no residual)."""
    dev = x.device
    lib = _prep(dev)
    n, c = x.shape
    if relu and y is None and (beta is None or want_gres):
        raise ValueError("batch-norm backward with ReLU needs y, or beta and no residual")
    views = [_row_view(g, c, "grad_output[%d]" % i) for i, g in enumerate(gys)]
    ptrs = (ctypes.c_void_p * len(views))(*[v[0] for v in views])
    lds = (ctypes.c_int64 * len(views))(*[v[1] for v in views])
    gx = torch.empty_like(x)
    gres = torch.empty_like(x) if want_gres else None
    ggamma = torch.empty(c, dtype=torch.float32, device=dev)
    gbeta = torch.empty(c, dtype=torch.float32, device=dev)
    ws = _ws(_cached("osn_bn_ws_bytes", n, c), dev)
    with _Dev(dev):
        check(lib.osn_bn_backward_multi2(_p(x), _p(y), ptrs, lds, len(views), _p(mean), _p(var), _p(gamma),
                                         _p(beta) if (relu and y is None) else None, float(eps),
                                         int(bool(relu)), int(bool(training)), _p(gx), _p(gres), _p(ggamma), _p(gbeta), n, c,
                                         _p(ws), ws.numel(), _stream(dev)), "osn_bn_backward_multi2")
    return gx, gres, ggamma, gbeta


# ------------------------------------------------------- elementwise (a11)
def relu_fwd(x):
    dev = x.device
    lib = _prep(dev)
    x = _f32c(x, "x")
    y = torch.empty_like(x)
    with _Dev(dev):
        check(lib.osn_relu_fwd(_p(x), _p(y), x.numel(), _stream(dev)), "osn_relu_fwd")
    return y


def relu_bwd(y, gy):
    dev = y.device
    lib = _prep(dev)
    gy = _f32c(gy, "grad_output")
    gx = torch.empty_like(y)
    with _Dev(dev):
        check(lib.osn_relu_bwd(_p(y), _p(gy), _p(gx), y.numel(), _stream(dev)), "osn_relu_bwd")
    return gx


def add(a, b):
    dev = a.device
    lib = _prep(dev)
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if a.shape != b.shape:
        raise ValueError("add: shapes %s and %s differ" % (tuple(a.shape), tuple(b.shape)))
    out = torch.empty_like(a)
    with _Dev(dev):
        check(lib.osn_add(_p(a), _p(b), _p(out), a.numel(), _stream(dev)), "osn_add")
    return out


def cat2(a, b):
    """[n, ca] ++ [n, cb] -> [n, ca + cb] in one launch (channel counts multiples of 4)."""
    dev = a.device
    lib = _prep(dev)
    a, b = _f32c(a, "a"), _f32c(b, "b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError("cat2: need two [n, c] matrices with equal n")
    n, ca, cb = a.shape[0], a.shape[1], b.shape[1]
    out = torch.empty((n, ca + cb), dtype=torch.float32, device=dev)
    with _Dev(dev):
        check(lib.osn_cat2(_p(a), ca, _p(b), cb, _p(out), n, _stream(dev)), "osn_cat2")
    return out


def cat2_bwd(gout, ca, cb):
    dev = gout.device
    lib = _prep(dev)
    gout = _f32c(gout, "grad_output")
    n = gout.shape[0]
    ga = torch.empty((n, ca), dtype=torch.float32, device=dev)
    gb = torch.empty((n, cb), dtype=torch.float32, device=dev)
    with _Dev(dev):
        check(lib.osn_cat2_bwd(_p(gout), _p(ga), ca, _p(gb), cb, n, _stream(dev)), "osn_cat2_bwd")
    return ga, gb


# ----------------------------------------------------------------------- query
def cosine_query(feats, text_half, gather=None, want_scores=True):
    """(scores fp16 [n, C] or None, argmax int64 [n]) of feats[gather].half() @ text.t()."""
    dev = feats.device
    lib = _prep(dev)
    feats = _query_feats(feats, "features")
    if text_half.dtype != torch.float16:
        raise TypeError("text features must be float16 (util/util.py:41-44 produces fp16)")
    text_half = text_half.contiguous()
    c, d = text_half.shape
    if feats.shape[1] != d:
        raise ValueError("feature dim %d != text dim %d" % (feats.shape[1], d))
    if gather is not None:
        if gather.dtype != torch.int64:
            gather = gather.long()
        gather = gather.contiguous()
        n = gather.shape[0]
    else:
        n = feats.shape[0]
    scores = torch.empty((n, c), dtype=torch.float16, device=dev) if want_scores else None
    amax = torch.empty(n, dtype=torch.int64, device=dev)
    with _Dev(dev):
        check(lib.osn_cosine_query(_p(feats), _p(gather), _p(text_half), _p(scores), _p(amax), n, d, c, _stream(dev)),
              "osn_cosine_query")
    return scores, amax


def query_ensemble(feat_distill, feat_fusion, text_half, gather_distill=None, gather_fusion=None, want_scores=True):
    dev = feat_distill.device
    lib = _prep(dev)
    fd = _query_feats(feat_distill, "distill features")
    ff = _query_feats(feat_fusion, "fusion features")
    text_half = text_half.contiguous()
    c, d = text_half.shape

    def _g(g):
        if g is None:
            return None
        return (g if g.dtype == torch.int64 else g.long()).contiguous()

    gd, gf = _g(gather_distill), _g(gather_fusion)
    n = gd.shape[0] if gd is not None else fd.shape[0]
    nf = gf.shape[0] if gf is not None else ff.shape[0]
    if n != nf:
        raise ValueError("the two feature sources address %d vs %d points" % (n, nf))
    scores = torch.empty((n, c), dtype=torch.float16, device=dev) if want_scores else None
    amax = torch.empty(n, dtype=torch.int64, device=dev)
    sel = torch.empty(n, dtype=torch.uint8, device=dev)
    wsb = lib.osn_query_ensemble_ws_bytes(n)
    ws = _ws(wsb, dev)
    with _Dev(dev):
        check(lib.osn_query_ensemble(_p(fd), _p(gd), _p(ff), _p(gf), _p(text_half), _p(scores), _p(amax), _p(sel), n,
                                     d, c, _p(ws), ws.numel(), _stream(dev)), "osn_query_ensemble")
    return scores, amax, sel.bool()


# ------------------------------------------------------------------- voxelizer
def voxelize_fnv(xyz, T):
    """xyz float64 [N,3] (device), T 4x4 float64 (host, numpy or tensor) ->
    (grid float64 [N,3] shifted integral coords, inds int64 [Nv], inverse int64 [N])."""
    import numpy as np
    dev = xyz.device
    lib = _prep(dev)
    if xyz.dtype != torch.float64:
        raise TypeError("xyz must be float64 (the reference voxelises in float64)")
    xyz = xyz.contiguous()
    n = xyz.shape[0]
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
    t12 = (ctypes.c_double * 12)(*T[:3, :].reshape(-1).tolist())
    grid = torch.empty((max(n, 1), 3), dtype=torch.float64, device=dev)
    inds = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    inverse = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    with _Dev(dev):
        wsb = lib.osn_voxelize_ws_bytes(n)
        ws = _ws(wsb, dev)
        nv = ctypes.c_int64(0)
        check(lib.osn_voxelize_fnv(_p(xyz), n, t12, _p(grid), _p(inds), _p(inverse), ctypes.byref(nv), _p(ws),
                                   ws.numel(), _stream(dev)), "osn_voxelize_fnv")
    return grid[:n], inds[:int(nv.value)], inverse[:n]


def fnv_hash(grid):
    dev = grid.device
    lib = _prep(dev)
    grid = grid.to(torch.float64).contiguous()
    n, ncol = grid.shape
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    with _Dev(dev):
        check(lib.osn_fnv_hash(_p(grid), n, ncol, _p(keys), _stream(dev)), "osn_fnv_hash")
    return keys


def ravel_hash(grid):
    """ravel_hash_vec of an integral float64 [n, ncol <= 4] matrix -> int64 keys (uint64 bit pattern)."""
    dev = grid.device
    lib = _prep(dev)
    grid = grid.to(torch.float64).contiguous()
    n, ncol = grid.shape
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    ws = _ws(128, dev)
    with _Dev(dev):
        check(lib.osn_ravel_hash(_p(grid), n, ncol, _p(keys), _p(ws), ws.numel(), _stream(dev)), "osn_ravel_hash")
    return keys


# ------------------------------------------------------------------ loader
def feature_remap(mask_chunk, vox_ind):
    """mask_chunk bool/uint8 [N_pts], vox_ind int64 [N_vox] (device) ->
    (mask_vox bool [N_vox], src_row int64 [N_vox] (-1 = no feature), indices int64 [n_sel])."""
    dev = vox_ind.device
    lib = _prep(dev)
    if mask_chunk.device != dev:
        raise ValueError("mask_chunk on %s but vox_ind on %s" % (mask_chunk.device, dev))
    if vox_ind.dtype != torch.int64:
        raise TypeError("vox_ind must be int64")
    m8 = mask_chunk.contiguous().view(torch.uint8) if mask_chunk.dtype == torch.bool else mask_chunk.to(torch.uint8).contiguous()
    vox_ind = vox_ind.contiguous()
    n_pts, n_vox = m8.shape[0], vox_ind.shape[0]
    mask_vox = torch.empty(max(n_vox, 1), dtype=torch.uint8, device=dev)
    src_row = torch.empty(max(n_vox, 1), dtype=torch.int64, device=dev)
    indices = torch.empty(max(n_vox, 1), dtype=torch.int64, device=dev)
    with _Dev(dev):
        wsb = lib.osn_feature_remap_ws_bytes(n_pts, n_vox)
        ws = _ws(wsb, dev)
        nsel = ctypes.c_int64(0)
        check(lib.osn_feature_remap(_p(m8), _p(vox_ind), n_pts, n_vox, _p(mask_vox), _p(src_row), _p(indices),
                                    ctypes.byref(nsel), _p(ws), ws.numel(), _stream(dev)), "osn_feature_remap")
    return mask_vox[:n_vox].view(torch.bool), src_row[:n_vox], indices[:int(nsel.value)]


def batch_coords(xyz3, batch_index, out):
    """Write (batch_index, x, y, z) int32 rows of one scene into `out` (a [n, 4] row slice of the batch)."""
    dev = xyz3.device
    lib = _prep(dev)
    if xyz3.dtype != torch.int32 or out.dtype != torch.int32 or not out.is_contiguous():
        raise TypeError("xyz3 and out must be int32, out contiguous")
    xyz3 = xyz3.contiguous()
    n = xyz3.shape[0]
    if out.shape != (n, 4):
        raise ValueError("out must be [%d, 4], got %s" % (n, tuple(out.shape)))
    with _Dev(dev):
        check(lib.osn_batch_coords(_p(xyz3), n, int(batch_index), _p(out), _stream(dev)), "osn_batch_coords")
    return out


# ------------------------------------------------------- multi-view feature fusion
def fusion_project(coords3, world_to_camera, intrinsic4, depth, image_hw, cut_bound, vis_thres):
    """int64 [n, 3] (pixel row, pixel column, visible) of fusion_util.py:93-139; coords3 float64 [n, 3] on the device,
    world_to_camera a 4x4 float64 numpy array (host), intrinsic4 = (fx, fy, cx, cy), depth float64 [H, W] device or None."""
    dev = coords3.device
    lib = _prep(dev)
    if coords3.dtype != torch.float64 or coords3.dim() != 2 or coords3.shape[1] != 3:
        raise TypeError("coords must be float64 [n, 3] (the reference concatenates with a float64 column of ones)")
    coords3 = coords3.contiguous()
    n = coords3.shape[0]
    H, W = int(image_hw[0]), int(image_hw[1])
    if depth is not None:
        if depth.dtype != torch.float64 or tuple(depth.shape) != (H, W):
            raise TypeError("depth must be float64 [%d, %d]" % (H, W))
        depth = depth.contiguous()
    m16 = (ctypes.c_double * 16)(*[float(v) for v in world_to_camera.reshape(-1)])
    i4 = (ctypes.c_double * 4)(*[float(v) for v in intrinsic4])
    mapping = torch.empty((n, 3), dtype=torch.int64, device=dev)
    with _Dev(dev):
        check(lib.osn_fusion_project(_p(coords3), n, m16, i4, _p(depth), H, W, int(cut_bound), float(vis_thres),
                                     _p(mapping), _stream(dev)), "osn_fusion_project")
    return mapping


def fusion_accumulate(feat2d, mapping, sum_features, counter, image_hw=None):
    """One view: counter[p] += 1 and sum_features[p] += feat2d[:, row, col] for the visible points (in place).
    image_hw: the (H, W) the mapping was computed for; a feature map of another size is refused (the reference's
    indexing raises IndexError there).  Without it the kernel still never reads outside feat2d: pixels beyond its
    H x W are skipped."""
    dev = feat2d.device
    lib = _prep(dev)
    feat2d = _f32c(feat2d, "feat_2d")
    D, H, W = feat2d.shape
    if image_hw is not None and (int(image_hw[0]), int(image_hw[1])) != (H, W):
        raise IndexError("feat_2d is %d x %d but the mapping was computed for a %d x %d image" % (H, W, image_hw[0], image_hw[1]))
    if mapping.device != dev or sum_features.device != dev or counter.device != dev:
        raise ValueError("feat_2d, mapping, sum_features and counter must live on one device")
    n = mapping.shape[0]
    if mapping.dtype != torch.int64 or tuple(mapping.shape) != (n, 3):
        raise TypeError("mapping must be an int64 [n, 3] tensor")
    mapping = mapping.contiguous()
    if (sum_features.dtype != torch.float32 or tuple(sum_features.shape) != (n, D) or not sum_features.is_contiguous()
            or counter.dtype != torch.float32 or counter.numel() != n or not counter.is_contiguous()):
        raise TypeError("sum_features must be contiguous float32 [n, D] and counter contiguous float32 [n] / [n, 1]")
    with _Dev(dev):
        check(lib.osn_fusion_accumulate(_p(feat2d), D, H, W, _p(mapping), n, _p(sum_features), _p(counter), _stream(dev)),
              "osn_fusion_accumulate")


def fusion_finish(sum_features, counter):
    dev = sum_features.device
    lib = _prep(dev)
    n, D = sum_features.shape
    bank = torch.empty_like(sum_features)
    with _Dev(dev):
        check(lib.osn_fusion_finish(_p(sum_features), _p(counter), n, D, _p(bank), _stream(dev)), "osn_fusion_finish")
    return bank
