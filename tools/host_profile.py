#!/usr/bin/env python
"""Host-side (Python + autograd + ctypes) cost of one training step, measured WITHOUT a GPU.

The C library is replaced by a mock whose entry points return immediately (generated from
openscene_amd._lib.PROTOTYPES and compiled with gcc), the map-building ops by the CPU stand-in of
tests/cpu_backend.py (needed for real sizes), tensors live on the CPU.  What remains is exactly the
per-launch host work the GPU has to wait for when its kernels are short: wrapper code, torch.empty,
argument marshalling, autograd bookkeeping.  Prints microseconds per C-ABI call and a cProfile top list.

    python tools/host_profile.py [--steps 20] [--profile]
"""
import argparse
import cProfile
import ctypes
import os
import pstats
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_mock():
    from openscene_amd import _lib
    src = ["#include <stdint.h>", "#include <stddef.h>", "static long long calls = 0;",
           "long long mock_calls(void) { return calls; }"]
    for name, (res, args) in _lib.PROTOTYPES.items():
        params = ", ".join("void* a%d" % i if a is ctypes.c_void_p or a is None else
                           ("double a%d" % i if a in (ctypes.c_float, ctypes.c_double) else "long long a%d" % i)
                           for i, a in enumerate(args)) or "void"
        # ctypes passes c_float as float: declare those precisely
        params = ", ".join(("float a%d" % i) if a is ctypes.c_float else p
                           for (i, a), p in zip(enumerate(args), params.split(", "))) if args else "void"
        if name == "osn_spconv_fwd_plan":
            body = "int32_t* p = (int32_t*)a4; p[0]=4; p[1]=1; p[2]=3; p[3]=32; p[4]=1; p[5]=1000; return 0;"
        elif name == "osn_last_error":
            body = 'return (long long)(intptr_t)"mock";'
        elif name in ("osn_version", "osn_device_ok"):
            body = "return 1;"
        elif name.endswith("_bytes") or name == "osn_hash_capacity":
            body = "return 4096;"
        else:
            body = "++calls; return 0;"
        src.append("long long %s(%s) { %s }" % (name, params, body))
    d = tempfile.mkdtemp(prefix="osn_mock_")
    c = os.path.join(d, "mock.c")
    open(c, "w").write("\n".join(src) + "\n")
    so = os.path.join(d, "libmock.so")
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-w", "-o", so, c])
    lib = ctypes.CDLL(so)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.mock_calls.restype = ctypes.c_longlong
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--arch", default="MinkUNet18A")
    args = ap.parse_args()

    from openscene_amd import _lib, ops, synthetic as syn
    import cpu_backend
    mock = build_mock()
    _lib._lib = mock                                   # load() now returns the mock
    _lib.require_device = lambda dev: None
    ops._prep = lambda dev: mock
    ops._stream = lambda dev: None
    ops._idx = lambda dev: 0
    ops._raw_stream = lambda dev: 0

    class NoDev:
        def __init__(self, dev): pass
        def __enter__(self): pass
        def __exit__(self, *a): pass
    ops._Dev = NoDev
    _pool = {}

    def ws(nbytes, dev):
        b = _pool.get("b")
        if b is None or b.numel() < nbytes:
            b = _pool["b"] = torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
        return b
    ops._ws = ws
    for n in ("HashTable", "coords_unique", "coords_pyramid", "kmap_build", "kmap_transpose", "kmap_sort", "kmap_count"):
        setattr(ops, n, getattr(cpu_backend, n))          # real sizes for the maps (CPU, outside the timing)

    from openscene_amd.disnet import DisNet
    from openscene_amd.sparse import CoordinateManager, SparseTensor

    class Cfg:
        arch_3d = args.arch
        feature_2d_extractor = "openseg"
    torch.manual_seed(0)
    model = DisNet(Cfg())
    optim = torch.optim.Adam(model.parameters(), lr=1e-4)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0, n_pts=20000), 0.05), 0)
    coords = torch.from_numpy(syn.batch_coords([vox]))
    feats = torch.ones(coords.shape[0], 3)
    cm = CoordinateManager(coords)
    cm.prebuild()

    def step():
        out = model(SparseTensor(feats, coordinate_manager=cm))
        loss = out.sum()
        optim.zero_grad(set_to_none=True)
        loss.backward()

    for _ in range(3):
        step()
    c0 = mock.mock_calls()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt = time.perf_counter() - t0
    calls = (mock.mock_calls() - c0) / args.steps
    print("host time per step (forward + backward, maps excluded): %.2f ms for %.0f C-ABI calls = %.2f us per call"
          % (dt * 1e3 / args.steps, calls, dt * 1e6 / args.steps / calls))
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            step()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()
