#!/usr/bin/env python
"""Dry run of one training step WITHOUT a GPU: the library is re-linked against tools/dryrun/hip_null.cpp (a null HIP
runtime that logs launches), the coordinate-manager ops that need real results come from tests/cpu_backend.py, tensors
live on the host.  Prints / compares the launch sequence (kernel symbol, grid) of the per-module path and of the network
executor, and times the host side of both.

    python tools/dryrun/dry_step.py [--arch MinkUNet18A] [--points 20000] [--time 10]
"""
import argparse
import ctypes
import os
import re
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_dry_lib():
    from openscene_amd import build as b
    b.build(verbose=False)
    out = "/tmp/osn_dry"
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, "hip_null.o")
    so = os.path.join(out, "libosn_dry.so")
    subprocess.check_call([b.hipcc(), "-O1", "-fPIC", "-std=c++17", "-c", os.path.join(ROOT, "tools", "dryrun", "hip_null.cpp"), "-o", obj])
    objs = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES]
    subprocess.check_call([b.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", so, obj] + objs)
    return so


def install(so, setattr_=setattr):
    """Swap the dry library in for libopenscene_amd.so.  setattr_: e.g. pytest's monkeypatch.setattr (undone after the test)."""
    from openscene_amd import _lib, executor, ops
    import cpu_backend
    lib = ctypes.CDLL(so)
    for name, (res, args) in _lib.PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib.osn_dry_log.restype = ctypes.c_char_p
    lib.osn_dry_launches.restype = ctypes.c_longlong
    setattr_(_lib, "_lib", lib)
    setattr_(_lib, "require_device", lambda dev: None)
    setattr_(ops, "_prep", lambda dev: lib)
    setattr_(ops, "_stream", lambda dev: None)

    class NoDev:
        def __init__(self, dev): pass
        def __enter__(self): pass
        def __exit__(self, *a): pass
    setattr_(ops, "_Dev", NoDev)
    setattr_(ops, "_idx", lambda dev: 0)
    setattr_(ops, "_ws_pool", {})
    setattr_(ops, "_tl_counters", {})
    setattr_(ops, "_weight_images", {})
    setattr_(ops, "_size_cache", {})
    for n in ("HashTable", "coords_unique", "coords_pyramid", "kmap_build", "kmap_transpose", "kmap_sort", "kmap_count"):
        setattr_(ops, n, getattr(cpu_backend, n))
    setattr_(executor, "_DRY_RUN", True)
    return lib


def conv_launches(log):
    return [l for l in log.split("\n") if re.search(r"spconv|wgrad_tl_kernel|stem_fwd", l)]


def bn_launches(log):
    return [l for l in log.split("\n") if re.search(r"col_reduce|bn_", l)]


def step_logs(lib, arch="MinkUNet18A", points=20000, out_dim=64, tl_min_rows=6000, setattr_=setattr, make_model=None):
    """{'modules': log, 'executor': log} of one training step (forward + backward) of `arch` on a synthetic room."""
    from openscene_amd import executor, functional as F_, synthetic as syn
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, SparseTensor
    setattr_(F_, "TL_FWD_MIN_ROWS", tl_min_rows)    # small scene: still send level 0 through the tile-list kernels
    torch.manual_seed(0)
    model = (make_model or mink_unet)(3, out_dim, 3, arch).train()       # make_model: e.g. the reference's own factory (drop-in test)
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0, n_pts=points), 0.04), 0)
    coords = torch.from_numpy(syn.batch_coords([vox]))
    feats = torch.ones(coords.shape[0], 3)
    cm = CoordinateManager(coords)
    cm.prebuild()

    def step():
        out = model(SparseTensor(feats, coordinate_manager=cm))
        model.zero_grad(set_to_none=True)
        out.sum().backward()

    logs = {}
    for name, on in (("modules", False), ("executor", True)):
        setattr_(executor, "ENABLED", on)
        step()                                    # warm: pair lists, weight images, plans
        lib.osn_dry_reset()
        step()
        logs[name] = demangle(lib.osn_dry_log().decode())
    return logs, step, [cm.size(s) for s in (1, 2, 4, 8, 16)]


def demangle(log):
    names = sorted(set(re.findall(r"K (\S+)", log)))
    if not names:
        return log
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    table = {}
    for n, d in zip(names, out):
        d = re.sub(r"\(.*", "", d.replace("osn::", "").replace("__device_stub__", ""))
        d = re.sub(r"^void ", "", d)
        table[n] = d
    return "\n".join(re.sub(r"K (\S+)", lambda m: "K " + table[m.group(1)], line) for line in log.split("\n"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="MinkUNet18A")
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--time", type=int, default=0, help="also time N steps of each path (host cost with null launches)")
    ap.add_argument("--dump", action="store_true")
    ap.add_argument("--profile", type=int, default=0, help="cProfile N executor steps (host cost by function)")
    args = ap.parse_args()
    import collections
    lib = install(build_dry_lib())
    from openscene_amd import executor
    logs, step, sizes = step_logs(lib, args.arch, args.points)
    print("voxels per level:", sizes)
    for name in logs:
        print("%s: %d launches" % (name, len([l for l in logs[name].split("\n") if l and not l.startswith("EVENT")])))
        if args.dump:
            print(logs[name])
    # the convolution launches must agree one to one (kernel instance, grid)
    ca, cb = collections.Counter(conv_launches(logs["modules"])), collections.Counter(conv_launches(logs["executor"]))
    if ca != cb:
        print("MISMATCH in the multiset of convolution launches:")
        for k in sorted(set(ca) | set(cb)):
            if ca[k] != cb[k]:
                print("   modules %d  executor %d   %s" % (ca[k], cb[k], k))
        sys.exit(1)
    print("same multiset of %d convolution launches (kernel instance, grid)" % sum(ca.values()))
    print("BN launches equal:", collections.Counter(bn_launches(logs["modules"])) == collections.Counter(bn_launches(logs["executor"])))
    if args.time:
        lib.osn_dry_logging(0)
        for name, on in (("modules", False), ("executor", True)):
            executor.ENABLED = on
            step()
            t0 = time.perf_counter()
            for _ in range(args.time):
                step()
            print("%s: %.2f ms host time per step (null launches, maps excluded)" % (name, (time.perf_counter() - t0) * 1e3 / args.time))


    if args.profile:
        import cProfile
        import pstats
        lib.osn_dry_logging(0)
        executor.ENABLED = True
        step()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.profile):
            step()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("cumulative").print_stats(45)
        st.sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
