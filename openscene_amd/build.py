"""Build libopenscene_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m openscene_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The shared object is written to
``openscene_amd/lib/libopenscene_amd.so`` (git-ignored, but it travels with the
tree to the GPU box).  Objects are rebuilt only when a source or header is newer.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libopenscene_amd.so")
ARCH = "gfx950"
SOURCES = ["lib.hip", "coords.hip", "kmap_sort.hip", "spconv.hip", "spconv_tl.hip", "spconv_ws.hip", "spconv_rg.hip", "dense.hip", "weight_prep.hip", "wgrad_tl.hip", "stem.hip", "bn.hip", "elementwise.hip", "loss.hip", "optim.hip", "query.hip", "voxelize.hip", "loader.hip", "fusion.hip", "net.hip", "maps.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-deprecated-declarations", "-DNDEBUG"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "openscene_amd.h"))
    return hs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    flags = FLAGS + (["-DOSN_BUILD_TOOLS"] if os.environ.get("OSN_BUILD_TOOLS") == "1" else [])   # tools-only entry points (tools/prof_tl.py)
    cc = hipcc()
    hdrs = _headers()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [cc] + flags + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        if verbose:
            print("[openscene_amd.build] compiling %d file(s) for %s" % (len(jobs), ARCH), flush=True)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for (s, o), r in ex.map(compile_one, jobs):
                if r.returncode != 0:
                    if os.path.exists(o):
                        os.remove(o)
                    raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (s, r.stdout, r.stderr))
                if verbose and r.stderr.strip():
                    print(r.stderr, file=sys.stderr)
    if force or jobs or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("[openscene_amd.build] linked %s" % LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
