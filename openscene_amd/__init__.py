"""openscene_amd -- MI355X (gfx950) implementation of ONE hot path of
pengsongyou/openscene: hash voxelisation, coordinate / kernel maps, the MinkUNet
sparse-conv backbone (forward + backward) and the per-point feature x CLIP-text
query, as hand-written HIP kernels behind a C ABI (include/openscene_amd.h).

Python surface (mirrors the reference's own, file:line in each module):
    openscene_amd.minkowski        the MinkowskiEngine symbols the reference imports
    openscene_amd.mink_unet        models/mink_unet.py  (MinkUNet14A..34C, `mink_unet` factory)
    openscene_amd.disnet           models/disnet.py     (DisNet)
    openscene_amd.voxelizer        dataset/voxelizer.py (Voxelizer)
    openscene_amd.query            run/evaluate.py:283-324 (distill / fusion / ensemble query)
    install_minkowski_alias()      make `import MinkowskiEngine` resolve to openscene_amd.minkowski

There is no CPU fallback: every op raises if libopenscene_amd.so is missing or
the tensors are not on a gfx950 device.
"""
import os
import sys
import types

__version__ = "0.1.0"

HW_QUEUES_TUNED = 3


def configure_hw_queues(n=HW_QUEUES_TUNED, log=True):
    """Cap the HIP hardware queues of THIS PROCESS (``GPU_MAX_HW_QUEUES``; read once, when the HIP runtime initialises -- call
    this before the first device call).  Opt-in: importing openscene_amd no longer touches the environment (it used to).

    Why 3: the training step runs on three streams (main, weight gradients, next batch's maps); a fourth ACTIVE hardware
    queue -- RCCL's stream in a multi-GPU run -- slowed every launch on this part: 13.4 ms per step against 10.1 ms with the
    cap at 3 and 20 ms with 8, measured with a ONE-rank RCCL group (profiles/r03_s12_hw_queues.txt).  Streams beyond the cap
    share a queue.  Its effect on an 8-rank all-reduce is unmeasured; `OSN_HW_QUEUES=default` (or any number) overrides.
    -> the value in force (None = the runtime's default)."""
    want = os.environ.get("OSN_HW_QUEUES")
    if want is not None:
        n = None if want.strip().lower() in ("default", "", "0") else int(want)
    if "GPU_MAX_HW_QUEUES" in os.environ:               # an explicit setting of the user's wins
        val = os.environ["GPU_MAX_HW_QUEUES"]
        src = "environment"
    elif n is None:
        val, src = None, "runtime default"
    else:
        val = os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
        src = "openscene_amd.configure_hw_queues"
    if log:
        print("[openscene_amd] GPU_MAX_HW_QUEUES=%s (%s)" % (val if val is not None else "unset", src), file=sys.stderr, flush=True)
    return None if val is None else int(val)


def accelerate(target):
    """Route a MinkUNet's passes through the network executor: `target` is a MinkUNet class, an instance of one -- e.g. of the
    reference's own models/mink_unet.py:28 class built on the alias -- or a module containing one (DisNet).  See drop_in.py."""
    from .drop_in import accelerate as f
    return f(target)


def install_minkowski_alias(force=False, accelerate=True):
    """Register ``MinkowskiEngine`` (+ ``.modules.resnet_block``, ``.utils``) in
    ``sys.modules`` so the reference's imports (models/mink_unet.py:25-26,
    models/resnet_base.py:27-28, run/distill.py:18, run/evaluate.py:18) resolve to
    the HIP implementation without editing the reference.

    accelerate (default): also arrange for the reference's ``models/mink_unet.py`` -- whenever it gets imported -- to run its
    forward / backward passes through the network executor (``openscene_amd.drop_in``: the class's ``forward`` becomes a
    dispatcher that falls back to the original, module-by-module code for every pass the executor does not compile)."""
    if accelerate:
        from .drop_in import install_import_hook
        install_import_hook()
    if "MinkowskiEngine" in sys.modules and not force:
        mod = sys.modules["MinkowskiEngine"]
        if getattr(mod, "__openscene_amd__", False):
            return mod
        raise RuntimeError("a different MinkowskiEngine is already imported; pass force=True to shadow it")
    from . import minkowski as mk

    me = types.ModuleType("MinkowskiEngine")
    me.__openscene_amd__ = True
    me.__version__ = mk.__version__
    for name in ("SparseTensor", "CoordinateManager", "cat", "MinkowskiConvolution", "MinkowskiConvolutionTranspose",
                 "MinkowskiBatchNorm", "MinkowskiReLU", "MinkowskiAvgPooling", "MinkowskiGlobalMaxPooling",
                 "MinkowskiLinear"):
        setattr(me, name, getattr(mk, name))
    modules = types.ModuleType("MinkowskiEngine.modules")
    resnet_block = types.ModuleType("MinkowskiEngine.modules.resnet_block")
    resnet_block.BasicBlock = mk.BasicBlock
    resnet_block.Bottleneck = mk.Bottleneck
    modules.resnet_block = resnet_block
    utils = types.ModuleType("MinkowskiEngine.utils")
    utils.kaiming_normal_ = mk.kaiming_normal_
    me.modules = modules
    me.utils = utils
    sys.modules["MinkowskiEngine"] = me
    sys.modules["MinkowskiEngine.modules"] = modules
    sys.modules["MinkowskiEngine.modules.resnet_block"] = resnet_block
    sys.modules["MinkowskiEngine.utils"] = utils
    return me
