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

# HIP maps streams to at most this many hardware queues per process (read once, when the runtime initialises).  The training
# step uses three streams (main, weight gradients, next batch's maps); a fourth ACTIVE hardware queue -- RCCL's stream in a
# multi-GPU run -- slows every launch on this part: 13.4 ms per step against 10.1 ms with the cap at 3 (and 20 ms with 8),
# measured with a one-rank RCCL group (profiles/r03_s12_hw_queues.txt).  Streams beyond the cap share a queue.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")


def install_minkowski_alias(force=False):
    """Register ``MinkowskiEngine`` (+ ``.modules.resnet_block``, ``.utils``) in
    ``sys.modules`` so the reference's imports (models/mink_unet.py:25-26,
    models/resnet_base.py:27-28, run/distill.py:18, run/evaluate.py:18) resolve to
    the HIP implementation without editing the reference."""
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
