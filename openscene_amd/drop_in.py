"""Give a MinkUNet that was NOT built from ``openscene_amd.mink_unet`` -- the reference's own ``models/mink_unet.py:28``
class, imported unchanged through the MinkowskiEngine alias -- the network executor (``executor.py``: one C call per pass,
fused conv -> BN -> ReLU stages, in-place ``cat``, one flat gradient buffer).

The reference's ``forward`` (models/mink_unet.py:116-174) walks the module tree conv by conv; its tree carries the same
attribute names as the mirror (``conv0p1s1 … final``, ``PLANES``, ``LAYERS``, ``BLOCK``, ``INIT_DIM``), which is all the
stage compiler reads.  ``accelerate`` swaps the class's ``forward`` for a dispatcher: executor when the pass is one it
compiles (float32 device tensors, arithmetic mode "tl", uniform BN flags, no input-feature gradient), the ORIGINAL
``forward`` otherwise.  ``install_minkowski_alias()`` arranges for this to happen by itself when a module named
``*.mink_unet`` is imported afterwards, so ``run/distill.py`` and ``run/evaluate.py`` keep their call sites
(``model = get_model(cfg)`` … ``output_3d = model(sinput)``) and still get the fast path.
"""
import importlib.abc
import importlib.util
import sys

import torch

_TREE = ("conv0p1s1", "bn0", "conv1p1s2", "block1", "conv4p8s2", "block4", "convtr4p16s2", "block8", "final")
_CLASS = ("PLANES", "LAYERS", "BLOCK", "INIT_DIM")


def is_unet(module):
    """Duck type of the MinkUNet family: the attribute names the stage compiler (executor.Program) reads."""
    return isinstance(module, torch.nn.Module) and all(hasattr(module, n) for n in _TREE + _CLASS)


def _dispatcher(orig):
    from . import executor, lazy_rows
    from . import functional as F_

    def forward(self, x):
        ex = executor.for_model(self)
        if ex is not None and ex.usable(x, self):
            # (inference: the output's row gather stays lazy so that the reference's query lines -- run/evaluate.py:290-292 -- reach
            #  the fused kernel unchanged; lazy_rows.py)
            return lazy_rows.wrap_output(ex.forward(self, x))
        with F_.deferred_bn_counters():
            return orig(self, x)

    forward.__osn_accelerated__ = True
    forward.__wrapped__ = orig
    forward.__doc__ = orig.__doc__
    return forward


def accelerate_class(cls):
    """Swap ``forward`` on the class of the MRO that defines it.  Idempotent.  -> True when a class was (or already is) patched."""
    from .mink_unet import MinkUNetBase
    if issubclass(cls, MinkUNetBase):
        return True                                   # the mirror dispatches by itself
    for k in cls.__mro__:
        f = k.__dict__.get("forward")
        if f is None:
            continue
        if getattr(f, "__osn_accelerated__", False):
            return True
        if k is torch.nn.Module:
            return False
        setattr(k, "forward", _dispatcher(f))
        return True
    return False


def accelerate(target):
    """target: a MinkUNet class, an instance, or any module containing one (e.g. the reference's DisNet, models/disnet.py).
    -> the target (so that ``model = accelerate(get_model(cfg))`` reads naturally).  Raises if nothing MinkUNet-shaped is found."""
    if isinstance(target, type):
        if not all(hasattr(target, n) for n in _CLASS) or not accelerate_class(target):
            raise TypeError("%s is not a MinkUNet class" % (target,))
        return target
    found = [m for m in target.modules() if is_unet(m)]
    if not found:
        raise TypeError("no MinkUNet (conv0p1s1 … final, PLANES, LAYERS, BLOCK) inside %s" % type(target).__name__)
    for m in found:
        accelerate_class(type(m))
    return target


def accelerated(module):
    """True when `module`'s forward goes through the executor dispatcher (the mirror's own forward counts)."""
    from .mink_unet import MinkUNetBase
    if isinstance(module, MinkUNetBase):
        return True
    return bool(getattr(type(module).forward, "__osn_accelerated__", False))


# ---------------------------------------------------------------------------------------------------------------- import hook
class _PostImport(importlib.abc.MetaPathFinder):
    """After a module whose last name component is ``mink_unet`` has executed, accelerate the MinkUNet base class it defines."""

    NAMES = ("mink_unet",)

    def find_spec(self, fullname, path, target=None):
        if fullname.rsplit(".", 1)[-1] not in self.NAMES or fullname.startswith("openscene_amd"):
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None and hasattr(spec.loader, "exec_module"):
                spec.loader = _Loader(spec.loader)
                return spec
        return None


class _Loader(importlib.abc.Loader):
    def __init__(self, inner):
        self.inner = inner

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        accelerate_module(module)

    def __getattr__(self, name):                      # get_code / get_source / is_package ... of the real loader
        return getattr(self.inner, name)


def accelerate_module(module):
    """Patch every MinkUNet class DEFINED in an imported module (its base class carries `forward`)."""
    done = []
    for v in list(vars(module).values()):
        if isinstance(v, type) and issubclass(v, torch.nn.Module) and all(hasattr(v, n) for n in _CLASS) \
                and v.__module__ == module.__name__ and "forward" in v.__dict__:
            if accelerate_class(v):
                done.append(v.__name__)
    return done


_HOOK = None


def install_import_hook():
    """Idempotent.  Also patches matching modules that are already imported."""
    global _HOOK
    if _HOOK is None:
        _HOOK = _PostImport()
        sys.meta_path.insert(0, _HOOK)
    for name, mod in list(sys.modules.items()):
        if mod is not None and name.rsplit(".", 1)[-1] in _PostImport.NAMES and not name.startswith("openscene_amd"):
            accelerate_module(mod)
    return _HOOK


def remove_import_hook():
    global _HOOK
    if _HOOK is not None and _HOOK in sys.meta_path:
        sys.meta_path.remove(_HOOK)
    _HOOK = None
