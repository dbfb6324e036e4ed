"""ctypes binding of libopenscene_amd.so (the C ABI declared in include/openscene_amd.h).

The product path has NO fallback: if the shared object is missing or the device
is not an MI355X, every op raises.  Nothing here imports ``oracle``.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSN_LIB_PATH") or os.path.join(_HERE, "lib", "libopenscene_amd.so")   # (override: tools' A/B builds)

_c = ctypes
_vp, _i32, _i64, _sz, _f32 = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_size_t, _c.c_float

# name -> (restype, argtypes); mirrors include/openscene_amd.h one to one
PROTOTYPES = {
    "osn_version": (_i32, []),
    "osn_last_error": (_c.c_char_p, []),
    "osn_device_ok": (_i32, []),
    "osn_hash_capacity": (_i64, [_i64]),
    "osn_coords_unique_ws_bytes": (_sz, [_i64]),
    "osn_coords_unique": (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _c.POINTER(_i64), _vp, _sz, _vp]),
    "osn_coords_unique_async": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "osn_coords_pyramid_async": (_i32, [_vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "osn_kmap_build": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "osn_kmap_build_self": (_i32, [_vp, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "osn_kmap_transpose": (_i32, [_vp, _i64, _i32, _i64, _vp, _vp]),
    "osn_kmap_sort_ws_bytes": (_sz, [_i64]),
    "osn_kmap_sort": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "osn_kmap_count": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "osn_spconv_fwd_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "osn_spconv_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_weight_prep_x6_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "osn_weight_prep_x6": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "osn_weight_prep_x6_pair": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "osn_spconv_fwd_x6": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_spconv_fwd_plan": (_i32, [_i64, _i32, _i32, _i32, _c.POINTER(_i32)]),
    "osn_tile_rows": (_i32, [_i64]),
    "osn_tile_lists_bytes": (_sz, [_i64, _i32, _i32]),
    "osn_tile_lists_build": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "osn_weight_prep_tl_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "osn_weight_prep_tl": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "osn_weight_prep_job_blocks": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "osn_weight_prep_batch": (_i32, [_vp, _i32, _i64, _vp]),
    "osn_spconv_fwd_tl_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "osn_spconv_fwd_tl": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_spconv_fwd_tl_pc": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _sz, _vp, _vp]),
    "osn_weight_transpose": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "osn_spconv_wgrad_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "osn_spconv_wgrad_items_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "osn_spconv_wgrad_plan": (_i32, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "osn_spconv_wgrad": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_pair_lists_bytes": (_sz, [_i64, _i32, _i32]),
    "osn_pair_lists_build": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "osn_spconv_wgrad_tl_ws_bytes": (_sz, [_i32, _i32, _i32]),
    "osn_spconv_wgrad_tl": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_distill_loss_state_bytes": (_sz, [_i64, _i64]),
    "osn_distill_loss_fwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "osn_distill_loss_bwd": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _sz, _vp]),
    "osn_distill_loss_bwd_rows": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "osn_rows_gather": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "osn_rows_scatter_zero": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "osn_distill_loss_check": (_i32, [_vp, _i64, _i64, _vp]),
    "osn_stem_conv_wgrad_ws_bytes": (_sz, [_i32, _i32]),
    "osn_stem_conv_wgrad": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_dense_fwd": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "osn_rows_argmax": (_i32, [_vp, _i64, _i32, _vp, _i64, _i64, _vp, _vp]),
    "osn_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _vp]),
    "osn_spconv_fwd_ws_ws_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "osn_spconv_fwd_ws": (_i32, [_vp, _i64, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "osn_spconv_wgrad_tl_partial": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _sz, _vp, _vp]),
    "osn_wgrad_tl_reduce_batch": (_i32, [_vp, _i32, _vp]),
    "osn_stem_conv_fwd": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "osn_bn_ws_bytes": (_sz, [_i64, _i32]),
    "osn_bn_stats": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _sz, _vp]),
    "osn_bn_apply": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _i64, _i32, _vp]),
    "osn_bn_forward_train": (_i32, [_vp, _i64, _i32, _vp, _vp, _f32, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "osn_bn_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32,
                               _vp, _sz, _vp]),
    "osn_bn_apply2": (_i32, [_vp, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _i64, _i64, _i32, _vp]),
    "osn_spconv_fwd_rg_ok": (_i32, [_i64, _i32, _i32, _i32]),
    "osn_spconv_fwd_rg": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "osn_bn_forward_train2": (_i32, [_vp, _i64, _i32, _vp, _vp, _f32, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _sz, _vp]),
    "osn_bn_backward_multi": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32,
                                     _vp, _sz, _vp]),
    "osn_bn_backward_multi2": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _i32,
                                     _vp, _sz, _vp]),
    "osn_net_plan_query": (_i32, [_vp, _vp, _i32, _vp]),
    "osn_net_forward": (_i32, [_vp, _vp, _vp]),
    "osn_net_backward": (_i32, [_vp, _vp, _vp]),
    "osn_maps_build": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "osn_events_create": (_vp, [_i32]),
    "osn_events_destroy": (None, [_vp]),
    "osn_prof_create": (_vp, [_i32]),
    "osn_prof_destroy": (None, [_vp]),
    "osn_prof_filter": (_i32, [_vp, _vp, _i32]),
    "osn_prof_read": (_i32, [_vp, _vp, _vp, _i32, _i32]),
    "osn_relu_fwd": (_i32, [_vp, _vp, _i64, _vp]),
    "osn_relu_bwd": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "osn_add": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "osn_cat2": (_i32, [_vp, _i32, _vp, _i32, _vp, _i64, _vp]),
    "osn_cat2_bwd": (_i32, [_vp, _vp, _i32, _vp, _i32, _i64, _vp]),
    "osn_cosine_query": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "osn_query_ensemble_ws_bytes": (_sz, [_i64]),
    "osn_query_ensemble": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _sz, _vp]),
    "osn_voxelize_ws_bytes": (_sz, [_i64]),
    "osn_voxelize_fnv": (_i32, [_vp, _i64, _c.POINTER(_c.c_double), _vp, _vp, _vp, _c.POINTER(_i64), _vp, _sz, _vp]),
    "osn_fnv_hash": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "osn_ravel_hash": (_i32, [_vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "osn_feature_remap_ws_bytes": (_sz, [_i64, _i64]),
    "osn_feature_remap": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "osn_batch_coords": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "osn_fusion_project": (_i32, [_vp, _i64, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _vp, _i32, _i32, _i32,
                                  _c.c_double, _vp, _vp]),
    "osn_fusion_accumulate": (_i32, [_vp, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _vp]),
    "osn_fusion_finish": (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
}

_lock = threading.Lock()
_lib = None


class OpenSceneAmdError(RuntimeError):
    pass


def load():
    """dlopen the library and bind every prototype (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch ships its own libamdhip64; it must be the HIP runtime this process binds to, so make sure it is loaded
        # BEFORE dlopen resolves the library's dependency (loading ours first pulls /opt/rocm's copy: two runtimes in one
        # process, and every HIP call from here then sees no device)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise OpenSceneAmdError(
                "libopenscene_amd.so is not built (%s).  Build it with `python -m openscene_amd.build` "
                "(needs hipcc; cross-compiles for gfx950).  There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def last_error():
    return load().osn_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise OpenSceneAmdError("%s failed (%d): %s" % (what, rc, last_error()))


_device_checked = {}


def require_device(device):
    """Raise unless `device` is a HIP device of arch gfx950 (cached per device)."""
    import torch
    if device.type != "cuda":
        raise OpenSceneAmdError("openscene_amd ops need tensors on a HIP device (got %s); there is no CPU path" % device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ok = _device_checked.get(idx)
    if ok is None:
        with torch.cuda.device(idx):
            ok = bool(load().osn_device_ok())
        _device_checked[idx] = ok
    if not ok:
        raise OpenSceneAmdError("device cuda:%d is not gfx950 (MI355X); this library targets gfx950 only" % idx)
