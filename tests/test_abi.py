"""The C-ABI library loads (no GPU needed) and exports exactly what include/openscene_amd.h declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "openscene_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(osn_[a-z0-9_]+)\s*\(", src))


def test_library_builds_loads_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from openscene_amd import _lib
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    assert names == set(_lib.PROTOTYPES), (names ^ set(_lib.PROTOTYPES))
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert lib.osn_version() == 2
    assert lib.osn_hash_capacity(1000) == 2048 and lib.osn_hash_capacity(0) == 1024
    assert lib.osn_bn_ws_bytes(10, 32) > 0 and lib.osn_coords_unique_ws_bytes(1000) > 0
    # big maps: room for the (K/8 - 1) partial buffers of the units mode; 1x1 convs need no scratch
    assert lib.osn_spconv_fwd_ws_bytes(100999, 27, 96, 96) >= 3 * 100999 * 96 * 4
    assert lib.osn_spconv_fwd_ws_bytes(100999, 1, 96, 768) == 0
    assert lib.osn_spconv_fwd_ws_bytes(700, 27, 256, 256) > 0            # deep level: split + reduce buffer
    assert lib.osn_weight_prep_x6_bytes(27, 96, 128, 0) == 3 * 27 * 128 * 96 * 2
    # second-generation kernels: tile rows between 32 and 64 (round 4: three workgroups per CU), fragment images of 1 KB blocks, pair-list buffers
    assert lib.osn_tile_rows(100999) == 64 and lib.osn_tile_rows(3052) == 32 and lib.osn_tile_rows(47618) == 32
    assert lib.osn_weight_prep_tl_bytes(27, 96, 96, 0) == 3 * 27 * 3 * 6 * 1024
    assert lib.osn_tile_lists_bytes(1000, 27, 32) > 27 * 1000 * 8 and lib.osn_pair_lists_bytes(1000, 27, 32) > 2 * 27 * 1000 * 4
    assert lib.osn_spconv_fwd_tl_ws_bytes(100999, 27, 96, 64) == 512                 # big table: no offset split
    assert lib.osn_spconv_fwd_tl_ws_bytes(700, 27, 256, 32) > 256                    # small table: partial tiles


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "openscene_amd.h")).read()
    for cite in ("run/evaluate.py:290-292", "dataset/voxelizer.py:117-129", "models/mink_unet.py", "run/distill.py:316-317"):
        assert cite in src
