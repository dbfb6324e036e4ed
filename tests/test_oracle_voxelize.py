"""Oracle vs golden vectors minted from the reference's own voxeliser code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import voxelize as ov


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_fnv_known_answers(golden_dir):
    g = _load(golden_dir, "hash_kat.npz")
    # values computed by importing the reference function
    assert [int(v) for v in g["fnv"][:3]] == [15658191375538532279, 15657232601398921515,
                                              15489006222804313940]
    assert np.array_equal(ov.fnv_keys(g["coords"]), g["fnv"])
    assert [int(v) for v in g["ravel"][:3]] == [int(v) for v in ov.ravel_keys(g["coords"])[:3]]
    assert np.array_equal(ov.ravel_keys(g["coords"]), g["ravel"])


@pytest.mark.parametrize("name", ["quantize_small.npz", "quantize_frac.npz"])
def test_quantize(golden_dir, name):
    g = _load(golden_dir, name)
    inds, inv = ov.quantize_first_occurrence(g["coords"])
    assert np.array_equal(inds, g["inds"])
    assert np.array_equal(inv, g["inverse"])


@pytest.mark.parametrize("name", ["voxelize_a.npz", "voxelize_b.npz"])
def test_voxelize_seeded(golden_dir, name):
    g = _load(golden_dir, name)
    np.random.seed(int(g["np_seed"]))
    coords, inds, inv, T = ov.voxelize(g["xyz"], float(g["voxel_size"]))
    assert np.array_equal(T, g["T"])                    # same RNG consumption, same matrix
    assert np.array_equal(coords, g["coords"])
    assert np.array_equal(inds, g["inds"])
    assert np.array_equal(inv, g["inverse"])
    # structural properties of np.unique semantics
    keys = ov.fnv_keys(coords)
    assert np.all(keys[1:] > keys[:-1])                 # ascending distinct keys
    assert np.array_equal(coords[inv], np.floor(coords[inv]))
