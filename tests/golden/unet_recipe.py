"""Seeded inputs and parameters of the network golden vector (tests/golden/unet_dense_ref.npz) -- shared by the script
that mints it (make_golden_unet.py, authoring container only) and by the tests that replay it.  numpy only: the same
bytes on every machine, whatever torch's generators do."""
import zlib

import numpy as np

ARCH, IN_CH, OUT_CH, SEED = "MinkUNet18A", 3, 16, 20260924
EXTENT = 48          # 3 x 16: three cells per axis at tensor stride 16
# second fixture (unet_dense_ref_34c.npz): the nuScenes configuration's MinkUNet34C (config/nuscenes/ours_openseg.yaml, BASELINE configs[4]) on
# the same cloud and parameter recipe; the stored output rows are every ROW_STEP-th (file size)
ARCH_B, ROW_STEP_B = "MinkUNet34C", 4


def _rng(*key):
    return np.random.default_rng([SEED] + [zlib.crc32(str(k).encode()) for k in key])


def cloud():
    """Two scenes of a bent sheet + a box of clutter on a 48^3 grid: (coords int32 [N, 4] (b, x, y, z), unique, shuffled;
    feats float64 [N, 3])."""
    out = []
    for b in range(2):
        r = _rng("cloud", b)
        u, v = np.meshgrid(np.arange(EXTENT), np.arange(EXTENT), indexing="ij")
        h = (10 + 6 * b + 5 * np.sin(u / 7.0 + b) + 4 * np.cos(v / 9.0)).astype(np.int64)
        sheet = np.stack([u.ravel(), v.ravel(), h.ravel()], 1)
        sheet = sheet[r.random(sheet.shape[0]) < 0.8]
        wall = np.stack([np.full(EXTENT * 20, 5 + 30 * b), np.repeat(np.arange(EXTENT), 20), np.tile(np.arange(20, 40), EXTENT)], 1)
        wall = wall[r.random(wall.shape[0]) < 0.6]
        clutter = r.integers(0, EXTENT, (300, 3))
        c = np.unique(np.clip(np.concatenate([sheet, wall, clutter]), 0, EXTENT - 1), axis=0)
        c = c[r.permutation(c.shape[0])]
        out.append(np.concatenate([np.full((c.shape[0], 1), b), c], 1))
    coords = np.concatenate(out).astype(np.int32)
    feats = _rng("feats").random((coords.shape[0], 3))
    return coords, feats


def parameter(name, shape):
    """float64 value of state-dict entry `name` (models/mink_unet.py naming; None for integer buffers)."""
    r = _rng("param", name)
    if name.endswith("num_batches_tracked"):
        return None
    if name.endswith(".kernel"):
        fan = (shape[0] if len(shape) == 3 else 1) * shape[-1]                 # offsets x output channels (fan_out)
        return r.standard_normal(shape) * np.sqrt(2.0 / fan)
    if name.endswith("running_mean"):
        return r.uniform(-0.2, 0.2, shape)
    if name.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape)
    if name.endswith(".bn.weight"):
        return r.uniform(0.5, 1.5, shape)
    if name.endswith(".bn.bias"):
        return r.uniform(-0.3, 0.3, shape)
    raise KeyError(name)


def probe(name, shape):
    """The direction a parameter's gradient is projected on (one scalar per parameter in the fixture)."""
    return _rng("probe", name).standard_normal(shape)


def output_weights(n):
    """loss = sum(out * output_weights): the scalar whose gradients the fixture holds."""
    return _rng("gout").standard_normal((n, OUT_CH))
