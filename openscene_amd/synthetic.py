"""Seeded synthetic inputs shaped like the reference's datasets.

There is no network/dataset in the build or bench environment, so every
benchmark and parity test runs on these generators (SURVEY.md section 8(d) /
appendix A).  The scene format they emulate is the reference's preprocessed
``(coords float64 [N,3] in metres, colors, labels)`` tuple
(``scripts/preprocess/preprocess_3d_scannet.py:24-25``) and the batch layout
of ``dataset/feature_loader.py:178-205`` (int32 ``[N,4]`` rows ``(b,x,y,z)``).
"""
import numpy as np


def _rect_area(face):
    _, u, v = face
    return np.linalg.norm(np.cross(u, v))


def room_points(seed=0, n_pts=120_000, dims=(4.8, 3.6, 2.4), n_boxes=6, jitter=0.004):
    """ScanNet-shaped indoor scene: floor + 4 walls + boxes, surface samples.

    S100k = room_points(0) voxelised at 2 cm -> 100 999 voxels."""
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    A = lambda *t: np.asarray(t, dtype=np.float64)
    faces = [
        (A(0, 0, 0), A(X, 0, 0), A(0, Y, 0)),
        (A(0, 0, 0), A(X, 0, 0), A(0, 0, Z)), (A(0, Y, 0), A(X, 0, 0), A(0, 0, Z)),
        (A(0, 0, 0), A(0, Y, 0), A(0, 0, Z)), (A(X, 0, 0), A(0, Y, 0), A(0, 0, Z)),
    ]
    for _ in range(n_boxes):
        s = rng.uniform([0.4, 0.4, 0.4], [1.8, 1.0, 1.2])
        o = A(rng.uniform(0, X - s[0]), rng.uniform(0, Y - s[1]), 0.0)
        ex, ey, ez = A(s[0], 0, 0), A(0, s[1], 0), A(0, 0, s[2])
        faces += [(o + ez, ex, ey), (o, ex, ez), (o + ey, ex, ez), (o, ey, ez), (o + ex, ey, ez)]
    area = np.array([_rect_area(f) for f in faces])
    cnt = rng.multinomial(n_pts, area / area.sum())
    chunks = []
    for (o, u, v), c in zip(faces, cnt):
        ab = rng.random((c, 2))
        chunks.append(o + ab[:, :1] * u + ab[:, 1:] * v)
    pts = np.concatenate(chunks, 0)
    return pts + rng.normal(0, jitter, (n_pts, 3))


def lidar_points(seed=0, n_beams=32, n_az=1090, n_sweeps=10, max_range=50.0, sigma=0.02):
    """nuScenes-shaped outdoor sweep stack: ground plane + two walls."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-30.67, 10.67, n_beams))
    az = np.linspace(-np.pi, np.pi, n_az, endpoint=False)
    e, a = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)], -1).reshape(-1, 3)
    out = []
    for s in range(n_sweeps):
        origin = np.array([0.5 * s, 0.0, 1.84])
        t = np.full(d.shape[0], np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(d[:, 2] < 0, -origin[2] / d[:, 2], np.inf)
            t = np.minimum(t, tg)
            for yw in (12.0, -12.0):
                tw = (yw - origin[1]) / d[:, 1]
                tw = np.where(tw > 0, tw, np.inf)
                t = np.minimum(t, tw)
        keep = t < max_range
        out.append(origin + d[keep] * t[keep, None])
    pts = np.concatenate(out, 0)
    return pts + rng.normal(0, sigma, pts.shape)


def grid_voxels(points, voxel_size):
    """Plain (un-augmented) voxel grid of a cloud: unique int32 voxel rows [N,3],
    origin-aligned, in lexicographic order.  Only used to make bench/test inputs
    (the reference-parity voxeliser is openscene_amd.voxelizer.Voxelizer)."""
    g = np.floor(points / voxel_size).astype(np.int64)
    g -= g.min(0)
    return np.unique(g, axis=0).astype(np.int32)


def batch_coords(voxel_lists):
    """[(Ni,3) int] per scene -> int32 [sum Ni, 4] rows (batch, x, y, z)."""
    rows = []
    for b, v in enumerate(voxel_lists):
        v = np.asarray(v, dtype=np.int32)
        rows.append(np.concatenate([np.full((v.shape[0], 1), b, np.int32), v], 1))
    return np.concatenate(rows, 0)


def shuffled(voxels, seed=0):
    """Voxel rows in a random (hash-like) order -- the reference's rows arrive
    ordered by FNV key (``np.unique``), i.e. spatially random."""
    rng = np.random.default_rng(seed)
    return voxels[rng.permutation(voxels.shape[0])]
