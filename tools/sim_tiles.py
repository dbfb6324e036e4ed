#!/usr/bin/env python
"""CPU study for the round-3 tile order question (no GPU): how do the tile-list statistics of the S100k level-0 3^3 map
change with the ROW ORDER?  For tiles of 88 consecutive rows it reports, per order,
  active (tile, offset) lists, 32-pair steps and 16-pair half-steps (= MFMA work incl. padding), weight-fragment loads,
  and the gather footprint: distinct input rows per tile vs pairs per tile (the reuse an L2-local tile could exploit).
Orders: `pattern` = the product's (rows sorted by offset-occupancy key, rarest offset most significant),
        `morton`  = z-order of the voxel coordinates, `morton+pattern` = pattern sort inside z-order blocks of B rows."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openscene_amd import synthetic as syn  # noqa: E402


def neighbour_table(xyz):
    """nbr[k, o] = row of the voxel at xyz[o] + offset k (x fastest), or -1: plain numpy (sorted keys + searchsorted)."""
    x = xyz.astype(np.int64) - xyz.min(0) + 1
    span = x.max(0) + 2
    key = (x[:, 2] * span[1] + x[:, 1]) * span[0] + x[:, 0]
    order = np.argsort(key)
    skey = key[order]
    nbr = np.full((27, x.shape[0]), -1, dtype=np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = ((x[:, 2] + dz) * span[1] + (x[:, 1] + dy)) * span[0] + (x[:, 0] + dx)
                pos = np.clip(np.searchsorted(skey, q), 0, skey.size - 1)
                hit = skey[pos] == q
                nbr[k, hit] = order[pos[hit]]
                k += 1
    return nbr


def morton(c):
    def part(v):
        v = v.astype(np.uint64) & 0x1FFFFF
        v = (v | (v << 32)) & 0x1F00000000FFFF
        v = (v | (v << 16)) & 0x1F0000FF0000FF
        v = (v | (v << 8)) & 0x100F00F00F00F00F
        v = (v | (v << 4)) & 0x10C30C30C30C30C3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    c = c - c.min(0)
    return part(c[:, 0]) | (part(c[:, 1]) << 1) | (part(c[:, 2]) << 2)


def pattern_key(nbr):
    occ = nbr >= 0                                   # [K, N]
    cnt = occ.sum(1)
    order = np.argsort(-cnt, kind="stable")          # most frequent offset -> least significant bit
    key = np.zeros(nbr.shape[1], dtype=np.uint64)
    for bit, k in enumerate(order):
        key |= occ[k].astype(np.uint64) << np.uint64(bit)
    return key


def stats(nbr, perm, bm=88):
    K, n = nbr.shape
    t = nbr[:, perm]
    n_tiles = (n + bm - 1) // bm
    lists = steps = halves = pairs = uniq = 0
    for ti in range(n_tiles):
        blk = t[:, ti * bm:(ti + 1) * bm]
        c = (blk >= 0).sum(1)
        a = c[c > 0]
        lists += a.size
        steps += int(((a + 31) // 32).sum())
        halves += int(((a + 15) // 16).sum())
        pairs += int(a.sum())
        uniq += np.unique(blk[blk >= 0]).size
    # reuse inside a WINDOW of 64 consecutive tiles = what one XCD's L2 would see if consecutive tiles ran on one XCD
    win = 64 * bm
    wpairs = wuniq = 0
    for w0 in range(0, n, win):
        blk = t[:, w0:w0 + win]
        v = blk[blk >= 0]
        wpairs += v.size
        wuniq += np.unique(v).size
    return {"window_reuse": wpairs / float(wuniq), "tiles": n_tiles, "lists_per_tile": lists / n_tiles, "steps32_per_tile": steps / n_tiles,
            "half_steps16_per_tile": halves / n_tiles, "pairs_per_tile": pairs / n_tiles,
            "slot_efficiency_16": pairs / (16.0 * halves), "distinct_rows_per_tile": uniq / n_tiles,
            "reuse_within_tile": pairs / float(uniq)}


def main():
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    coords = syn.batch_coords([vox])
    nbr = neighbour_table(coords[:, 1:4])
    n = nbr.shape[1]
    key = pattern_key(nbr)
    mz = morton(coords[:, 1:4].astype(np.int64))
    orders = {"hash (input) order": np.arange(n), "pattern": np.argsort(key, kind="stable"), "morton": np.argsort(mz, kind="stable")}
    orders["pattern, z-order inside a pattern"] = np.lexsort((mz, key))
    for B in (512, 2048, 8192):
        zperm = np.argsort(mz, kind="stable")
        blk = np.arange(n) // B
        orders["morton blocks of %d + pattern inside" % B] = zperm[np.lexsort((key[zperm], blk))]
    for name, perm in orders.items():
        s = stats(nbr, perm)
        print("%-38s lists %5.1f  steps32 %5.1f  half-steps16 %5.1f  slot-eff %.2f  distinct rows %6.1f  reuse %.2f  "
              "reuse in 64-tile window %.2f" % (
                  name, s["lists_per_tile"], s["steps32_per_tile"], s["half_steps16_per_tile"], s["slot_efficiency_16"],
                  s["distinct_rows_per_tile"], s["reuse_within_tile"], s["window_reuse"]))


def levels():
    """Per level of the S100k pyramid: the product's pattern order and tile height (osn_tile_rows) -> pairs per list etc."""
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0), 0.02), 0)
    c = vox.astype(np.int64)
    print("level rows  bm  tiles  lists/tile  pairs/list  half-steps16/tile  slot-eff  (pattern order, 3^3 map)")
    for lvl in range(5):
        s = 1 << lvl
        q = np.unique(np.floor_divide(c, s), axis=0)              # stride-s voxels in units of s
        nbr = neighbour_table(q)
        n = nbr.shape[1]
        bm = int(min(88, max(32, -(-(-(-n // 1024)) // 8) * 8)))
        st = stats(nbr, np.argsort(pattern_key(nbr), kind="stable"), bm)
        print("%5d %6d %3d %5d  %9.1f  %9.1f  %16.1f  %7.2f" % (lvl, n, bm, st["tiles"], st["lists_per_tile"],
              st["pairs_per_tile"] / st["lists_per_tile"], st["half_steps16_per_tile"], st["slot_efficiency_16"]))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "levels":
        levels()
    else:
        main()
