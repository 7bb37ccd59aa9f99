#!/usr/bin/env python3
"""A model of the primary rays' box crossings over the bench world (CPU, numpy, double precision; not bit exact and
not meant to be): how many trips of the traversal loop a wavefront needs when the column-top boxes are ALIGNED blocks
of cell columns (what the kernel uses: 2^L x 2^L blocks of the max-mip pyramid) against squares ANCHORED at the ray's
own cell that extend in the ray's direction of travel (a sparse table of the same tops).

  python tools/box_model.py [--width 1920 --height 1080 --n 1024]

The terrain is treated as a height field (column tops of 4 x 4 voxel columns, as the kernel's pyramid has them); a ray
ends when its voxel is at or below its own column's top (the kernel's unit-step phase takes over there) or when it
leaves the loaded window.  Prints trips per ray and per 16 x 4 wave tile for each strategy.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from raytracing_amd import scenes  # noqa: E402


def column_tops(n, wmin):
    """Window level just above the highest solid voxel of every 4 x 4 block of voxel columns (0: none)."""
    import oracle_lib as O
    block = O.terrain_fill(n, wmin)                       # [z, y, x] at texel rem_euclid(p, n)
    shift = [-(int(w) % n) for w in wmin]
    block = np.roll(block, (shift[2], shift[1], shift[0]), axis=(0, 1, 2))   # index = p - wmin
    solid = block != 0
    lvl = np.arange(1, n + 1, dtype=np.int32)[None, :, None]
    top = (solid * lvl).max(axis=1).astype(np.int32)      # [z, x]
    nc = n // 4
    return top.reshape(nc, 4, nc, 4).max(axis=(1, 3))


def rays(cam, w, h):
    m = np.asarray(cam, np.float64).reshape(4, 4)  # m[c] = column c
    gx = np.arange(w, dtype=np.float64)[None, :]
    gy = (h - np.arange(h, dtype=np.float64))[:, None]
    ncx = gx / w - 0.5 + 0 * gy
    ncy = gy / h - (h / w) * 0.5 + 0 * gx
    v = np.stack([ncx, ncy, np.ones_like(ncx), np.ones_like(ncx)], -1)
    rd = np.stack([(v * m[i]).sum(-1) for i in range(3)], -1)
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = m[3, :3]
    return ro, rd.reshape(-1, 3)


def aligned_tables(tops0):
    t, out = tops0, [tops0]
    while t.shape[0] % 2 == 0 and t.shape[0] > 1 and len(out) < 9:
        t = t.reshape(t.shape[0] // 2, 2, t.shape[1] // 2, 2).max(axis=(1, 3))
        out.append(t)
    return out


def anchored_tables(tops0, sx, sz):
    """M[L][cz, cx] = max over the 2^L x 2^L square of cell columns with (cx, cz) at its corner, extending towards sx, sz."""
    out, t, L = [tops0], tops0, 0
    while (2 << L) <= tops0.shape[0] and len(out) < 9:
        h = 1 << L
        a = np.maximum(t, np.roll(t, -sx * h, axis=1))
        t = np.maximum(a, np.roll(a, -sz * h, axis=0))
        out.append(t)
        L += 1
    return out


def run(ro, rd, tops0, n, wmin, strategy, score_mul=4):
    nc = n // 4
    lo = np.asarray(wmin, np.float64) + 1.0
    hi = np.asarray(wmin, np.float64) + n - 1.0   # voxels wmin+1 .. wmin+n-2
    p = np.broadcast_to(ro, rd.shape).copy()
    trips = np.zeros(len(rd), np.int32)
    live = np.ones(len(rd), bool)
    sgn = np.where(rd > 0, 1, -1)
    tabs = {}
    if strategy == "aligned":
        al = aligned_tables(tops0)
    else:
        for sx in (-1, 1):
            for sz in (-1, 1):
                tabs[(sx, sz)] = anchored_tables(tops0, sx, sz)
    eps = 1e-6
    for it in range(400):
        idx = np.nonzero(live)[0]
        if len(idx) == 0:
            break
        q, d, s = p[idx], rd[idx], sgn[idx]
        vox = np.floor(q).astype(np.int64)
        inside = np.all((q >= lo) & (q < hi), axis=1)
        cx = np.clip((vox[:, 0] - wmin[0]) // 4, 0, nc - 1)
        cz = np.clip((vox[:, 2] - wmin[2]) // 4, 0, nc - 1)
        v = vox[:, 1] - wmin[1]
        ground = tops0[cz, cx] > v
        done = ~inside | ground
        trips[idx[ground & inside]] += 1        # the unit-step phase / the hit
        live[idx[done]] = False
        keep = ~done
        idx, q, d, s, cx, cz, v = idx[keep], q[keep], d[keep], s[keep], cx[keep], cz[keep], v[keep]
        if len(idx) == 0:
            continue
        bestL = np.zeros(len(idx), np.int64)
        bestT = tops0[cz, cx].astype(np.int64)
        score = np.minimum(4, score_mul * (v - bestT + 1))
        ok = np.ones(len(idx), bool)
        rising = d[:, 1] > 0
        nlev = len(al) if strategy == "aligned" else len(tabs[(1, 1)])
        for L in range(1, nlev):
            if strategy == "aligned":
                t = al[L][cz >> L, cx >> L]
            else:
                t = np.zeros(len(idx), np.int64)
                for key, tb in tabs.items():
                    msk = (s[:, 0] == key[0]) & (s[:, 2] == key[1])
                    t[msk] = tb[L][cz[msk], cx[msk]]
            ok &= t <= v
            sc = np.minimum(4 << L, score_mul * (v - t + 1))
            take = ok & (rising | (sc >= score))
            bestL = np.where(take, L, bestL)
            bestT = np.where(take, t, bestT)
            score = np.where(take, sc, score)
        side = 4 << bestL
        if strategy == "aligned":
            x0 = wmin[0] + ((cx >> bestL) << bestL) * 4
            z0 = wmin[2] + ((cz >> bestL) << bestL) * 4
        else:
            x0 = np.where(s[:, 0] > 0, wmin[0] + 4 * cx, wmin[0] + 4 * (cx + 1) - side)
            z0 = np.where(s[:, 2] > 0, wmin[2] + 4 * cz, wmin[2] + 4 * (cz + 1) - side)
        y0 = wmin[1] + bestT
        bx = np.where(d[:, 0] > 0, x0 + side, x0).astype(np.float64)
        bz = np.where(d[:, 2] > 0, z0 + side, z0).astype(np.float64)
        by = np.where(d[:, 1] > 0, hi[1] + 1.0, y0).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            tx = (bx - q[:, 0]) / d[:, 0]
            ty = (by - q[:, 1]) / d[:, 1]
            tz = (bz - q[:, 2]) / d[:, 2]
        t = np.minimum(np.minimum(tx, ty), tz)
        t = np.maximum(t, 0.0) + eps
        p[idx] = q + d * t[:, None]
        trips[idx] += 1
    return trips


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--n", type=int, default=1024)
    args = ap.parse_args()
    pose = scenes.pose_headline()
    cam = scenes.cam16(pose)
    wmin = scenes.window_min(pose.get_position(), args.n)
    tops0 = column_tops(args.n, wmin)
    ro, rd = rays(cam, args.width, args.height)
    for strategy in ("aligned", "anchored"):
        for mul in (4, 2, 8):
            tr = run(ro, rd, tops0, args.n, wmin, strategy, mul)
            img = tr.reshape(args.height, args.width)
            hh, ww = args.height // 4 * 4, args.width // 16 * 16
            waves = img[:hh, :ww].reshape(hh // 4, 4, ww // 16, 16).max(axis=(1, 3))
            print(f"{strategy:9s} score_mul {mul}: trips per ray {tr.mean():.3f} (max {tr.max()}), per wave tile {waves.mean():.3f}, "
                  f"lanes live per wave trip {tr.sum() / (waves.sum() * 64):.3f}", flush=True)


if __name__ == "__main__":
    main()
