#!/usr/bin/env python3
"""Long parity run (not part of pytest): many random poses over the procedural terrain, full frames,
every output plane of the HIP skip kernel against the CPU oracle, bit for bit, and the planes of a second launch that
asks for the frame without the voxel and steps planes (the launch a host makes; wavefronts of sky take a short cut there).
usage: soak_parity.py [poses per world, default 40] [seed] [default|odd|sparse|axis]   -> one summary line per world + total
(axis: the default worlds seen along the axes -- yaw and pitch within 1e-3 of multiples of pi/2, often exactly on them, and
positions with integer coordinates: direction components that are zero, or 1e-8, and plane parameters up to 1e11)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from raytracing_amd import GraphicsCore, Player, TRACE_SKIP, VOXEL_PACKED_U8, VOXEL_REF_U16U16, scenes

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
PLANES = ("bgra8", "voxel", "face", "steps", "t", "hit_point", "rgba")
total_px = total_bad = 0
t_start = time.time()
WORLDS = {"default": [(160, VOXEL_REF_U16U16, np.uint16, 640, 480), (256, VOXEL_PACKED_U8, np.uint8, 960, 540),
                      (1024, VOXEL_PACKED_U8, np.uint8, 1920, 1080)],
          # grids that are not powers of two (texel wrap by subtraction, fewer pyramid levels)
          "odd": [(96, VOXEL_PACKED_U8, np.uint8, 640, 360), (480, VOXEL_REF_U16U16, np.uint16, 960, 540),
                  (736, VOXEL_PACKED_U8, np.uint8, 1280, 720)]}
kind = sys.argv[3] if len(sys.argv) > 3 else "default"
WORLDS["sparse"] = [(64, VOXEL_PACKED_U8, np.uint8, 640, 360), (160, VOXEL_REF_U16U16, np.uint16, 640, 480),
                    (256, VOXEL_PACKED_U8, np.uint8, 960, 540)]
WORLDS["axis"] = WORLDS["default"]
for n, fmt, dtype, w, h in WORLDS[kind]:
    centre = np.array([rng.uniform(-200, 200), 20.0, rng.uniform(-200, 200)], np.float32)
    wmin = scenes.window_min(centre, n)
    with GraphicsCore(grid_n=n, voxel_format=fmt, trace_mode=TRACE_SKIP) as g:
        if kind == "sparse":
            # not a height field: random voxels of two densities, floating slabs, a hollow shell
            blk = (rng.random((n, n, n)) < (0.0005 if n > 128 else 0.01)).astype(dtype)
            blk[:, n // 3, : n // 2] = 1
            blk[n // 4: n // 2, 2 * n // 3, :] = 1
            blk[n // 8: n // 4, :, n // 8] = 1
            g.upload_world(blk)
        else:
            g.generate_window(wmin)
        world = O.World(g.download_world())
        base_chunk = scenes.chunk_of(centre)
        bad = px = 0
        for i in range(poses):
            # any position inside the chunk the window was built for, any yaw, pitch from straight down to up
            pos = (np.array(base_chunk, np.float32) * 32 + rng.uniform(0.01, 31.99, 3)).astype(np.float32)
            if i % 3 == 0: pos[1] = np.float32(base_chunk[1] * 32 + rng.uniform(0.01, 12.0))   # near / under the surface
            yaw, pitch = float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5))
            if kind == "axis":
                eps = [0.0, 0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3]
                yaw = float(rng.integers(-2, 3)) * (np.pi / 2) + eps[rng.integers(0, 8)]
                pitch = float(rng.integers(-1, 2)) * (np.pi / 2) * (1.0 if rng.random() < 0.3 else 0.0) + eps[rng.integers(0, 8)]
                if i % 2 == 0: pos = np.floor(pos) + np.float32(rng.integers(0, 2)) * np.float32(0.5)   # integer / half-integer coordinates
            p = Player.at(pos, yaw, pitch)
            cam = scenes.cam16(p)
            got = g.trace(cam, w, h)
            want = O.trace(world, cam, w, h)
            for k in PLANES:
                a, b = got[k], want[k]
                if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
                d = int((a != b).sum())
                if d:
                    bad += d
                    print(f"MISMATCH n={n} pose {i} plane {k}: {d} elements, pos={pos.tolist()} look={p.look_direction}", flush=True)
            # the frame alone: a launch that writes neither the voxel nor the steps plane lets wavefronts whose rays can
            # only meet air skip the traversal (the pixel needs no more than "1000 iterations or not")
            alone = g.trace(cam, w, h, outputs=("bgra8", "rgba", "face", "t"))
            for k in alone:
                a, b = alone[k], want[k]
                if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
                d = int((a != b).sum())
                if d:
                    bad += d
                    print(f"MISMATCH n={n} pose {i} plane {k} of the frame-only launch: {d} elements, pos={pos.tolist()} look={p.look_direction}", flush=True)
            px += w * h
            if i % 10 == 9: print(f"  n={n}: {i + 1} poses, {bad} mismatching elements, {time.time() - t_start:.0f} s", flush=True)
        print(f"world n={n} {w}x{h}: {poses} poses, {px} pixels, {len(PLANES)} planes each, mismatching elements: {bad}", flush=True)
        total_px += px; total_bad += bad
print(f"TOTAL: {total_px} pixels compared on {len(PLANES)} planes, mismatching elements: {total_bad}")
sys.exit(1 if total_bad else 0)
