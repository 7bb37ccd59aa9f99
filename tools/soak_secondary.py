#!/usr/bin/env python3
"""Long parity run of the secondary-ray kernels (not part of pytest): random poses over procedural terrain, worlds
that are not height fields and grids that are not powers of two; the shadow, ao and sec_steps planes of both kernels
(rays binned in LDS, and on their pixel's lane) against the CPU oracle, bit for bit.
usage: soak_secondary.py [poses per world, default 12] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from raytracing_amd import GraphicsCore, Player, TRACE_SKIP, VOXEL_PACKED_U8, VOXEL_REF_U16U16, scenes

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WORLDS = [("terrain", 256, VOXEL_PACKED_U8, np.uint8, 480, 270), ("terrain", 160, VOXEL_REF_U16U16, np.uint16, 320, 240),
          ("terrain", 96, VOXEL_PACKED_U8, np.uint8, 320, 180), ("sparse", 128, VOXEL_PACKED_U8, np.uint8, 320, 180),
          ("terrain", 480, VOXEL_PACKED_U8, np.uint8, 640, 360)]
total_px = total_bad = 0
t0 = time.time()
for kind, n, fmt, dtype, w, h in WORLDS:
    centre = np.array([rng.uniform(-200, 200), 20.0, rng.uniform(-200, 200)], np.float32)
    cores = {}
    for v in ("binned", "pixel"):
        os.environ["VRT_SECONDARY"] = v
        cores[v] = GraphicsCore(grid_n=n, voxel_format=fmt, trace_mode=TRACE_SKIP)
    g = cores["binned"]
    if kind == "sparse":
        blk = (rng.random((n, n, n)) < 0.004).astype(dtype)
        blk[:, n // 3, : n // 2] = 1; blk[n // 4: n // 2, 2 * n // 3, :] = 1; blk[n // 8: n // 4, :, n // 8] = 1
        g.upload_world(blk)
    else:
        g.generate_window(scenes.window_min(centre, n))
    block = g.download_world()
    cores["pixel"].upload_world(block)
    world = O.World(block)
    base = scenes.chunk_of(centre)
    bad = px = 0
    for i in range(poses):
        pos = (np.array(base, np.float32) * 32 + rng.uniform(0.01, 31.99, 3)).astype(np.float32)
        if i % 3 == 0: pos[1] = np.float32(base[1] * 32 + rng.uniform(0.01, 12.0))
        cam = scenes.cam16(Player.at(pos, float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.2))))
        want = O.trace_secondary(world, cam, w, h)
        for v, c in cores.items():
            got = c.trace_secondary(cam, w, h)
            for k in ("shadow", "ao", "sec_steps"):
                bad += int((got[k] != want[k]).sum()); px += want[k].size
    for c in cores.values(): c.close()
    print(f"{kind} n={n} {w}x{h}: {poses} poses x 2 kernels, {px} plane elements, {bad} mismatches", flush=True)
    total_px += px; total_bad += bad
print(f"TOTAL {total_px} plane elements compared, {total_bad} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if total_bad else 0)
