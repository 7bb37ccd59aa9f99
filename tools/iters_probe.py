#!/usr/bin/env python3
"""Diagnostic: per-pixel loop iterations of the skip kernel on the bench workload.  Needs the
diagnostic library build, whose steps plane carries the kernel's own loop count:
    make -C raytracing_amd/csrc clean && make -C raytracing_amd/csrc EXTRA=-DVRT_DIAG_ITERS
(rebuild without EXTRA afterwards)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
if len(sys.argv) > 1:   # another pitch from the same position
    p = Player.at(p.get_position(), scenes.POSE_B_YAW, float(sys.argv[1]))
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    out = g.trace(scenes.cam16(p), W, H, outputs=("steps", "face"))
raw = out["steps"].astype(np.int64); it = raw & 0xffff; unit = raw >> 16; hit = out["face"] >= 0
print("unit-step iterations: mean", unit.mean(), "hit", unit[hit].mean(), "miss", unit[~hit].mean())
print("mean iters", it.mean(), "max", it.max(), "hit frac", hit.mean())
print("mean iters hit", it[hit].mean(), "miss", it[~hit].mean())
print("by row band (mean iters, hit frac):")
for r in range(0, H, 90):
    print(r, round(it[r:r+90].mean(), 1), round(hit[r:r+90].mean(), 2), "max", it[r:r+90].max())
# per-wave (8x8 tile) max vs mean -> divergence
t = it[:H//8*8, :W//8*8].reshape(H//8, 8, W//8, 8)
mx = t.max(axis=(1, 3)); mn = t.mean(axis=(1, 3))
print("wave-level: sum of max", mx.sum() * 64, "sum of lanes", it.sum(), "lane efficiency", it.sum() / (mx.sum() * 64))
print("histogram of iters:", np.histogram(it, bins=[0, 2, 4, 8, 16, 32, 64, 128, 256, 100000])[0])
worst = np.argwhere(it >= it.max() * 0.9)
print("pixels within 10 % of the max:", len(worst), "rows", worst[:, 0].min(), worst[:, 0].max(), "cols", worst[:, 1].min(), worst[:, 1].max())
print("reference steps there:", out["steps_ref"][it >= it.max() * 0.9][:8] if "steps_ref" in out else "")
