#!/usr/bin/env python3
"""Which rays of the bench frame make the most trips of the traversal loop (diagnostic build -DVRT_DIAG_ITERS through VRT_LIB):
rows and columns of the longest, their unit steps, reference iterations and whether they hit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    out = g.trace(scenes.cam16(p), W, H, outputs=("steps", "face", "voxel", "t"))
it = out["steps"].astype(np.int64)
trips, unit = it & 0xffff, it >> 16
print("trips histogram:", np.bincount(trips.ravel())[:30].tolist())
rows = trips.max(axis=1)
print("rows whose longest ray makes >= 12 trips:", np.nonzero(rows >= 12)[0].tolist()[:60])
order = np.argsort(trips.ravel())[::-1][:25]
for o in order:
    r, c = divmod(int(o), W)
    print(f"row {r} col {c}: trips {trips[r, c]} unit steps {unit[r, c]} face {out['face'][r, c]} voxel {out['voxel'][r, c].tolist()} t {out['t'][r, c]:.1f}")
