#!/usr/bin/env python3
"""Diagnostic: how many box-skip iterations use a distance of 1, 2, 3-4, >=5 cells."""
import os, sys
os.environ["VRT_DEBUG_ITERS"] = "2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    out = g.trace(scenes.cam16(p), W, H, outputs=("steps", "face"))
raw = out["steps"].astype(np.int64); hit = out["face"] >= 0
f = [(raw >> s) & 0xff for s in (0, 8, 16, 24)]
for name, m in (("hit", hit), ("miss", ~hit)):
    print(name, "mean skip iterations with d=1: %.2f  d=2: %.2f  d=3-4: %.2f  d>=5: %.2f" % tuple(x[m].mean() for x in f))
for r in range(0, H, 90):
    print(r, ["%.1f" % x[r:r+90].mean() for x in f])
