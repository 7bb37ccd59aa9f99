#!/usr/bin/env python3
"""Diagnostic: kernel time of row bands of the bench frame (HIP events), to see where
the frame time goes and how well bands overlap when traced together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
def t(y0, y1, reps=30):
    for _ in range(3): g.trace_device(cam, W, H, y0, y1, {"bgra8": frame.data_ptr()})
    g.kernel_time_reset()
    for _ in range(reps): g.trace_device(cam, W, H, y0, y1, {"bgra8": frame.data_ptr()})
    ms, n = g.kernel_time(); return ms / n * 1e3
bands = [(0, 360), (360, 450), (450, 540), (540, 630), (630, 810), (810, 1080), (448, 464), (464, 480), (432, 496), (432, 688), (0, 1080)]
tot = 0
for b in bands:
    us = t(*b); print(b, "%.1f us" % us, "%.0f Mrays/s" % ((b[1]-b[0])*W/us))
    if b in bands[:6]: tot += us
print("sum of bands %.1f us" % tot)
