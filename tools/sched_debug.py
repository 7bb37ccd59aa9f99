#!/usr/bin/env python3
"""Diagnostic: with VRT_SCHED_DEBUG=1 in the environment the library prints, before every skip launch,
the start rows planned so far and the trip count per workgroup row the previous launch recorded."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
for i in range(8):
    g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
g.sync()
