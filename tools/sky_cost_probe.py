#!/usr/bin/env python3
"""A frame of nothing but sky (the camera looks up): every ray takes one trip, so the vector instructions of this launch are the
per-ray fixed cost of the primary kernel (ray generation, begin, one box, finish, shading, store).  Run under
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES; prints the trips it saw for reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
up = Player.at(p.get_position(), scenes.POSE_B_YAW, 1.2)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.kernel_time_sampling(0)
    fr = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    for cam in (scenes.cam16(up), scenes.cam16(p)):
        for _ in range(4):
            g.trace_device(cam, W, H, 0, H, {"bgra8": fr.data_ptr()})
        g.sync()
    face = g.trace(scenes.cam16(up), W, H, outputs=("face",))["face"]
    print("hits in the sky frame:", int((face >= 0).sum()))
