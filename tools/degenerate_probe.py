#!/usr/bin/env python3
"""Diagnostic: kernel time of frames that hold rays with a zero direction component.
  a  the reference's start pose (identity transform at the origin), 640x480 over 160^3 u16:
     the middle column and row gy = 180 are degenerate and spin until the step cap
  b  the bench position with the start orientation, 1080p over 1024^3 u8: the middle column
  c  the bench position looking up (pitch 1.2): one such pixel in the frame
  d  the bench frame (none)
Set VRT_LIB to time another build of the library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, VOXEL_REF_U16U16, TRACE_SKIP, scenes


def timed(g, cam, w, h, reps=8):
    frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    face = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    g.trace_device(cam, w, h, 0, h, {"bgra8": frame.data_ptr(), "face": face.data_ptr()})
    g.sync()
    capped = int((face == -2).sum())
    g.kernel_time_reset()
    for _ in range(reps):
        g.trace_device(cam, w, h, 0, h, {"bgra8": frame.data_ptr()})
    ms, n = g.kernel_time()
    return ms / n * 1e3, capped


p = scenes.pose_spawn()
with GraphicsCore(grid_n=160, voxel_format=VOXEL_REF_U16U16, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), 160)); g.reset_distance_field(); g.sync()
    print("a spawn pose 640x480 / 160^3: %.1f us, %d capped rays" % timed(g, scenes.cam16(p), 640, 480))
    print("  pose B     640x480 / 160^3: %.1f us, %d capped rays" % timed(g, scenes.cam16(scenes.pose_b()), 640, 480))
base = scenes.pose_headline()
with GraphicsCore(grid_n=1024, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(base.get_position(), 1024)); g.reset_distance_field(); g.sync()
    for name, yaw, pitch in [("b start orientation", 0.0, 0.0), ("c pitch 1.2", scenes.POSE_B_YAW, 1.2),
                             ("d bench frame", scenes.POSE_B_YAW, scenes.POSE_B_PITCH)]:
        cam = scenes.cam16(Player.at(base.get_position(), yaw, pitch))
        print("%s 1080p / 1024^3: %.1f us, %d capped rays" % ((name,) + timed(g, cam, 1920, 1080)))
