#!/usr/bin/env python3
"""Diagnostic: a few launches of the skip kernel from the bench position with a given pitch
(radians; +1.2 looks at the sky only, -1.4 at the ground only).  Run under
`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -- python3 tools/pose_probe.py <pitch>`
to read the instruction cost of the kernel's fixed part and of its loop separately."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, TRACE_SKIP, scenes
pitch = float(sys.argv[1]) if len(sys.argv) > 1 else scenes.POSE_B_PITCH
W, H, N = 1920, 1080, 1024
base = scenes.pose_headline()
p = Player.at(base.get_position(), scenes.POSE_B_YAW, pitch); cam = scenes.cam16(p)
g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda"); st = torch.zeros((H, W), dtype=torch.int32, device="cuda")
g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr(), "steps": st.data_ptr()}); g.sync()
print("pitch", pitch, "mean steps", float(st.float().mean()), "hit frac", float((frame[..., :3].sum(-1) > 0).float().mean()))
g.kernel_time_reset()
for _ in range(6): g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
ms, n = g.kernel_time(); print("kernel us", ms / n * 1e3)
