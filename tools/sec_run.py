#!/usr/bin/env python3
"""The secondary-ray workload (1080p primary + 5 secondary rays per hit over 1024^3 u8) with one setting of VRT_SECONDARY,
for profiler runs: usage: sec_run.py pixel|binned [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VRT_SECONDARY"] = sys.argv[1] if len(sys.argv) > 1 else "binned"
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
cam = scenes.cam16(p)
g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
g.kernel_time_sampling(0)
sh = torch.zeros((H, W), dtype=torch.uint8, device="cuda"); ao = torch.zeros_like(sh)
st = torch.zeros((H, W), dtype=torch.int32, device="cuda")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    g.trace_secondary_device(cam, W, H, 0, H, sh.data_ptr(), ao.data_ptr(), st.data_ptr())
g.sync()
g.close()
