#!/usr/bin/env python3
"""Eight launches of the bench frame (1920x1080 over the 1024^3 terrain, headline pose), one at a time with rows in plain
order, through the library given on the command line: the workload of tools/stage_budget.sh."""
import os
import sys
os.environ["VRT_LIB"] = os.path.abspath(sys.argv[1])
os.environ["VRT_NO_ROW_ORDER"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracing_amd import GraphicsCore, TRACE_SKIP, VOXEL_PACKED_U8, scenes  # noqa: E402

W, H, N = 1920, 1080, 1024
pose = scenes.pose_headline()
cam = scenes.cam16(pose)
with GraphicsCore(width=W, height=H, grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(pose.get_position(), N))
    g.reset_distance_field()
    frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    for _ in range(8):
        g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
    g.sync()
print("done")
