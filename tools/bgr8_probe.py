#!/usr/bin/env python3
"""Cost of the three-byte plane: the bench frame traced into bgra8, into bgr8, and into bgra8 followed by vrt_pack_bgr --
HIP-event time of a launch alone and wall time per frame with two frames in flight.  usage: bgr8_probe.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
out = {}
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    f4 = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    f3 = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
    kinds = {"bgra8": lambda i: g.trace_device(cam, W, H, 0, H, {"bgra8": f4[i].data_ptr()}),
             "bgr8": lambda i: g.trace_device(cam, W, H, 0, H, {"bgr8": f3[i].data_ptr()}),
             "bgra8 + pack": lambda i: (g.trace_device(cam, W, H, 0, H, {"bgra8": f4[i].data_ptr()}), g.pack_bgr(f4[i].data_ptr(), f3[i].data_ptr(), W * H))}
    for rnd in range(2):
        for name, fn in kinds.items():
            g.set_frame_overlap(1); g.kernel_time_sampling(1)
            for _ in range(6): fn(0)
            g.sync(); g.kernel_time_reset()
            for _ in range(32): fn(0)
            ms, n = g.kernel_time()
            g.kernel_time_sampling(0); g.set_frame_overlap(2)
            for i in range(20): fn(i & 1)
            g.sync()
            t0 = time.perf_counter()
            for i in range(400): fn(i & 1)
            g.sync()
            out[name] = {"launch_alone_us": ms / n * 1e3, "two_in_flight_us": (time.perf_counter() - t0) / 400 * 1e6}
print(json.dumps(out, indent=1))
