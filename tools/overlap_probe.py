#!/usr/bin/env python3
"""Do consecutive frames gain from overlapping (the tail of one frame's launch under the head of the next)?  Frames of
the bench workload issued to ONE context (one stream: each launch waits for the one before) against the same frames dealt
alternately to k contexts on the same GPU (k streams).  Wall time per frame over 600 frames, best of 3.
usage: overlap_probe.py [k ...]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
ks = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
cores = []
for _ in range(max(ks)):
    g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync(); g.kernel_time_sampling(0)
    cores.append(g)
frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2 * max(ks))]
out = {}
for k in ks:
    best = 1e9
    for rep in range(3):
        for i in range(40): cores[i % k].trace_device(cam, W, H, 0, H, {"bgra8": frames[i % (2 * k)].data_ptr()})
        for g in cores[:k]: g.sync()
        t0 = time.perf_counter()
        for i in range(600): cores[i % k].trace_device(cam, W, H, 0, H, {"bgra8": frames[i % (2 * k)].data_ptr()})
        for g in cores[:k]: g.sync()
        best = min(best, (time.perf_counter() - t0) / 600)
    out[f"{k}_streams_us_per_frame"] = round(best * 1e6, 2)
print(json.dumps(out))
