#!/usr/bin/env python3
"""What a multi-GPU run of bench.py would trace, measured on ONE GPU: the frame of every rank of a job of 2, 4 and 8 ranks (rank r
looks from the bench position with its yaw turned by 2*pi*r/N, as bench.py's rank_pose does) -- HIP-event time of a launch alone
and wall time per frame with two frames in flight -- and the cost of the pack / unpack kernels around the gather.  The slowest
rank's figures are what the job's step time is made of.  Prints one JSON object.
usage: scale_inputs.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, TRACE_SKIP, scenes

W, H, N = 1920, 1080, 1024
base = scenes.pose_headline()
out = {"frame": [W, H], "grid_n": N, "jobs": {}}
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(base.get_position(), N)); g.reset_distance_field(); g.sync()
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    packed = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    seen = {}
    for world in (1, 2, 4, 8):
        rows = []
        for r in range(world):
            key = round(r / world, 6)
            if key not in seen:
                p = base if world == 1 else Player.at(base.get_position(), scenes.POSE_B_YAW + 2.0 * np.pi * r / world, scenes.POSE_B_PITCH)
                cam = scenes.cam16(p)
                g.set_frame_overlap(1); g.kernel_time_sampling(1)
                for _ in range(6): g.trace_device(cam, W, H, 0, H, {"bgra8": frames[0].data_ptr()})
                g.sync(); g.kernel_time_reset()
                for _ in range(32): g.trace_device(cam, W, H, 0, H, {"bgra8": frames[0].data_ptr()})
                ms, n = g.kernel_time()
                alone = ms / n
                g.kernel_time_sampling(0); g.set_frame_overlap(2)
                for i in range(20): g.trace_device(cam, W, H, 0, H, {"bgra8": frames[i & 1].data_ptr()})
                g.sync()
                t0 = time.perf_counter()
                for i in range(400): g.trace_device(cam, W, H, 0, H, {"bgra8": frames[i & 1].data_ptr()})
                g.sync()
                seen[key] = {"turn": key, "launch_alone_us": alone * 1e3, "two_in_flight_us": (time.perf_counter() - t0) / 400 * 1e6}
            rows.append(seen[key])
        out["jobs"][str(world)] = {"ranks": rows, "slowest_alone_us": max(x["launch_alone_us"] for x in rows),
                                   "slowest_two_in_flight_us": max(x["two_in_flight_us"] for x in rows)}
    g.set_frame_overlap(1)
    for name, fn in (("pack_bgr_us", lambda: g.pack_bgr(frames[0].data_ptr(), packed.data_ptr(), W * H)),
                     ("unpack_bgra_us", lambda: g.unpack_bgra(packed.data_ptr(), frames[1].data_ptr(), W * H))):
        for _ in range(5): fn()
        g.sync()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        g.sync()
        out[name] = (time.perf_counter() - t0) / 200 * 1e6
print(json.dumps(out, indent=1))
