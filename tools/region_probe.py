#!/usr/bin/env python3
"""Where the fixed cost of a short timed region goes: regions of K launches of the bench frame (two streams of one context) timed
by the host clock between two waits, as bench.py does, and by two HIP events on the context's first stream around the same
launches; the launches' own span comes from a kernel trace of this script.  usage: region_probe.py [K, default 20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
cam = scenes.cam16(p)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    stream = torch.cuda.ExternalStream(g.stream(), device=torch.device("cuda", 0))
    torch.cuda.set_stream(stream)
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    bound = [g.bound_trace_device(cam, W, H, 0, H, {"bgra8": f.data_ptr()}) for f in frames]
    g.set_frame_overlap(2)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        for i in range(16): bound[i & 1]()
        g.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode in ("plain", "events"):
        host, dev, first = [], [], []
        for rep in range(9):
            g.sync(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "events": e0.record(stream)
            bound[0]()
            t1 = time.perf_counter()
            for i in range(1, K): bound[i & 1]()
            if mode == "events":
                g.join(); e1.record(stream)
            g.sync(); torch.cuda.synchronize()
            host.append((time.perf_counter() - t0) * 1e6); first.append((t1 - t0) * 1e6)
            if mode == "events": dev.append(e0.elapsed_time(e1) * 1e3)
        print(f"{mode}: K={K} host region us median {np.median(host):.1f} (min {min(host):.1f}); first launch call {np.median(first):.1f} us"
              + (f"; between the events {np.median(dev):.1f} us" if dev else ""), flush=True)
