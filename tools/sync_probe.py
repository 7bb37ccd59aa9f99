#!/usr/bin/env python3
"""Where a short timed region's fixed cost goes: wall time of k launches + one wait against the GPU's own time for them
(HIP events), the cost of a wait on an idle stream, and of the torch-side synchronize.  usage: sync_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    g.kernel_time_sampling(0); g.set_frame_overlap(2)
    f = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    calls = [g.bound_trace_device(cam, W, H, 0, H, {"bgra8": t.data_ptr()}) for t in f]
    for i in range(50): calls[i & 1]()
    g.sync(); torch.cuda.synchronize()
    def med(fn, reps=30):
        v = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); v.append((time.perf_counter() - t0) * 1e6)
        return sorted(v)[len(v) // 2]
    print(f"vrt_sync on an idle context        {med(g.sync):8.1f} us")
    print(f"torch.cuda.synchronize when idle   {med(torch.cuda.synchronize):8.1f} us")
    for k in (1, 2, 5, 20, 100):
        def run():
            for i in range(k): calls[i & 1]()
            g.sync()
        t = med(run, 20)
        print(f"{k:4d} launches + vrt_sync: {t:8.1f} us wall, {t / k:7.2f} per launch, over 35.3 us each: {t - 35.3 * k:7.1f} us")
