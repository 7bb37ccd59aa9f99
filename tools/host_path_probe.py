#!/usr/bin/env python3
"""Diagnostic: wall time of vrt_trace with HOST outputs (kernel + copy of the planes to host memory),
the path a host that presents frames itself would use.  1080p over 1024^3, BGRA8 only and with the
hit record; pageable numpy buffers and buffers pinned with torch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
from raytracing_amd import _lib as L

W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
cam = np.asarray(scenes.cam16(p), np.float32)
camp = cam.ctypes.data_as(C.POINTER(C.c_float))
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    g.kernel_time_sampling(0)
    for name, pinned in (("pageable", False), ("pinned", True)):
        bgra = torch.zeros((H, W, 4), dtype=torch.uint8, pin_memory=pinned)
        steps = torch.zeros((H, W), dtype=torch.int32, pin_memory=pinned)
        voxel = torch.zeros((H, W, 3), dtype=torch.int32, pin_memory=pinned)
        for label, planes in (("bgra8", dict(bgra8=bgra)), ("bgra8+steps+voxel", dict(bgra8=bgra, steps=steps, voxel=voxel))):
            names = ("bgra8", "voxel", "face", "steps", "t", "hit_point", "rgba")
            o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_HOST, *[planes[k].data_ptr() if k in planes else None for k in names])
            for _ in range(3):
                L.check(L.lib().vrt_trace(g._ctx, camp, W, H, 0, H, C.byref(o)))
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                L.check(L.lib().vrt_trace(g._ctx, camp, W, H, 0, H, C.byref(o)))
            dt = (time.perf_counter() - t0) / reps
            nbytes = sum(v.numel() * v.element_size() for v in planes.values())
            print(f"{name:9s} {label:18s}: {dt * 1e6:8.1f} us per frame, {nbytes / dt / 1e9:6.1f} GB/s of planes, {W * H / dt / 1e6:8.1f} Mrays/s")
    # the kernel storing straight into pinned host memory (device-visible under unified addressing)
    bgra = torch.zeros((H, W, 4), dtype=torch.uint8, pin_memory=True)
    ref = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
    g.trace_device(cam, W, H, 0, H, {"bgra8": ref.data_ptr()}); g.sync()
    for _ in range(3):
        g.trace_device(cam, W, H, 0, H, {"bgra8": bgra.data_ptr()}); g.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        g.trace_device(cam, W, H, 0, H, {"bgra8": bgra.data_ptr()}); g.sync()
    dt = (time.perf_counter() - t0) / 20
    print(f"direct stores into pinned memory, bgra8: {dt * 1e6:8.1f} us per frame, equal to the device frame: {bool(torch.equal(bgra, ref.cpu()))}")
