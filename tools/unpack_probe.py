#!/usr/bin/env python3
"""Time of vrt_unpack_bgra and vrt_pack_bgr on a batch of eight 1080p frames (GPU time per frame, microseconds), one library
per run (VRT_LIB).  usage: unpack_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8
W, H, G = 1920, 1080, 8
with GraphicsCore(grid_n=64, voxel_format=VOXEL_PACKED_U8) as g:
    a = torch.randint(0, 256, (G, H, W, 3), dtype=torch.uint8, device="cuda"); b = torch.zeros((G, H, W, 4), dtype=torch.uint8, device="cuda")
    c = torch.zeros_like(a)
    for name, fn in (("unpack", lambda: g.unpack_bgra(a.data_ptr(), b.data_ptr(), G * H * W)), ("pack", lambda: g.pack_bgr(b.data_ptr(), c.data_ptr(), G * H * W))):
        for _ in range(5): fn()
        g.sync(); t0 = time.perf_counter()
        for _ in range(100): fn()
        g.sync(); dt = (time.perf_counter() - t0) / 100 / G * 1e6
        print(f"{name}: {dt:.2f} us per frame, {7 * W * H / dt / 1e6:.2f} TB/s")
    assert torch.equal(c, a) and int(b[..., 3].min()) == 255
