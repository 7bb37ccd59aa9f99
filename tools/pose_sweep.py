#!/usr/bin/env python3
"""Diagnostic: kernel time of the skip kernel over many random poses (1080p over 1024^3 u8, the bench
world): any yaw, pitch from straight down to straight up, a third of the positions near or under the
surface, heights up to 400 voxels.  Prints the distribution and the slowest poses, to find frames that
cost far more than the bench frame.
usage: pose_sweep.py [poses, default 200] [seed] [random|axis]   (axis: yaw / pitch on or near multiples of pi/2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, TRACE_SKIP, scenes

poses = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
W, H, N = 1920, 1080, 1024
base = scenes.pose_headline()
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
steps = torch.zeros((H, W), dtype=torch.int32, device="cuda")
rows = []
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(base.get_position(), N)); g.reset_distance_field(); g.sync()
    chunk = np.array(scenes.chunk_of(base.get_position()), np.float32) * 32
    for i in range(poses):
        pos = (chunk + rng.uniform(0.01, 31.99, 3)).astype(np.float32)
        kind = i % 3
        if kind == 0: pos[1] = np.float32(rng.uniform(-8.0, 12.0))       # near / under the surface
        elif kind == 1: pos[1] = np.float32(rng.uniform(12.0, 80.0))
        else: pos[1] = np.float32(rng.uniform(80.0, 400.0))
        yaw, pitch = float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5))
        if len(sys.argv) > 3 and sys.argv[3] == "axis":
            eps = [0.0, 0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3]
            yaw = float(rng.integers(-2, 3)) * (np.pi / 2) + eps[rng.integers(0, 8)]
            pitch = float(rng.integers(-1, 2)) * (np.pi / 2) * (1.0 if rng.random() < 0.3 else 0.0) + eps[rng.integers(0, 8)]
            if i % 2 == 0: pos = np.floor(pos) + np.float32(rng.integers(0, 2)) * np.float32(0.5)
        cam = scenes.cam16(Player.at(pos, yaw, pitch))
        for _ in range(4):   # lets the dispatch order settle
            g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr(), "steps": steps.data_ptr()})
        g.sync(); g.kernel_time_reset()
        for _ in range(3):
            g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
        ms, n = g.kernel_time()
        rows.append((ms / n * 1e3, float(steps.float().mean()), pos.tolist(), yaw, pitch))
us = np.array([r[0] for r in rows])
print("poses", len(us), "kernel us: min %.1f  median %.1f  mean %.1f  p90 %.1f  p99 %.1f  max %.1f" %
      (us.min(), np.median(us), us.mean(), np.percentile(us, 90), np.percentile(us, 99), us.max()))
for r in sorted(rows, key=lambda r: -r[0])[:12]:
    print("  %.1f us  mean steps %.0f  pos (%.2f, %.2f, %.2f) yaw %.3f pitch %.3f" % (r[0], r[1], *r[2], r[3], r[4]))
