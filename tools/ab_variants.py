#!/usr/bin/env python3
"""A/B of skip-kernel settings in ONE process on the bench workload: interleaved rounds,
HIP-event kernel time per launch (median and min per setting); every setting must return the
frame of the first, bit for bit.  The setting is the value of the environment variable named by
AB_VAR, which the library reads when a context is created:
  VRT_NO_ROW_ORDER  1 = workgroup rows top to bottom, 0 = measured dispatch order (default)
  VRT_ROW_BLOCK0    fixed start row block (-1 = measured)
  VRT_TILE_LOG2H    wave tile height 2^k rows
usage: AB_VAR=VRT_NO_ROW_ORDER ab_variants.py 1,0 [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, TRACE_LITERAL, scenes

variants = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
var = os.environ.get("AB_VAR", "VRT_NO_ROW_ORDER")
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
cam = scenes.cam16(p)
cores = {}
for v in variants:  # one context per setting: the library reads its knobs when a context is created
    os.environ[var] = v
    g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    cores[v] = g
frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
ref = None
times = {v: [] for v in variants}
for v in variants:  # correctness of each setting against the first
    out = cores[v].trace(cam, W, H, outputs=("bgra8", "steps", "voxel", "face", "t"))
    if ref is None:
        ref = out
    else:
        for k in ref:
            a, b = ref[k], out[k]
            if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
            assert np.array_equal(a, b), f"variant {v} differs from {variants[0]} on {k}: {(a != b).sum()} elements"
for r in range(rounds):
    for v in variants:
        g = cores[v]
        for _ in range(4): g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
        g.kernel_time_reset()
        for _ in range(20):
            g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
        ms, n = g.kernel_time()
        times[v].append(ms / n)
for v in variants:
    t = np.array(times[v][1:])
    print(f"variant {v}: median {np.median(t)*1e3:.1f} us  min {t.min()*1e3:.1f} us  -> {W*H/np.median(t)/1e3:.0f} Mrays/s")
