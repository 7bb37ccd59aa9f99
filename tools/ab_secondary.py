#!/usr/bin/env python3
"""A/B of the two secondary-ray kernels on the secondary-ray workload (1920x1080 primary rays + one shadow and four
ambient-occlusion rays per hit over the 1024^3 u8 terrain) in ONE process: one context per setting of
VRT_SECONDARY (pixel: every pixel's rays on its own lane; binned: the workgroup's rays staged in LDS and sorted
by direction class), identical outputs required, interleaved rounds of HIP-event kernel times.
usage: ab_secondary.py [rounds]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
cam = scenes.cam16(p)
cores = {}
for v in ("pixel", "binned"):
    os.environ["VRT_SECONDARY"] = v
    g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    cores[v] = g
ref = cores["pixel"].trace_secondary(cam, W, H)
got = cores["binned"].trace_secondary(cam, W, H)
for k in ref:
    assert np.array_equal(ref[k], got[k]), f"{k}: {(ref[k] != got[k]).sum()} pixels differ"
hits = int(cores["pixel"].trace(cam, W, H, outputs=("face",))["face"].__ge__(0).sum())
sh = torch.zeros((H, W), dtype=torch.uint8, device="cuda"); ao = torch.zeros_like(sh)
st = torch.zeros((H, W), dtype=torch.int32, device="cuda")
fr = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
times = {v: [] for v in cores}
prim = []
for r in range(rounds):
    for v, g in cores.items():
        for _ in range(3): g.trace_secondary_device(cam, W, H, 0, H, sh.data_ptr(), ao.data_ptr(), st.data_ptr())
        g.sync(); g.kernel_time_reset()
        for _ in range(10): g.trace_secondary_device(cam, W, H, 0, H, sh.data_ptr(), ao.data_ptr(), st.data_ptr())
        ms, n = g.kernel_time(); times[v].append(ms / n)
    g = cores["binned"]
    for _ in range(3): g.trace_device(cam, W, H, 0, H, {"bgra8": fr.data_ptr()})
    g.sync(); g.kernel_time_reset()
    for _ in range(10): g.trace_device(cam, W, H, 0, H, {"bgra8": fr.data_ptr()})
    ms, n = g.kernel_time(); prim.append(ms / n)
pm = float(np.median(prim))
out = {"primary_ms": pm, "secondary_rays": 5 * hits, "mean_secondary_steps": float(ref["sec_steps"].astype(np.int64).sum()) / (5 * hits)}
for v in cores:
    t = float(np.median(times[v][1:] or times[v]))
    out[v] = {"primary_plus_secondary_ms": t, "secondary_grays_s": 5 * hits / max(t - pm, 1e-9) / 1e6}
print(json.dumps(out))
