#!/usr/bin/env python3
"""Distribution of the secondary rays' trip counts (two-phase loop) on the secondary-ray workload, from a diagnostic build
of the library (make EXTRA=-DVRT_DIAG_SEC_TRIPS, loaded through VRT_LIB) with VRT_SECONDARY=pixel.
usage: VRT_LIB=raytracing_amd/ab_diagsec.so sec_trips_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VRT_SECONDARY"] = "pixel"
import numpy as np
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    face = g.trace(cam, W, H, outputs=("face",))["face"]
    d = g.trace_secondary(cam, W, H)
hit = face >= 0
sh, ao, tot = d["shadow"][hit].astype(int), d["ao"][hit].astype(int), d["sec_steps"][hit].astype(int)
for name, v in (("shadow ray trips", sh), ("longest AO ray trips", ao), ("mean trips per ray (sum/5)", tot / 5)):
    q = np.percentile(v, [10, 25, 50, 75, 90, 99, 100])
    print(f"{name}: mean {v.mean():.1f}  p10/25/50/75/90/99/max {q}")
# per 16x16 tile: mean over max
t = tot.sum() / 5 / hit.sum()
print("mean trips per secondary ray", t)
