#!/usr/bin/env python3
"""Per-lane trip counts of the two-phase loop on the bench frame (diagnostic build -DVRT_DIAG_ITERS through VRT_LIB) and
what lane scheduling could buy: wavefront-trips (a 16 x 4 tile runs as long as its longest lane) against the lane-trips
they hold, and the same if every ray were cut off after K trips and the rest traced densely in a second pass."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline()
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    out = g.trace(scenes.cam16(p), W, H, outputs=("steps", "face"))
it = (out["steps"].astype(np.int64) & 0xffff)
Hc = H // 2 * 2
t = it[:Hc].reshape(Hc // 2, 2, W // 32, 32)   # wave tiles of 32 x 2 pixels
mx = t.max(axis=(1, 3))
res = {"lane_trips": int(it.sum()), "wave_trips": int(mx.sum()), "lanes_per_wave_trip": float(it[:Hc].sum() / mx.sum()),
       "mean_trips_per_ray": float(it.mean()), "p50_p90_p99_max": [int(v) for v in np.percentile(it, [50, 90, 99, 100])]}
for K in (4, 6, 8, 12):
    first = np.minimum(mx, K).sum()                       # wave-trips of the capped first pass
    rest = np.maximum(it - K, 0)
    left = int((rest > 0).sum())
    # second pass, perfectly packed 64 rays per wavefront in descending order of what is left
    r = np.sort(rest[rest > 0])[::-1]
    pad = (-len(r)) % 64
    r = np.concatenate([r, np.zeros(pad, r.dtype)]).reshape(-1, 64)
    second = int(r.max(axis=1).sum())
    res[f"cap_{K}"] = {"first_pass_wave_trips": int(first), "rays_left": left, "second_pass_wave_trips": second,
                       "total_vs_now": float((first + second) / mx.sum())}
print(json.dumps(res, indent=1))
# unit steps (the loop's second phase): per lane, and a bound on how often a wavefront runs that phase -- at least as
# often as its busiest lane needs (two steps per trip), at most once per trip in which any lane walks
us = out["steps"].astype(np.int64) >> 16
u = us[:Hc].reshape(Hc // 2, 2, W // 32, 32)
hit = (out["face"] >= 0)
print(json.dumps({"unit_steps_per_ray": float(us.mean()), "unit_steps_per_hit_ray": float(us[hit].mean()) if hit.any() else 0.0,
                  "waves": int(mx.size), "waves_with_a_hit": int((hit[:Hc].reshape(Hc // 2, 2, W // 32, 32).any(axis=(1, 3))).sum()),
                  "trips_per_wave": float(mx.mean()),
                  "phase_b_runs_per_wave_at_least": float(np.ceil(u.max(axis=(1, 3)) / 2).mean()),
                  "lanes_that_walk_per_wave": float((u > 0).sum(axis=(1, 3)).mean())}, indent=1))
