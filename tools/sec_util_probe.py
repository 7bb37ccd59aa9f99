#!/usr/bin/env python3
"""Where the lanes of the binned secondary-ray kernel go on the secondary-ray workload: counters of a diagnostic build
(tools/build_flags.sh diagutil -DVRT_DIAG_SEC_UTIL; loaded through VRT_LIB), summed over one launch.
usage: VRT_LIB=raytracing_amd/ab_diagutil.so sec_util_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field()
    st = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    g.trace_secondary_device(cam, W, H, 0, H, None, None, st.data_ptr()); g.sync()
    c = [int(v) for v in st.flatten()[:14].cpu()]
names = ["wave_trips", "live_lane_trips", "phaseA_lane_trips", "phaseB_step1_lanes", "phaseB_step2_lanes", "wave_trips_with_phaseA",
         "refills", "free_lanes_at_refill", "rays_begun", "waves", "lanes_near_a_plane", "wave_trips_that_settle_near_planes",
         "wave_trips_after_the_list_ran_dry", "live_lane_trips_after_the_list_ran_dry"]
d = dict(zip(names, c))
d["live_lanes_per_trip"] = d["live_lane_trips"] / max(d["wave_trips"], 1)
d["phaseA_lanes_per_trip_that_runs_it"] = d["phaseA_lane_trips"] / max(d["wave_trips_with_phaseA"], 1)
d["phaseB_lanes_per_trip"] = d["phaseB_step1_lanes"] / max(d["wave_trips"], 1)
d["trips_per_ray"] = d["live_lane_trips"] / max(d["rays_begun"], 1)
d["lanes_begun_per_refill"] = d["rays_begun"] / max(d["refills"], 1)
d["live_lanes_per_trip_after_the_list_ran_dry"] = d["live_lane_trips_after_the_list_ran_dry"] / max(d["wave_trips_after_the_list_ran_dry"], 1)
print(json.dumps(d, indent=1))
