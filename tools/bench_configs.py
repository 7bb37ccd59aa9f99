#!/usr/bin/env python3
"""Kernel-time measurements of the parity/bench configurations other than the headline one
(one GPU; HIP-event time of the trace kernel per launch).  Prints one JSON line per config.
usage: bench_configs.py [2 3 4 5 ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, Player, VOXEL_PACKED_U8, VOXEL_REF_U16U16, scenes

which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5]


def world(n, fmt, pose):
    g = GraphicsCore(grid_n=n, voxel_format=fmt)
    t0 = time.perf_counter()
    g.generate_window(scenes.window_min(pose.get_position(), n)); g.reset_distance_field(); g.sync()
    return g, time.perf_counter() - t0


def timed(g, fn, reps):
    for _ in range(3): fn()
    g.sync(); g.kernel_time_reset()
    for _ in range(reps): fn()
    ms, n = g.kernel_time()
    return ms / n


def in_flight(g, fns, reps, streams=2):
    """Wall time per call with the calls dealt to `streams` streams of the context (frames in flight), ms."""
    g.kernel_time_sampling(0); g.set_frame_overlap(streams)
    for f in fns: f()
    g.sync()
    dt = 1e9
    for _ in range(2):   # (the better of two passes: the first one after another context's memory went back to the driver can stall once)
        t0 = time.perf_counter()
        for _ in range(reps):
            for f in fns: f()
        g.sync()
        one = (time.perf_counter() - t0) / (reps * len(fns))
        print(f"  in flight, pass: {one * 1e3:.4f} ms per call", file=sys.stderr)
        dt = min(dt, one)
    g.set_frame_overlap(1); g.kernel_time_sampling(1)
    return dt * 1e3


def steps_sum(g, cam, w, h):
    st = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    g.trace_device(cam, w, h, 0, h, {"steps": st.data_ptr()}); g.sync()
    return int(st.to(torch.int64).sum().item())


for c in which:
    if c == 1:
        p = scenes.pose_b(); w, h, n = 640, 480, 160
        g, build = world(n, VOXEL_REF_U16U16, p); cam = scenes.cam16(p)
        fr = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        ms = timed(g, lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()}), 50)
        s = steps_sum(g, cam, w, h)
        frs = [torch.zeros_like(fr) for _ in range(4)]
        ov = {k: in_flight(g, [(lambda i=i: g.trace_device(cam, w, h, 0, h, {"bgra8": frs[i].data_ptr()})) for i in range(4)], 50, streams=k) for k in (2, 4)}
        print(json.dumps({"config": 1, "desc": "640x480 over 160^3 u16+u16 (the reference's own window)", "kernel_ms": ms,
                          "frames_in_flight_ms_per_frame": ov,
                          "mrays_s": w*h/ms/1e3, "alg_GBs": (s*2 + w*h*(2+4))/ms/1e6, "build_s": build}))
    elif c == 2:
        p = scenes.pose_b(); w, h, n = 1920, 1080, 256
        g, build = world(n, VOXEL_REF_U16U16, p); cam = scenes.cam16(p)
        fr = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        ms = timed(g, lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()}), 50)
        s = steps_sum(g, cam, w, h)
        print(json.dumps({"config": 2, "desc": "1920x1080 over 256^3 u16+u16", "kernel_ms": ms, "mrays_s": w*h/ms/1e3,
                          "alg_GBs": (s*2 + w*h*(2+4))/ms/1e6, "build_s": build}))
    elif c == 3:
        p = scenes.pose_headline(); w, h, n = 3840, 2160, 1024
        g, build = world(n, VOXEL_PACKED_U8, p); cam = scenes.cam16(p)
        fr = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        ms = timed(g, lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()}), 30)
        s = steps_sum(g, cam, w, h)
        rows = g.trace_strips(cam, w, h, 16, 3, 8, outputs=("bgra8",))["bgra8"].shape[0]
        ms8 = timed(g, lambda: g.trace_strips_device(cam, w, h, 16, 3, 8, {"bgra8": fr.data_ptr()}), 30)
        fr2 = torch.zeros_like(fr)
        ov = in_flight(g, [lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()}),
                           lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr2.data_ptr()})], 30)
        print(json.dumps({"config": 3, "desc": "3840x2160 over 1024^3 u8", "kernel_ms": ms, "mrays_s": w*h/ms/1e3,
                          "two_frames_in_flight_ms_per_frame": ov, "two_frames_in_flight_mrays_s": w*h/ov/1e3,
                          "alg_GBs": (s + w*h*4)/ms/1e6, "one_of_8_strip_parts_ms": ms8, "part_rows": rows, "build_s": build}))
    elif c == 4:
        n, w, h = 2048, 1920, 1080
        poses = scenes.flythrough_poses(64)
        g, build = world(n, VOXEL_PACKED_U8, Player.at((0.0, 60.3, 0.0), 0, 0))
        fr = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        cams = [scenes.cam16(q) for q in poses]
        def fly():
            for cam in cams: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()})
        ms = timed(g, fly, 3)   # per launch
        frs = [torch.zeros_like(fr) for _ in range(4)]
        ov = in_flight(g, [(lambda cam=cam, i=i: g.trace_device(cam, w, h, 0, h, {"bgra8": frs[i % 4].data_ptr()})) for i, cam in enumerate(cams)], 3)
        print(json.dumps({"config": 4, "desc": "64 poses x 1920x1080 over 2048^3 u8 (8 GiB)", "kernel_ms_per_pose": ms,
                          "two_frames_in_flight_ms_per_pose": ov, "two_frames_in_flight_mrays_s": w*h/ov/1e3,
                          "mrays_s": w*h/ms/1e3, "build_s": build}))
    elif c == 5:
        p = scenes.pose_headline(); w, h, n = 1920, 1080, 1024
        g, build = world(n, VOXEL_PACKED_U8, p); cam = scenes.cam16(p)
        sh = torch.zeros((h, w), dtype=torch.uint8, device="cuda"); ao = torch.zeros_like(sh)
        st = torch.zeros((h, w), dtype=torch.int32, device="cuda"); fc = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        fr = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        prim = timed(g, lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()}), 30)
        both = timed(g, lambda: g.trace_secondary_device(cam, w, h, 0, h, sh.data_ptr(), ao.data_ptr(), st.data_ptr()), 30)
        g.trace_device(cam, w, h, 0, h, {"face": fc.data_ptr()}); g.sync()
        hits = int((fc >= 0).sum().item())
        bufs = [(torch.zeros_like(sh), torch.zeros_like(ao), torch.zeros_like(st)) for _ in range(2)]
        both2 = in_flight(g, [(lambda b=b: g.trace_secondary_device(cam, w, h, 0, h, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr())) for b in bufs], 30)
        prim2 = in_flight(g, [lambda: g.trace_device(cam, w, h, 0, h, {"bgra8": fr.data_ptr()})] * 2, 30)
        print(json.dumps({"config": 5, "desc": "1080p primary + 5 secondary rays per hit over 1024^3 u8",
                          "two_frames_in_flight": {"primary_ms": prim2, "primary_plus_secondary_ms": both2,
                                                   "secondary_mrays_s": 5*hits/max(both2-prim2, 1e-9)/1e3},
                          "primary_ms": prim, "primary_mrays_s": w*h/prim/1e3, "primary_plus_secondary_ms": both,
                          "secondary_rays": 5*hits, "secondary_mrays_s": 5*hits/max(both-prim, 1e-9)/1e3,
                          "mean_secondary_steps": float(st.to(torch.int64).sum().item())/max(5*hits, 1)}))
    g.close()
