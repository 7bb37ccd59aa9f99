#!/usr/bin/env python3
"""Diagnostic: when each wave of the skip kernel starts, how long it runs and how many loop trips
it makes, on the bench workload.  Needs the diagnostic library build (never ship it):
    make -C raytracing_amd/csrc clean && make -C raytracing_amd/csrc EXTRA="-DVRT_DIAG_CLOCK -DVRT_DIAG_ITERS"
whose face plane carries the wave's start tick (100 MHz), steps plane its duration and t plane
the loop counts (low half: outer trips, high half: unit steps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
W, H, N = 1920, 1080, 1024
p = scenes.pose_headline(); cam = scenes.cam16(p)
g = GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP)
g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
face = torch.zeros((H, W), dtype=torch.int32, device="cuda"); steps = torch.zeros((H, W), dtype=torch.int32, device="cuda")
tt = torch.zeros((H, W), dtype=torch.float32, device="cuda")
for _ in range(3):
    g.trace_device(cam, W, H, 0, H, {"face": face.data_ptr(), "steps": steps.data_ptr(), "t": tt.data_ptr()})
g.sync()
st = face.cpu().numpy().astype(np.int64); du = steps.cpu().numpy().astype(np.int64)
it = tt.cpu().numpy().astype(np.int64); trips = it & 0xffff; units = it >> 16
t0 = st.min(); st -= t0; en = st + du
print("kernel span %.1f us" % (en.max() / 100.0))
print("block row: start min..max us | mean duration us | end max us | max trips | max unit steps")
for r in range(0, H, 16):
    s, d, e = st[r:r+16], du[r:r+16], en[r:r+16]
    print("%4d  %6.1f..%6.1f | %6.1f | %6.1f | %4d | %4d" % (r, s.min() / 100, s.max() / 100, d.mean() / 100, e.max() / 100,
                                                   trips[r:r+16].max(), units[r:r+16].max()))
# per wave (16x4 tile): duration against its trip counts
Hh, Ww = H // 4 * 4, W // 16 * 16
def tiles(a, f): return f(f(a[:Hh, :Ww].reshape(Hh // 4, 4, Ww // 16, 16), axis=3), axis=1)
wd = tiles(du, np.max).ravel() / 100.0; wt = tiles(trips, np.max).ravel(); wu = tiles(units, np.max).ravel()
ws = tiles(st, np.min).ravel() / 100.0
print("slowest waves: duration us, start us, trips, unit steps")
for i in np.argsort(-wd)[:15]:
    print("  %6.1f %6.1f %4d %4d" % (wd[i], ws[i], wt[i], wu[i]))
A = np.stack([wt, wu, np.ones_like(wt)], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(A, wd, rcond=None)
print("least squares: duration ~ %.3f us * trips + %.3f us * unit steps + %.2f us" % tuple(coef))
for lo, hi in [(0, 10), (10, 30), (30, 60), (60, 100)]:
    m = (ws >= lo) & (ws < hi) & (wt > 8)
    if m.sum() > 50:
        c, *_ = np.linalg.lstsq(A[m], wd[m], rcond=None)
        print("  waves starting in [%d,%d) us (%d): %.3f us/trip %.3f us/unit %.2f" % (lo, hi, m.sum(), *c))
ws2 = st[::4, ::16].ravel(); we2 = en[::4, ::16].ravel()
for t in range(0, min(int(en.max()), 100000) + 1, 500):
    print("t=%5.1f us live waves %d" % (t / 100, int(((ws2 <= t) & (we2 > t)).sum())))
