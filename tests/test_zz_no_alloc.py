"""vrt_trace_static -- the trace entry point that allocates nothing (its tables and dispatch-order slots live in a
static device array of the library) -- against vrt_trace, through the C ABI: every plane bit for bit, on two calls
in a row, the second with a smaller frame (the array keeps the first call's contents), then again after more calls
of both kinds, from a second context on the same GPU, and with the literal kernel; and its argument checks."""
import ctypes as C

import numpy as np
import pytest
import torch

from raytracing_amd import _lib as L
from raytracing_amd import GraphicsCore, Player, TRACE_LITERAL, TRACE_SKIP, VOXEL_PACKED_U8, VrtError, scenes

pytestmark = pytest.mark.gpu
TORCH = {"bgra8": (4, torch.uint8), "voxel": (3, torch.int32), "face": (1, torch.int32), "steps": (1, torch.int32),
         "t": (1, torch.float32), "hit_point": (3, torch.float32), "rgba": (4, torch.float32)}


def planes(rows, w):
    return {k: torch.full((rows, w) if c == 1 else (rows, w, c), 7, dtype=dt, device="cuda") for k, (c, dt) in TORCH.items()}


def both_ways(g, cam, w, h, y0=0, y1=None):
    y1 = h if y1 is None else y1
    a, b = planes(y1 - y0, w), planes(y1 - y0, w)
    g.trace_device(cam, w, h, y0, y1, {k: v.data_ptr() for k, v in a.items()})
    g.sync()
    g.trace_static_device(cam, w, h, y0, y1, {k: v.data_ptr() for k, v in b.items()})
    g.sync()
    for k in TORCH:
        x, y = a[k].cpu().numpy(), b[k].cpu().numpy()
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert np.array_equal(x, y), (k, w, h, y0, y1)
    return a


def test_static_entry_point_equals_vrt_trace_on_two_calls_in_a_row_the_second_smaller():
    n = 256
    p = scenes.pose_b()
    cam = scenes.cam16(p)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p.get_position(), n))
        # the index is not built yet: the entry point may not build it (that allocates) and says so
        out = planes(8, 64)
        with pytest.raises(VrtError) as e:
            g.trace_static_device(cam, 64, 8, 0, 8, {k: v.data_ptr() for k, v in out.items()})
        assert e.value.code == L.VRT_E_UNSUPPORTED and "vrt_reset_distance_field" in str(e.value)
        g.reset_distance_field()

        first = both_ways(g, cam, 1920, 1080)            # call 1
        second = both_ways(g, cam, 640, 360)             # call 2: a smaller frame over the same array
        assert float((first["face"] >= 0).float().mean()) > 0.3 and float((second["face"] >= 0).float().mean()) > 0.3
        # more calls in a row: the dispatch order measured by earlier launches comes into use, the window moves
        # (other chunk: tables refilled), bands, an odd size
        for i in range(6):
            both_ways(g, cam, 640, 360)
        cam2 = scenes.cam16(Player.at((70.2, 50.9, -40.4), 2.2, -0.6))
        both_ways(g, cam2, 1280, 720)
        both_ways(g, cam2, 1280, 720, 100, 333)
        both_ways(g, cam, 333, 217)
        # a second context on the same GPU shares the array
        with GraphicsCore(grid_n=160, trace_mode=TRACE_SKIP) as g2:
            p2 = scenes.pose_b()
            g2.generate_window(scenes.window_min(p2.get_position(), 160))
            g2.reset_distance_field()
            both_ways(g2, cam, 640, 480)
            both_ways(g, cam, 640, 360)
            both_ways(g2, cam, 640, 480)
        # limits and argument checks
        big = planes(1, 1)
        ptrs = {k: v.data_ptr() for k, v in big.items()}
        for w, h in ((8193, 16), (16, 8193), (0, 16)):
            with pytest.raises(VrtError) as e:
                g.trace_static_device(cam, w, h, 0, min(h, 16), ptrs)
            assert e.value.code == L.VRT_E_INVALID_ARG
        with pytest.raises(VrtError) as e:                 # 8192 x 4096 pixels: 524288 wavefronts
            g.trace_static_device(cam, 8192, 4096, 0, 4096, ptrs)
        assert e.value.code == L.VRT_E_INVALID_ARG and "wavefronts" in str(e.value)
        host = np.zeros((8, 64, 4), np.uint8)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_HOST, host.ctypes.data, None, None, None, None, None, None)
        camp = np.ascontiguousarray(cam, np.float32).ctypes.data_as(C.POINTER(C.c_float))
        assert L.lib().vrt_trace_static(g._ctx, camp, 64, 8, 0, 8, C.byref(o)) == L.VRT_E_INVALID_ARG


def test_static_entry_point_with_the_literal_kernel():
    n = 160
    p = scenes.pose_b()
    cam = scenes.cam16(p)
    with GraphicsCore(grid_n=n, trace_mode=TRACE_LITERAL) as g:
        g.generate_window(scenes.window_min(p.get_position(), n))
        both_ways(g, cam, 640, 480)
        both_ways(g, cam, 320, 200)
        assert g.last_kernel_name() == "literal"


def test_static_entry_point_leaves_device_memory_as_it_found_it():
    """What the entry point claims: no device allocation.  Free device memory (hipMemGetInfo through torch) is the same
    before and after 200 calls of mixed shapes that include every state change the path has -- tables refilled for
    other frame shapes, the window moving to another chunk row (the pyramid is re-levelled and the tables refilled),
    bands, a second context taking turns -- while every buffer the test itself needs exists beforehand."""
    n = 256
    p = scenes.pose_b()
    cams = [scenes.cam16(p), scenes.cam16(Player.at((70.2, 50.9, -40.4), 2.2, -0.6)),      # another chunk in x, y and z
            scenes.cam16(Player.at((3.3, 90.7, -5.1), 0.4, -1.1))]
    shapes = [(1920, 1080, 0, 1080), (640, 360, 0, 360), (1280, 720, 100, 333), (333, 217, 0, 217)]
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g, \
            GraphicsCore(grid_n=160, trace_mode=TRACE_SKIP) as g2:
        g.generate_window(scenes.window_min(p.get_position(), n))
        g.reset_distance_field()
        g2.generate_window(scenes.window_min(p.get_position(), 160))
        g2.reset_distance_field()
        out = planes(1080, 1920)
        ptrs = {k: v.data_ptr() for k, v in out.items()}
        for cam in cams:                      # one ordinary trace per window position and context: the index is complete for all
            g.trace_device(cam, 64, 32, 0, 32, ptrs)
            g2.trace_device(cam, 64, 32, 0, 32, ptrs)
        g.trace_static_device(cams[0], 64, 32, 0, 32, ptrs)
        g.sync(); g2.sync()
        torch.cuda.synchronize()
        free0, total = torch.cuda.mem_get_info()
        for i in range(200):
            w, h, y0, y1 = shapes[i % len(shapes)]
            core = g2 if i % 7 == 3 else g
            core.trace_static_device(cams[(i // 5) % len(cams)], w, h, y0, y1, ptrs)
            if i % 50 == 49:
                g.sync(); g2.sync()
                assert torch.cuda.mem_get_info()[0] == free0, i
        g.sync(); g2.sync()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info() == (free0, total)
        # and the last frame is still the right one
        w, h, y0, y1 = shapes[1]
        both_ways(g, cams[2], w, h, y0, y1)


def test_static_entry_point_keeps_two_contexts_and_two_threads_apart():
    """Every context of a GPU shares the one static scratch.  Launches of two contexts on their two streams, enqueued
    back to back without a host wait, come out right (the later stream waits for the earlier launch); and two host
    threads that call at once never tear the scratch: a call either goes through, and its frame is right, or returns
    VRT_E_UNSUPPORTED ("in use") with nothing enqueued."""
    import threading
    p = scenes.pose_b()
    cam_a, cam_b = scenes.cam16(p), scenes.cam16(Player.at(p.get_position(), scenes.POSE_B_YAW + 1.3, -0.5))
    w, h = 640, 360
    with GraphicsCore(grid_n=160, trace_mode=TRACE_SKIP) as ga, GraphicsCore(grid_n=160, trace_mode=TRACE_SKIP) as gb:
        for g in (ga, gb):
            g.generate_window(scenes.window_min(p.get_position(), 160))
            g.reset_distance_field()
        want = {}
        for name, g, cam in (("a", ga, cam_a), ("b", gb, cam_b)):
            ref = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
            g.trace_device(cam, w, h, 0, h, {"bgra8": ref.data_ptr()})
            g.sync()
            want[name] = ref
        # two streams, no host wait in between
        fa = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(24)]
        fb = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(24)]
        for i in range(24):
            ga.trace_static_device(cam_a, w, h, 0, h, {"bgra8": fa[i].data_ptr()})
            gb.trace_static_device(cam_b, w, h, 0, h, {"bgra8": fb[i].data_ptr()})
        ga.sync(); gb.sync()
        assert all(torch.equal(f, want["a"]) for f in fa) and all(torch.equal(f, want["b"]) for f in fb)

        # two threads
        results = {"a": [], "b": []}

        def worker(name, g, cam, frames):
            for f in frames:
                try:
                    g.trace_static_device(cam, w, h, 0, h, {"bgra8": f.data_ptr()})
                    results[name].append(True)
                except VrtError as e:
                    assert e.code == L.VRT_E_UNSUPPORTED and "in use" in str(e), str(e)
                    results[name].append(False)

        for f in fa + fb:
            f.zero_()
        torch.cuda.synchronize()
        ta = threading.Thread(target=worker, args=("a", ga, cam_a, fa * 20))
        tb = threading.Thread(target=worker, args=("b", gb, cam_b, fb * 20))
        ta.start(); tb.start(); ta.join(); tb.join()
        ga.sync(); gb.sync()
        for name, frames in (("a", fa), ("b", fb)):
            ok = results[name]
            assert len(ok) == 480 and any(ok)
            for i, f in enumerate(frames):          # a buffer some successful call wrote holds the frame; the others are untouched
                wrote = any(ok[i + 24 * r] for r in range(20))
                assert torch.equal(f, want[name]) if wrote else int(f.max()) == 0, (name, i)
        print("refused calls:", results["a"].count(False), results["b"].count(False))
