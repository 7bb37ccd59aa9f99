#!/usr/bin/env python3
"""Cost of a chunk crossing: the player steps over a chunk boundary in x, the newly exposed slab of chunks is
generated chunk by chunk (vrt_generate_chunk), the traversal index is updated (vrt_reset_distance_field) and the
next trace levels the column-top pyramid.  Host wall time per stage with a stream wait after each, median over the
crossings.  usage: crossing_probe.py [grid_n ...]   (grid_n = 32 * (2 * render_distance + 3))"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import GraphicsCore, GraphicsSettings, Player, TRACE_SKIP, VOXEL_PACKED_U8, scenes

for n in [int(a) for a in sys.argv[1:]] or [160, 480, 1056]:
    rd = (n // 32 - 3) // 2
    assert 32 * (2 * rd + 3) == n, "grid_n must be 32 * (2 * render_distance + 3)"
    W, H = (640, 480) if n == 160 else (1920, 1080)
    with GraphicsCore(settings=GraphicsSettings(render_distance=rd), voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.kernel_time_sampling(0)
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        pos = [0.0, 40.3, 0.0]
        g.upload_chunks_around_player(pos, None)
        old = (0, 1, 0)
        gen, gen_batched, index, level, frame_ms, chunks = [], [], [], [], [], 0
        for i in range(12):
            pos[0] += 32.0 if i % 4 < 2 else -32.0
            cam = scenes.cam16(Player.at(pos, 2.4, -0.2))
            new = tuple(int(np.floor(v / 32.0)) for v in pos)
            gd = rd + 1
            todo = [(new[0] + x, new[1] + y, new[2] + z) for x in range(-gd, gd + 1) for y in range(-gd, gd + 1)
                    for z in range(-gd, gd + 1)]
            todo = [cp for cp in todo if max(abs(a - b) for a, b in zip(cp, old)) > gd]
            chunks = len(todo)
            t0 = time.perf_counter()
            if i % 2:
                for cp in todo: g.upload_chunk_gpu(cp)
            else:
                g.upload_chunks_gpu(todo)
            g.sync(); t1 = time.perf_counter()
            g.reset_distance_field(); g.sync(); t2 = time.perf_counter()
            g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()}); g.sync(); t3 = time.perf_counter()
            g.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()}); g.sync(); t4 = time.perf_counter()
            old = new
            if i:
                (gen if i % 2 else gen_batched).append(t1 - t0); index.append(t2 - t1); level.append((t3 - t2) - (t4 - t3)); frame_ms.append(t4 - t3)
        # the whole build, for comparison
        g.generate_window(scenes.window_min(pos, n)); g.sync()
        t0 = time.perf_counter(); g.reset_distance_field(); g.sync(); t1 = time.perf_counter()
        med = lambda v: round(float(np.median(v)) * 1e3, 4)
        print(json.dumps({"grid_n": n, "chunks_per_crossing": chunks, "generate_chunk_by_chunk_ms": med(gen), "generate_batched_ms": med(gen_batched), "index_update_ms": med(index),
                          "pyramid_level_ms": med(level), "frame_with_wait_ms": med(frame_ms), "index_whole_build_ms": round((t1 - t0) * 1e3, 4)}))
