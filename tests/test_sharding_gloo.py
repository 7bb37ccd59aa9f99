"""world_size-2 (and 3) CPU rehearsal of the multi-GPU frame path over gloo: every rank
produces its interleaved strips, the packed buffers are gathered to rank 0 and scattered
back into the frame.  On CPU the strips come from the oracle (the checker standing in for
the tracer); the code under test is raytracing_amd.sharding and the strip arithmetic that
vrt_trace_strips / vrt_strip_rows implement on the library side."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import _lib as L
from raytracing_amd import scenes, sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, strip_h, q):
    sys.path.insert(0, HERE)
    import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = scenes.pose_b()
        block = O.terrain_fill(64, scenes.window_min(p.get_position(), 64), np.uint8)
        world_o = O.World(block)
        cam = scenes.cam16(p)
        rows = sharding.strip_rows(h, strip_h, rank, world)
        assert len(rows) == L.lib().vrt_strip_rows(h, strip_h, rank, world)
        packed = np.zeros((len(rows), w, 4), np.uint8)
        i = 0
        for s in range(rank, -(-h // strip_h), world):       # trace my strips, packed in strip order
            y0, y1 = s * strip_h, min(h, (s + 1) * strip_h)
            packed[i:i + (y1 - y0)] = O.trace(world_o, cam, w, h, y0, y1, nthreads=1, want=("bgra8",))["bgra8"]
            i += y1 - y0
        frame = sharding.gather_frame(torch.from_numpy(packed), h, strip_h, dst=0)
        if rank == 0:
            full = O.trace(world_o, cam, w, h, nthreads=2, want=("bgra8",))["bgra8"]
            q.put(bool(np.array_equal(frame.numpy(), full)))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,strip_h", [(2, 48, 16), (2, 80, 16), (3, 64, 8)])
def test_strip_sharded_frame_gathers_to_the_full_frame(world, h, strip_h):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, 64, strip_h, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_strip_rows_cover_the_frame_once():
    for h, sh, parts in [(2160, 16, 8), (1080, 8, 4), (480, 16, 3), (40, 16, 2)]:
        allrows = np.concatenate([sharding.strip_rows(h, sh, p, parts) for p in range(parts)])
        assert sorted(allrows.tolist()) == list(range(h))


def test_pose_sharding_covers_every_pose_once():
    for count, parts in [(64, 8), (64, 3), (5, 8), (1, 1)]:
        dealt = [k for part in range(parts) for k in sharding.pose_indices(count, part, parts)]
        assert sorted(dealt) == list(range(count))
        sizes = [len(sharding.pose_indices(count, part, parts)) for part in range(parts)]
        assert max(sizes) - min(sizes) <= 1
