#!/usr/bin/env python3
"""Host time of the pieces of bench.py's multi-rank step, on one rank over RCCL: microseconds per call.  usage: host_cost_probe.py"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from raytracing_amd import GraphicsCore, VOXEL_PACKED_U8, TRACE_SKIP, scenes
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
W, H, N, G = 1920, 1080, 256, 4
p = scenes.pose_b(); cam = scenes.cam16(p)
def t(fn, reps=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    return round(dt, 1)
with GraphicsCore(grid_n=N, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
    g.generate_window(scenes.window_min(p.get_position(), N)); g.reset_distance_field(); g.sync()
    g.kernel_time_sampling(0); g.set_frame_overlap(2)
    stream = torch.cuda.ExternalStream(g.stream(), device=torch.device("cuda", 0))
    b3 = torch.zeros((G, H, W, 3), dtype=torch.uint8, device="cuda")
    out = [torch.zeros((G, H, W, 3), dtype=torch.uint8, device="cuda")]
    arrived = torch.zeros((G, H, W, 4), dtype=torch.uint8, device="cuda")
    small = torch.zeros((16,), dtype=torch.uint8, device="cuda"); small_out = [torch.zeros_like(small)]
    res = {}
    res["trace_device (ctypes call)"] = t(lambda: g.trace_device(cam, W, H, 0, H, {"bgr8": b3[0].data_ptr()}))
    def ctx():
        with torch.cuda.stream(stream): pass
    res["with torch.cuda.stream(...)"] = t(ctx)
    res["core.join"] = t(g.join); res["core.fork"] = t(g.fork)
    def gather_big():
        with torch.cuda.stream(stream):
            w = dist.gather(b3, out, dst=0, async_op=True); w.wait()
    res["dist.gather of a batch + wait"] = t(gather_big, 50)
    def gather_small():
        with torch.cuda.stream(stream):
            w = dist.gather(small, small_out, dst=0, async_op=True); w.wait()
    res["dist.gather of 16 bytes + wait"] = t(gather_small, 50)
    res["unpack launch"] = t(lambda: g.unpack_bgra(out[0].data_ptr(), arrived.data_ptr(), G * H * W), 50)
    ag_out = torch.zeros((1, 16), dtype=torch.uint8, device="cuda")
    def ag():
        with torch.cuda.stream(stream):
            w = dist.all_gather_into_tensor(ag_out, small, async_op=True); w.wait()
    res["dist.all_gather_into_tensor of 16 bytes + wait"] = t(ag, 50)
    for k, v in res.items(): print(f"{k:50s} {v:8.1f} us")
dist.destroy_process_group()
