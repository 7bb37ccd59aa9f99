#!/usr/bin/env python3
"""Headline benchmark: primary Mrays/s at 1920x1080 over a 1024^3 u8 voxel grid.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One step = one full 1080p frame of primary rays (ray generation + traversal + shading +
BGRA8 store) through libvrt.so.  The world (default procedural terrain over the 1024^3
window around the camera) is generated on the GPU and indexed once, before timing, and
stays resident in HBM.  With N > 1 every rank holds a replica of the world and traces a
frame of its own camera pose (weak scaling); the BGRA8 framebuffers are gathered to rank
0 over RCCL inside the timed region, `--gather-every` frames per gather.  `--shard strips` is the other partition of the path: ONE
frame dealt to the ranks in interleaved 16-row strips (strong scaling), gathered and put back
in row order on rank 0 by the library's copy kernel.

Prints ONE JSON line on rank 0.

`roofline` names the roof that binds the kernel of the timed region.  The skip kernel moves about 15 MB per
frame and is bound by vector-instruction issue: `achieved` = wave64 vector instructions per launch (SQ_INSTS_VALU of
the committed counter passes of this command, profiles/traffic.json -- used only when that record carries the git
blob hashes of the kernel sources this run was built from, else `valu_count_stale` is true and no fraction is
printed) / the launch's duration measured in this run; `peak` = the chip's issue rate for the kernels' own mix of
vector instructions (fma, integer add, compare + select, min/max, conversions) with eight wavefronts per SIMD at
the clock the chip holds under that load, measured in this run by libvrt_selftest.so; `frac_fma_peak` prices the
same rate against independent v_fma_f32 (a hotter load: the chip clocks lower under it), `frac_datasheet` against
1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction.  `roofline.counters` carries the cache hit rates, lanes
active per vector instruction and wait fractions of the same counter passes.  `roofline.hbm` is the memory side of
the same launch: counter bytes (FETCH_SIZE corrected + WRITE_SIZE) / the same duration against the 8 TB/s data-sheet
peak.  `roofline.reference_equivalent_GBs` is the rate at which the REFERENCE's loop would have to move bytes to
produce the frame in that time (one voxel byte per loop iteration per ray + 4 bytes of BGRA8 per ray); the
skip kernel does not move those bytes, so this is a speed-up figure, not a bandwidth.
`roofline_literal` is the literal kernel (one voxel load per reference iteration) on the same frame: the
kernel whose HBM-side work IS the algorithmic byte count, priced against the HBM peak.
`cpu_baseline` = the CPU oracle (a C restatement of the reference shader, not the
reference itself) timed on a bounded sample of the same frame on this host's cores.

Started as plain `python bench.py --gpus N` with N > 1 (no WORLD_SIZE in the environment), this process
touches no GPU: it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child,
relays the child's JSON line and exits with its code.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracing_amd import GraphicsCore, Player, TRACE_AUTO, TRACE_LITERAL, TRACE_SKIP, VOXEL_PACKED_U8, scenes, sharding  # noqa: E402
from raytracing_amd.kernel_sources import blob_hashes  # noqa: E402

W, H, N_GRID = 1920, 1080, 1024
STRIP_H = 16
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
VALU_DATASHEET_GINSTR_S = 1024 * 2.4 / 2.0   # 256 CUs x 4 SIMDs, 2.4 GHz, 2 cycles per wave64 v_fma_f32: 1228.8 G/s


def rank_pose(rank, world):
    """Rank r looks from the headline position with its yaw turned by 2*pi*r/world."""
    base = scenes.pose_headline()
    if world == 1:
        return base
    yaw = scenes.POSE_B_YAW + 2.0 * np.pi * rank / world
    return Player.at(base.get_position(), yaw, scenes.POSE_B_PITCH)


def cpu_baseline(core, cam, steps_plane):
    """Oracle on the whole frame, strip by strip (about 30 CPU-seconds on 16 threads)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    block = core.download_world()
    world = O.World(block)
    threads = O.host_threads()
    strips = list(range(0, (H + 15) // 16))
    rays = 0
    t0 = time.perf_counter()
    ok = True
    for s in strips:
        out = O.trace(world, cam, W, H, s * 16, min(H, s * 16 + 16), nthreads=threads, want=("bgra8", "steps"))
        rays += out["steps"].size
        ok = ok and np.array_equal(out["steps"], steps_plane[s * 16:s * 16 + 16])
    dt = time.perf_counter() - t0
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"the same 1080p frame, all {len(strips)} 16-row strips ({rays} rays, {dt:.1f} s wall)",
            "matches_gpu_steps": bool(ok)}


def valu_issue_rates(device):
    """Issue rate of wave64 vector instructions on this GPU, measured now by libvrt_selftest.so (long streams of
    independent instructions on every SIMD of every compute unit)."""
    import ctypes as C
    lib = C.CDLL(os.path.join(ROOT, "raytracing_amd", "libvrt_selftest.so"))
    lib.vrt_selftest_valu_rate.restype = C.c_int
    lib.vrt_selftest_valu_rate.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    res = {}
    for key, kind, waves in (("fma_8_waves", 0, 8), ("trace_mix_8_waves", 2, 8), ("fma_1_wave", 0, 1)):
        out = (C.c_double * 4)()
        rc = lib.vrt_selftest_valu_rate(device, kind, waves, out)
        if rc != 0:
            raise RuntimeError(f"vrt_selftest_valu_rate failed with HIP error {rc}")
        res[key] = {"ginstr_s": out[0] / 1e9, "shader_ghz": out[1], "cycles_per_instr_per_simd": out[2]}
    return res


def literal_roofline(device, cam, wmin, alg_bytes, skip_frame):
    """The literal kernel on the same frame (a second context with its own copy of the world): one voxel byte
    per reference iteration + 4 bytes per ray over the mean HIP-event duration of 16 launches."""
    with GraphicsCore(width=W, height=H, device=device, grid_n=N_GRID, voxel_format=VOXEL_PACKED_U8,
                      trace_mode=TRACE_LITERAL) as lit:
        lit.generate_window(wmin)
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        for _ in range(2):
            lit.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
        lit.sync()
        lit.kernel_time_sampling(1)
        lit.kernel_time_reset()
        for _ in range(16):
            lit.trace_device(cam, W, H, 0, H, {"bgra8": frame.data_ptr()})
        ms, n = lit.kernel_time()
        same = bool(torch.equal(frame, skip_frame))
    s = ms / 1e3 / n
    return {"bound": "hbm", "achieved": alg_bytes / s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg_bytes / s / 1e9 / HBM_PEAK_GBS, "kernel": "literal", "kernel_ms": s * 1e3, "timed_launches": int(n),
            "algorithmic_bytes_per_launch": alg_bytes, "frame_equals_skip_kernel_frame": same,
            "limiter": "one dependent voxel load per iteration (about 0.65 us per iteration of a wavefront) beside 55-75 vector "
                       "instructions per iteration from each of the SIMD's eight wavefronts: load latency and instruction "
                       "issue bind together, HBM bandwidth does not (profiles/r02_dropped/literal_lean_loop.txt)"}


def copy_bandwidth_gbs(nbytes=1 << 30, reps=10):
    """Device-to-device copy rate of this GPU (bytes read + bytes written per second), the practical
    ceiling beside the 8 TB/s data-sheet figure."""
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    b = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) / 1e3) / 1e9


def launch_ranks(n):
    """Plain `python bench.py --gpus N`, N > 1: one rank per GPU through torch.distributed.run, as a CHILD
    process (this process has made no HIP call and makes none; a process that has touched the GPU must not
    be replaced by another program).  Rank 0's JSON line is relayed on stdout, everything else on stderr;
    returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL / device-memory sharing between ranks needs dmabuf IPC here
    env.setdefault("OMP_NUM_THREADS", "4")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in child.stdout:
        (sys.stdout if line.lstrip().startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    return child.wait()


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its communicator comes up; this run's stdout is
    reserved for the one JSON line, so the file descriptor is pointed at stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["auto", "skip", "literal"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks, let them find each other over gloo, print {n_gpus, ranks_seen} and stop: "
                         "checks the launch path on a host without GPUs")
    ap.add_argument("--no-literal", action="store_true", help="skip the literal kernel's line (roofline_literal)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the RCCL gather path with a single rank as well (rehearsal of the N > 1 code on one GPU)")
    ap.add_argument("--shard", choices=["poses", "strips"], default="poses",
                    help="poses (the headline job): one frame of its own pose per rank, weak scaling; strips: ONE frame "
                         "dealt to the ranks in interleaved 16-row strips, gathered and put back in row order on rank 0")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="frames per gather (pose sharding, N > 1): each rank fills a buffer of this many frames and hands it "
                         "to one gather, which overlaps the tracing of the next buffer; every frame still reaches rank 0 "
                         "inside the timed region.  1 = one gather per frame")
    ap.add_argument("--gather-pixels", choices=["bgr", "bgr_pack", "bgra"], default="bgr",
                    help="what crosses the links of a gather (pose- or strip-sharded): bgr = three bytes per pixel (the alpha byte, 255 on "
                         "every path of the shader, is not written by the trace kernel -- vrt_outputs.bgr8 -- and restored on rank 0 "
                         "after the gather); bgr_pack = the same bytes made by a pack kernel behind a BGRA8 trace; bgra = the frames as they are")
    ap.add_argument("--settle-ms", type=float, default=30.0,
                    help="untimed launches of the same frame for this long before the warm-up steps: the chip raises its clock over "
                         "some ten milliseconds of load, and a region of 20 steps (0.8 ms) measured right after set-up runs 12-15 %% "
                         "slower than the same region a few regions later (0: none)")
    ap.add_argument("--overlap", type=int, default=2,
                    help="streams the context deals its frames to (vrt_set_frame_overlap): 2 lets the end of one frame's "
                         "launch run under the beginning of the next; 1 = every launch behind the one before.  A strip-sharded "
                         "frame (--shard strips with a gather) stays on one stream")
    ap.add_argument("--time-every", type=int, default=0,
                    help="bracket one frame in this many of the timed region with HIP events as well (0 = none: the two "
                         "events cost about 7 us of stream time per launch; the kernel's duration is measured on "
                         "--probe launches right after the timed region instead)")
    ap.add_argument("--probe", type=int, default=32,
                    help="launches of the same frame, each bracketed by HIP events on the context's stream, from which "
                         "the kernel's mean duration is taken (at least 16)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal of the multi-rank path on hosts without one GPU per rank "
                         "(frames are gathered through host memory; not a measurement)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         "(or plain `python bench.py --gpus N`, which starts them itself)")
    if args.launch_check:
        # the launch path alone (no GPU work): every rank joins a gloo group, rank 0 reports how many it saw
        if world > 1:
            dist.init_process_group("gloo")
        seen = torch.ones(1, dtype=torch.int64)
        if world > 1:
            dist.all_reduce(seen)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "launch check", "n_gpus": world, "ranks_seen": int(seen.item()),
                              "steps": args.steps, "warmup": args.warmup}), flush=True)
        return
    if args.backend == "gloo":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    multi = world > 1 or args.force_gather  # frames are gathered to rank 0
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        with stdout_to_stderr():
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()  # brings the communicator up now rather than inside the timed region
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo")

    mode = {"auto": TRACE_AUTO, "skip": TRACE_SKIP, "literal": TRACE_LITERAL}[args.mode]
    core = GraphicsCore(width=W, height=H, device=local_rank, grid_n=N_GRID, voxel_format=VOXEL_PACKED_U8,
                        trace_mode=mode)
    strips = args.shard == "strips"
    pose = rank_pose(0 if strips else rank, world)
    cam = scenes.cam16(pose)
    wmin = scenes.window_min(pose.get_position(), N_GRID)
    t0 = time.perf_counter()
    core.generate_window(wmin)          # world resident in HBM ...
    core.reset_distance_field()         # ... and indexed, once, before timing
    core.sync()
    build_s = time.perf_counter() - t0

    # (a strip-sharded frame is one launch per rank and step: nothing to overlap)
    overlap = 1 if (multi and strips) else max(1, min(4, args.overlap))
    stream = torch.cuda.ExternalStream(core.stream(), device=torch.device("cuda", local_rank))
    host = args.backend == "gloo"
    # The context's stream is torch's current stream from here on: the collectives and the waits of a step are enqueued on
    # it without entering a stream context per step (6.7 us of host time each, measured: tools/host_cost_probe.py -- with
    # a gather per four frames the host side of a step was 42 us against 36 us of GPU work).
    import contextlib
    torch.cuda.set_stream(stream)
    on_stream = contextlib.nullcontext
    # frame buffers: two (the gather of frame k overlaps the trace of frame k+1), and one per stream when more than
    # two frames are in flight, so that no two launches in flight write one buffer
    frames = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(max(2, overlap))]
    steps_t = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    # pose sharding with N > 1: two buffers of G frames each; the gather of one overlaps the tracing into the other
    G = max(1, args.gather_every)
    gathered = [None, None]
    batch = batch3 = arrived = None
    bgr = args.gather_pixels != "bgra"
    pack = args.gather_pixels == "bgr_pack"   # BGRA8 frames packed by a kernel of their own behind the trace (A/B runs); default: the
                                              # trace kernel writes the three-byte plane itself
    C_SENT = 3 if bgr else 4            # bytes per pixel on the link
    if multi and not strips:
        batch = [torch.zeros((G, H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        if bgr:
            batch3 = [torch.zeros((G, H, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
        if rank == 0:
            gathered = [[torch.zeros((G, H, W, C_SENT), dtype=torch.uint8, device="cpu" if host else "cuda") for _ in range(world)]
                        for _ in range(2)]
            if bgr:   # the frames of every rank as pixels again
                arrived = [torch.zeros((world, G, H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    fill = {"buf": 0, "n": 0, "last": (0, 0)}   # buffer being filled, frames in it, (buffer, slot) of the last frame traced
    if strips:
        pad_rows = sharding.max_strip_rows(H, STRIP_H, world)
        packed = [torch.zeros((pad_rows, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        # what crosses the links: the packed rows, without their alpha byte unless --gather-pixels bgra
        packed3 = [torch.zeros((pad_rows, W, 3), dtype=torch.uint8, device="cuda") for _ in range(2)] if bgr else None
        parts_buf = [torch.zeros((world, pad_rows, W, C_SENT), dtype=torch.uint8, device="cuda") if rank == 0 else None
                     for _ in range(2)]
    pending = [None, None]
    torch.cuda.synchronize()

    # untimed: step counts of this frame -> algorithmic bytes
    core.trace_device(cam, W, H, 0, H, {"bgra8": frames[0].data_ptr(), "steps": steps_t.data_ptr()})
    core.sync()
    steps_np = steps_t.cpu().numpy().view(np.uint32)
    sum_steps = int(steps_np.astype(np.int64).sum())
    launch_rays = W * H
    if strips:                          # this rank's launch covers its own strips only
        mine = sharding.strip_rows(H, STRIP_H, rank, world)
        sum_steps = int(steps_np[mine].astype(np.int64).sum())
        launch_rays = W * len(mine)
    alg_bytes = sum_steps * 1 + launch_rays * 4

    counter = [0]

    # every launch of the timed loop with its arguments converted once (GraphicsCore.bound_trace_device)
    bound = {}
    if strips:
        for b in range(2):
            key = "bgr8" if (bgr and not pack) else "bgra8"
            dst = packed3[b] if key == "bgr8" else packed[b]
            bound[b] = core.bound_trace_strips_device(cam, W, H, STRIP_H, rank, world, {key: dst.data_ptr()})
    elif multi:
        for b in range(2):
            for j in range(G):
                dst = {"bgr8": batch3[b][j].data_ptr()} if (bgr and not pack) else {"bgra8": batch[b][j].data_ptr()}
                bound[(b, j)] = core.bound_trace_device(cam, W, H, 0, H, dst)
    else:
        for i, f in enumerate(frames):
            bound[i] = core.bound_trace_device(cam, W, H, 0, H, {"bgra8": f.data_ptr()})

    def step():
        b = counter[0] & 1
        counter[0] += 1
        if strips and host:
            send = trace_strip_part(b)
            core.sync()
            cpu_parts = [torch.empty((pad_rows, W, C_SENT), dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
            dist.gather(send.cpu(), cpu_parts, dst=0)
            if rank == 0:
                parts_buf[b].copy_(torch.stack(cpu_parts))
                torch.cuda.synchronize()
                assemble(b)
        elif strips:
            with on_stream():
                if pending[b] is not None:
                    pending[b].wait()
                send = trace_strip_part(b)
                if multi:
                    pending[b] = dist.gather(send, list(parts_buf[b].unbind(0)) if rank == 0 else None, dst=0, async_op=True)
                    if rank == 0:
                        pending[b].wait()            # orders the copy below behind the gather on this stream
                        pending[b] = None
                        assemble(b)
        elif multi:
            b, j = fill["buf"], fill["n"]
            with on_stream():
                if j == 0:
                    landed(b)                        # the traces below may overwrite batch[b] only after its gather
                    core.fork()                      # ... on whichever of the context's streams they run
                bound[(b, j)]()
            fill["n"], fill["last"] = j + 1, (b, j)
            if fill["n"] == G:
                flush()
        else:
            bound[(counter[0] - 1) % len(frames)]()

    def trace_strip_part(b):
        """This rank's strips of the frame into buffer b, as they will travel: returns the tensor to send."""
        bound[b]()
        if bgr and not pack:
            return packed3[b]
        if pack:   # the pack kernel runs on the context's stream, behind the trace
            core.pack_bgr(packed[b].data_ptr(), packed3[b].data_ptr(), pad_rows * W)
            return packed3[b]
        return packed[b]

    def assemble(b):
        """Rank 0, strips: the gathered parts of buffer b back in image order (and, if they travelled without it, with
        their alpha byte again), one launch on the context's stream."""
        if bgr:
            core.assemble_strips_bgr(parts_buf[b].data_ptr(), frames[b].data_ptr(), W, H, STRIP_H, world, pad_rows)
        else:
            core.assemble_strips(parts_buf[b].data_ptr(), frames[b].data_ptr(), W * 4, H, STRIP_H, world, pad_rows)

    def landed(b):
        """Called with the context's stream current: that stream waits for the gather of buffer b, and on rank 0 the
        frames that arrived without their alpha byte become pixels again (unpack kernels on the same stream)."""
        if pending[b] is None:
            return
        if pending[b] is not True:
            pending[b].wait()
        pending[b] = None
        if bgr and rank == 0:
            for r in range(world):
                src = gathered[b][r]
                if host:
                    src = src.cuda()
                    torch.cuda.current_stream().synchronize()
                core.unpack_bgra(src.data_ptr(), arrived[b][r].data_ptr(), G * H * W)
                if host:
                    core.sync()

    def flush():
        """Hand the buffer being filled to the gather (a partial buffer goes whole: its tail holds older frames)."""
        b = fill["buf"]
        if batch is None or fill["n"] == 0:
            return
        send = batch3[b] if bgr else batch[b]
        if pack:
            core.pack_bgr(batch[b].data_ptr(), batch3[b].data_ptr(), G * H * W)   # on the context's stream, behind the traces (it joins them)
        else:
            core.join()                              # the context's stream waits for the traces on its other streams
        if host:
            core.sync()
            dist.gather(send.cpu(), gathered[b], dst=0)
            pending[b] = True
        else:
            with on_stream():
                # RCCL gather to rank 0, ordered behind the traces (and the pack) on the context's stream
                pending[b] = dist.gather(send, gathered[b], dst=0, async_op=True)
        fill["buf"], fill["n"] = b ^ 1, 0

    def fence():
        flush()
        for b in range(2):
            if pending[b] is not None:
                if batch is not None:
                    with on_stream():
                        landed(b)
                else:
                    with on_stream():
                        pending[b].wait()
                    pending[b] = None
        core.sync()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    core.set_frame_overlap(overlap)
    # the clock: untimed launches of this rank's frame until the chip holds its sustained clock (see --settle-ms); then the
    # warm-up steps proper, which also fill the gather's buffers
    settle = [core.bound_trace_device(cam, W, H, 0, H, {"bgra8": f.data_ptr()}) for f in frames[:2]]
    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
        for i in range(16):
            settle[i & 1]()
        core.sync()
    for _ in range(args.warmup):
        step()
    fence()
    time_every = args.time_every if args.time_every or overlap == 1 else 8   # overlapped launches: their durations in the region too
    core.kernel_time_sampling(max(0, min(time_every, args.steps)))
    core.kernel_time_reset()

    def timed_region():
        """EXACTLY --steps steps between two fences (stream waits, and a barrier with more than one rank)."""
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        enqueue_s.append(time.perf_counter() - t0)   # host time to enqueue the region's work (no wait in it unless a queue is full)
        fence()
        return time.perf_counter() - t0

    # A region of a few milliseconds is at the mercy of one slow wake-up from its closing wait: such a run times the
    # region three times (each one the same --steps steps between the same fences) and reports the median, with all
    # three in `region_ms`.  The decision is taken on the first region's time and agreed between the ranks.
    enqueue_s = []
    regions = [timed_region()]
    again = torch.tensor([1.0 if regions[0] < 20e-3 else 0.0], dtype=torch.float64, device="cpu" if host else "cuda")
    if multi:
        dist.all_reduce(again, op=dist.ReduceOp.MAX)
    if again.item() > 0:
        regions += [timed_region(), timed_region()]
    region_ms, region_launches = core.kernel_time()

    # the kernel alone: the same launch again, one at a time, every one bracketed by two HIP events on the context's stream
    core.set_frame_overlap(1)
    probe = max(16, args.probe)
    core.kernel_time_sampling(1)
    core.kernel_time_reset()
    for _ in range(probe):
        if strips:
            core.trace_strips_device(cam, W, H, STRIP_H, rank, world, {"bgra8": packed[0].data_ptr()})
        else:
            core.trace_device(cam, W, H, 0, H, {"bgra8": frames[0].data_ptr()})
    core.sync()
    kern_ms, launches = core.kernel_time()
    core.kernel_time_sampling(0)

    kern_max_s = kern_ms / 1e3 / max(launches, 1)
    if multi:   # every region's time is the slowest rank's
        t = torch.tensor(regions + [kern_max_s], dtype=torch.float64, device="cpu" if host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions, kern_max_s = [float(v) for v in t[:-1].tolist()], float(t[-1].item())
    dt = sorted(regions)[len(regions) // 2]

    gather_ok = None
    if rank == 0 and multi and not strips:
        # what arrived on rank 0 for every rank's last frame against that rank's pose traced here (the ranks share
        # the world window: they look from one position with different yaws)
        lb, lj = fill["last"]
        check = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        gather_ok = True
        for r in range(world):
            core.trace_device(scenes.cam16(rank_pose(r, world)), W, H, 0, H, {"bgra8": check.data_ptr()})
            core.sync()
            got = arrived[lb][r][lj] if bgr else gathered[lb][r][lj]
            gather_ok = gather_ok and bool(torch.equal(got.cpu(), check.cpu()))
        core.trace_device(cam, W, H, 0, H, {"bgra8": check.data_ptr()})   # leave the context on this rank's own pose
        core.sync()
    if rank == 0:
        rays = W * H * args.steps * (1 if strips else world)
        kern_s = kern_ms / 1e3 / max(launches, 1)
        name = core.last_kernel_name()
        ref_equiv_gbs = alg_bytes / kern_s / 1e9
        # the committed counter record of this command, if it was measured on the kernel sources of this tree
        traffic = valu = valu_alone = counters = None
        stale = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tf) and not strips:   # the counters describe whole-frame launches
            try:
                prof = json.load(open(tf))
                stale = prof.get("kernel_sources") != blob_hashes()
                if not stale:
                    key = name + "_plain_order" if overlap > 1 and (name + "_plain_order") in prof else name
                    traffic = prof.get(key)
                    valu = prof.get(key + "_valu_instructions")
                    valu_alone = prof.get(name + "_valu_instructions")
                    counters = prof.get(key + "_counters")
            except Exception:
                traffic = valu = valu_alone = counters = None
        hbm = {"achieved": traffic / kern_s / 1e9 if traffic else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": traffic / kern_s / 1e9 / HBM_PEAK_GBS if traffic else None,
               "bytes": "FETCH_SIZE x 2 (gfx950 tallies 128-B fills as 64 B) + WRITE_SIZE per launch, profiles/traffic.json"}
        common = {"traffic": traffic, "kernel_ms": kern_s * 1e3, "timed_launches": int(launches),
                  "timed_how": f"{int(launches)} launches of the same frame right after the timed region, one at a time, each "
                               "bracketed by two HIP events on the context's stream",
                  "kernel_mrays_s": launch_rays / kern_s / 1e6, "mean_steps_per_ray": sum_steps / launch_rays,
                  "algorithmic_bytes_per_launch": alg_bytes, "reference_equivalent_GBs": ref_equiv_gbs}
        if region_launches:
            common["region_sampled_kernel_ms"] = region_ms / region_launches
        if name == "literal":
            # one voxel load per reference iteration: the algorithmic bytes are this kernel's own work
            roofline = {"bound": "hbm", "achieved": ref_equiv_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ref_equiv_gbs / HBM_PEAK_GBS, **common}
        else:
            issue = valu_issue_rates(local_rank)
            # one launch at a time: instructions of a launch / its duration.  Overlapped launches share the GPU, so a
            # launch's own duration says little; the rate is then instructions of a launch / the region's time per launch
            per_launch_s = kern_s if overlap == 1 else dt / args.steps
            achieved = valu / per_launch_s / 1e9 if valu else None
            alone = (valu_alone or valu) / kern_s / 1e9 if valu else None
            mix, fma = issue["trace_mix_8_waves"]["ginstr_s"], issue["fma_8_waves"]["ginstr_s"]
            roofline = {"bound": "valu_issue",
                        "achieved": achieved, "peak": mix, "unit": "G wave64 vector instructions/s",
                        "frac": achieved / mix if valu else None,
                        "peak_how": "issue rate of the trace kernels' own instruction mix (fma, integer add, compare + select, min/max, "
                                    "float->int, floor, xor; eight wavefronts per SIMD, every SIMD of the chip) at the clock the chip holds "
                                    "under it, measured in this run (peak_measured.trace_mix_8_waves)",
                        "frac_fma_peak": achieved / fma if valu else None,
                        "frac_datasheet": achieved / VALU_DATASHEET_GINSTR_S if valu else None,
                        "datasheet_peak": VALU_DATASHEET_GINSTR_S,
                        "achieved_how": ("instructions of a launch / mean duration of a launch (kernel_ms)" if overlap == 1 else
                                         f"instructions of a launch / time per launch over the timed region (ms_per_step): {overlap} "
                                         "streams keep launches overlapping, a launch alone takes kernel_ms, in the region "
                                         "region_sampled_kernel_ms"),
                        "frac_one_launch_alone": alone / mix if valu else None,
                        "frac_one_launch_alone_datasheet": alone / VALU_DATASHEET_GINSTR_S if valu else None,
                        "valu_instructions_per_launch": valu,
                        "valu_count_stale": stale,
                        "valu_count_from": "SQ_INSTS_VALU per launch, rocprofv3 --pmc pass of this command (profiles/traffic.json), taken only "
                                           "when the record's kernel-source hashes are those of this tree",
                        "counters": counters,
                        "peak_measured": issue, "hbm": hbm, **common}
        copy_gbs = copy_bandwidth_gbs()
        roofline["measured_copy_GBs"] = copy_gbs
        line = {
            "metric": "primary Mrays/s at 1920x1080 on 1024^3 grid",
            "value": rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "settle_ms": args.settle_ms,   # untimed launches before the warm-up steps (clock ramp), see --settle-ms
            "host_enqueue_ms_per_step": sorted(enqueue_s)[len(enqueue_s) // 2] / args.steps * 1e3,   # this rank's host side of a step
            "region_ms": [r * 1e3 for r in regions],   # every timed region (each: --steps steps between two fences); value and ms_per_step are the median's
            # the same job without the framebuffer gather: every rank's frame / the slowest rank's kernel
            "value_without_gather": rays / args.steps / kern_max_s / 1e6,
            "higher_is_better": True,
            "scaling": "strong" if strips else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "frames_in_flight": overlap,
            "config": {"workload": "1920x1080 primary rays over 1024^3 u8 procedural terrain (pose B at y=40.3), "
                                   + ("world replicated per GPU, ONE frame dealt in interleaved 16-row strips, "
                                      "gather (three bytes per pixel) + row reorder on rank 0" if strips else
                                      "world replicated per GPU, one pose per GPU, BGRA8 gather to rank 0"
                                      + (f" ({G} frames per gather)" if multi else "")),
                       "kernel": name, "grid_n": N_GRID, "width": W, "height": H,
                       "world_build_s": round(build_s, 3)},
            "roofline": roofline,
        }
        if world == 1 and name == "skip" and not args.no_literal:
            line["roofline_literal"] = literal_roofline(local_rank, cam, wmin, alg_bytes, frames[0])
        # the frame just timed against the frozen oracle output for this pose (tests/golden/fullsize)
        gold = os.path.join(ROOT, "tests", "golden", "fullsize", "headline_1080p_n1024.json")
        if os.path.exists(gold) and core.last_kernel_name() in ("skip", "literal"):
            import hashlib
            fx = json.load(open(gold))
            if np.array_equal(np.asarray(fx["cam_col_major_bits"], np.uint32), np.asarray(cam, np.float32).view(np.uint32)):
                last_frame = frames[(counter[0] - 1) & 1] if strips else frames[(counter[0] - 1) % len(frames)]
                if batch is not None and bgr and not pack:   # traced as three-byte pixels: the frame as rank 0 holds it after the gather
                    last_frame = arrived[fill["last"][0]][0][fill["last"][1]]
                elif batch is not None:
                    last_frame = batch[fill["last"][0]][fill["last"][1]]
                last = last_frame.cpu().numpy()
                line["parity"] = {"bgra8_sha256_equals_oracle_golden": hashlib.sha256(last.tobytes()).hexdigest() == fx["sha256"]["bgra8"],
                                  "steps_sum_equals_oracle_golden": sum_steps == fx["totals"]["steps"] if not strips else None}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(core, cam, steps_np)
        if multi and not strips:
            line["gather"] = {"bytes_per_pixel_on_the_link": C_SENT, "frames_per_gather": G,
                              "last_frame_of_every_rank_equals_its_trace": gather_ok}
        if multi and strips:
            line["gather"] = {"bytes_per_pixel_on_the_link": C_SENT, "bytes_per_frame_on_the_links": (world - 1) * pad_rows * W * C_SENT}
        print(json.dumps(line), flush=True)

    if strips and multi and rank == 0:
        # the reassembled frame is the frame traced in one piece
        whole = torch.zeros_like(frames[0])
        core.trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
        core.sync()
        assert torch.equal(frames[(counter[0] - 1) & 1], whole), "strips did not reassemble to the frame"
    elif multi and rank == 0:
        assert gather_ok, "a gathered frame differs from the frame its rank traced"
    core.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
