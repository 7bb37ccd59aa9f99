"""vrt_comm_* on the GPU box: one process, one device, so a communicator of one context.  RCCL is loaded and
initialised (ncclCommInitAll), and with VRT_COMM_SELF_SENDRECV=1 the root's own part travels through a grouped
ncclSend / ncclRecv on the context's stream, the path every further GPU's part takes; without it, by a device
copy.  Either way the gather is enqueued behind the trace that fills the packed buffer, with no host wait in
between, so the frame is compared on every one of many back-to-back rounds, each with a different camera."""
import numpy as np
import pytest
import torch

from raytracing_amd import GraphicsCore, Player, TRACE_SKIP, VOXEL_PACKED_U8, VrtError, scenes, sharding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("through_rccl", ["1", "0"])
def test_single_context_gather_of_strips_and_of_whole_frames(through_rccl, monkeypatch):
    monkeypatch.setenv("VRT_COMM_SELF_SENDRECV", through_rccl)
    W, H, strip_h, n = 1280, 720, 8, 256
    p0 = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p0.get_position(), n))
        g.reset_distance_field()
        with sharding.Comm([g]) as comm:
            assert comm.size() == 1
            stride = H + 24  # padding rows beyond the part's own
            packed = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            gathered = torch.zeros((stride, W, 4), dtype=torch.uint8, device="cuda")
            frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            batch = torch.zeros((1, H, W, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for i in range(16):
                cam = scenes.cam16(Player.at(p0.get_position(), scenes.POSE_B_YAW + 0.37 * i, scenes.POSE_B_PITCH))
                g.trace_strips_device(cam, W, H, strip_h, 0, 1, {"bgra8": packed.data_ptr()})
                comm.gather_strips([packed.data_ptr()], gathered.data_ptr(), frame.data_ptr(), W * 4, H, strip_h, stride)
                g.trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
                comm.gather([whole.data_ptr()], H * W * 4, batch.data_ptr())
                g.sync()
                assert torch.equal(frame, whole), i
                assert torch.equal(batch[0], whole), i
            with pytest.raises(VrtError):  # stride smaller than the part
                comm.gather_strips([packed.data_ptr()], gathered.data_ptr(), frame.data_ptr(), W * 4, H, strip_h, H - 1)
        monkeypatch.delenv("VRT_COMM_ALLOW_SHARED_DEVICE", raising=False)
        with pytest.raises(VrtError):  # two contexts on one device
            with GraphicsCore(grid_n=64, voxel_format=VOXEL_PACKED_U8) as g2:
                sharding.Comm([g, g2])


@pytest.mark.parametrize("through_rccl", ["1", "0"])
def test_single_context_gather_of_strips_with_three_byte_pixels(through_rccl, monkeypatch):
    """vrt_gather_strips_bgr with the one context the box allows: the context packs its rows (alpha dropped), the BGR8
    part travels (through RCCL with VRT_COMM_SELF_SENDRECV=1), the reorder kernel restores the alpha byte: equal to
    the frame traced whole, on 16 rounds back to back."""
    monkeypatch.setenv("VRT_COMM_SELF_SENDRECV", through_rccl)
    W, H, strip_h, n = 1280, 720, 16, 256
    p0 = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p0.get_position(), n))
        g.reset_distance_field()
        with sharding.Comm([g]) as comm:
            stride = H + 8
            packed = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            packed3 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
            gathered = torch.zeros((stride, W, 3), dtype=torch.uint8, device="cuda")
            frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for i in range(16):
                cam = scenes.cam16(Player.at(p0.get_position(), scenes.POSE_B_YAW + 0.37 * i, scenes.POSE_B_PITCH))
                g.trace_strips_device(cam, W, H, strip_h, 0, 1, {"bgra8": packed.data_ptr()})
                comm.gather_strips_bgr([packed.data_ptr()], [packed3.data_ptr()], gathered.data_ptr(), frame.data_ptr(), W, H,
                                       strip_h, stride)
                g.trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
                g.sync()
                assert torch.equal(frame, whole), i
            with pytest.raises(VrtError):  # a width that is not a multiple of four pixels
                comm.gather_strips_bgr([packed.data_ptr()], [packed3.data_ptr()], gathered.data_ptr(), frame.data_ptr(), W - 2, H,
                                       strip_h, stride)


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_control_flow_of_several_contexts_rehearsed_on_one_device(parts, monkeypatch):
    """The n-context paths of vrt_gather, vrt_gather_strips and vrt_gather_strips_bgr -- part sizes, offsets, strides,
    joins, the order between the contexts' streams -- with all contexts on the box's one GPU (VRT_COMM_ALLOW_SHARED_DEVICE=1:
    parts on the root's device are copied between two events instead of travelling through RCCL, which needs a device
    per rank).  Every context holds its own replica of the world and traces its own strips / its own pose; frames in
    flight on two streams of every context; eight rounds back to back without a host wait inside a round."""
    monkeypatch.setenv("VRT_COMM_ALLOW_SHARED_DEVICE", "1")
    W, H, strip_h, n = 640, 360, 16, 64          # 23 strips: parts of different sizes for 2, 3 and 8 contexts
    p0 = Player.at((3.4, 20.7, -5.1), 0.6, -0.5)
    wmin = scenes.window_min(p0.get_position(), n)
    cores = [GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) for _ in range(parts)]
    try:
        for g in cores:
            g.generate_window(wmin)
            g.reset_distance_field()
            g.set_frame_overlap(2)
        with sharding.Comm(cores) as comm:
            assert comm.size() == parts
            stride = sharding.max_strip_rows(H, strip_h, parts) + 3
            packed = [torch.zeros((stride, W, 4), dtype=torch.uint8, device="cuda") for _ in range(parts)]
            packed3 = [torch.zeros((stride, W, 3), dtype=torch.uint8, device="cuda") for _ in range(parts)]
            gathered = torch.zeros((parts, stride, W, 4), dtype=torch.uint8, device="cuda")
            gathered3 = torch.zeros((parts, stride, W, 3), dtype=torch.uint8, device="cuda")
            frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            frame3 = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            poses = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(parts)]
            batch = torch.zeros((parts, H, W, 4), dtype=torch.uint8, device="cuda")
            whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for i in range(8):
                cam = scenes.cam16(Player.at(p0.get_position(), 0.6 + 0.41 * i, -0.5))
                cams = [scenes.cam16(Player.at(p0.get_position(), 0.6 + 0.41 * i + 0.2 * r, -0.45)) for r in range(parts)]
                for r, g in enumerate(cores):
                    g.trace_strips_device(cam, W, H, strip_h, r, parts, {"bgra8": packed[r].data_ptr()})
                comm.gather_strips([t.data_ptr() for t in packed], gathered.data_ptr(), frame.data_ptr(), W * 4, H, strip_h, stride)
                comm.gather_strips_bgr([t.data_ptr() for t in packed], [t.data_ptr() for t in packed3], gathered3.data_ptr(),
                                       frame3.data_ptr(), W, H, strip_h, stride)
                for r, g in enumerate(cores):
                    g.trace_device(cams[r], W, H, 0, H, {"bgra8": poses[r].data_ptr()})
                comm.gather([t.data_ptr() for t in poses], H * W * 4, batch.data_ptr())
                cores[0].sync()
                cores[0].trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
                cores[0].sync()
                assert torch.equal(frame, whole), i
                assert torch.equal(frame3, whole), i
                for r in range(parts):
                    cores[0].trace_device(cams[r], W, H, 0, H, {"bgra8": whole.data_ptr()})
                    cores[0].sync()
                    assert torch.equal(batch[r], whole), (i, r)
    finally:
        for g in cores:
            g.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process: RCCL between devices (never run by the build: its box has one)")
def test_two_contexts_on_two_devices_gather_through_rccl():
    """The path no one-GPU box can run: ncclCommInitAll over two devices, the peer's part through ncclSend / ncclRecv.
    Two contexts, each with a replica of the world on its own device; vrt_gather of each context's own pose and
    vrt_gather_strips / _bgr of one frame, compared with the same frames traced on the root alone."""
    W, H, strip_h, n, parts = 640, 360, 16, 64, 2
    p0 = Player.at((3.4, 20.7, -5.1), 0.6, -0.5)
    wmin = scenes.window_min(p0.get_position(), n)
    cores = [GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP, device=r) for r in range(parts)]
    try:
        for g in cores:
            g.generate_window(wmin)
            g.reset_distance_field()
        with sharding.Comm(cores) as comm:
            stride = sharding.max_strip_rows(H, strip_h, parts) + 3
            dev = [f"cuda:{r}" for r in range(parts)]
            packed = [torch.zeros((stride, W, 4), dtype=torch.uint8, device=dev[r]) for r in range(parts)]
            packed3 = [torch.zeros((stride, W, 3), dtype=torch.uint8, device=dev[r]) for r in range(parts)]
            poses = [torch.zeros((H, W, 4), dtype=torch.uint8, device=dev[r]) for r in range(parts)]
            gathered = torch.zeros((parts, stride, W, 4), dtype=torch.uint8, device=dev[0])
            gathered3 = torch.zeros((parts, stride, W, 3), dtype=torch.uint8, device=dev[0])
            frame, frame3, whole = (torch.zeros((H, W, 4), dtype=torch.uint8, device=dev[0]) for _ in range(3))
            batch = torch.zeros((parts, H, W, 4), dtype=torch.uint8, device=dev[0])
            for r in range(parts):
                torch.cuda.synchronize(r)
            for i in range(8):
                cam = scenes.cam16(Player.at(p0.get_position(), 0.6 + 0.41 * i, -0.5))
                cams = [scenes.cam16(Player.at(p0.get_position(), 0.6 + 0.41 * i + 0.2 * r, -0.45)) for r in range(parts)]
                for r, g in enumerate(cores):
                    g.trace_strips_device(cam, W, H, strip_h, r, parts, {"bgra8": packed[r].data_ptr()})
                comm.gather_strips([t.data_ptr() for t in packed], gathered.data_ptr(), frame.data_ptr(), W * 4, H, strip_h, stride)
                comm.gather_strips_bgr([t.data_ptr() for t in packed], [t.data_ptr() for t in packed3], gathered3.data_ptr(),
                                       frame3.data_ptr(), W, H, strip_h, stride)
                for r, g in enumerate(cores):
                    g.trace_device(cams[r], W, H, 0, H, {"bgra8": poses[r].data_ptr()})
                comm.gather([t.data_ptr() for t in poses], H * W * 4, batch.data_ptr())
                for g in cores:
                    g.sync()
                cores[0].trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
                cores[0].sync()
                assert torch.equal(frame, whole), i
                assert torch.equal(frame3, whole), i
                for r in range(parts):
                    cores[0].trace_device(cams[r], W, H, 0, H, {"bgra8": whole.data_ptr()})
                    cores[0].sync()
                    assert torch.equal(batch[r], whole), (i, r)
    finally:
        for g in cores:
            g.close()
