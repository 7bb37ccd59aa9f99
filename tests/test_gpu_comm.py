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
        with pytest.raises(VrtError):  # two contexts on one device
            with GraphicsCore(grid_n=64, voxel_format=VOXEL_PACKED_U8) as g2:
                sharding.Comm([g, g2])
