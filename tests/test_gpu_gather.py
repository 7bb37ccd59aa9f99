"""The device path of sharding.gather_frame on the GPU box: one rank, RCCL, the context's own stream.  The
padding copy, the gather and the row reorder must all be ordered behind the trace that fills the packed
buffer (a stream the default stream does not wait for), so the frame is compared on every one of many
back-to-back rounds, each with a different camera."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from raytracing_amd import GraphicsCore, Player, TRACE_SKIP, VOXEL_PACKED_U8, VrtError, scenes, sharding

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def single_rank_rccl():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_padded_strip_buffers_are_gathered_behind_the_trace(single_rank_rccl):
    W, H, strip_h, n = 1920, 1080, 8, 256          # 135 strips of 8 rows
    p0 = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p0.get_position(), n))
        g.reset_distance_field()
        rows = sharding.max_strip_rows(H, strip_h, 1)
        assert rows == H
        packed = torch.zeros((rows, W, 4), dtype=torch.uint8, device="cuda")
        whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for i in range(24):
            cam = scenes.cam16(Player.at(p0.get_position(), scenes.POSE_B_YAW + 0.21 * i, scenes.POSE_B_PITCH))
            g.trace_strips_device(cam, W, H, strip_h, 0, 1, {"bgra8": packed.data_ptr()})
            # padded by 40 rows: the copy into the padded buffer is the step that must wait for the trace
            frame = sharding.gather_frame(packed, H, strip_h, dst=0, core=g, pad_rows=rows + 40)
            g.trace_device(cam, W, H, 0, H, {"bgra8": whole.data_ptr()})
            g.sync()
            torch.cuda.synchronize()
            assert torch.equal(frame, whole), i
        with pytest.raises(ValueError):
            sharding.gather_frame(packed, H, strip_h, dst=0, core=g, pad_rows=rows - 1)


@pytest.mark.parametrize("pixels", ["bgra", "bgr"])
def test_gather_waits_for_traces_on_every_stream_of_the_context(single_rank_rccl, pixels):
    """With two frames in flight (vrt_set_frame_overlap(2)) the strip trace that fills the packed buffer may run on the
    context's second stream; gather_frame(core=...) joins the context's streams before it touches the buffer.  Two
    packed buffers are traced back to back (one per stream) and each is gathered right away, 16 rounds; with pixels="bgr"
    the rows also cross the gather without their alpha byte."""
    W, H, strip_h, n = 1920, 1080, 16, 256
    p0 = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p0.get_position(), n))
        g.reset_distance_field()
        g.set_frame_overlap(2)
        packed = [torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        whole = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for i in range(16):
            cams = [scenes.cam16(Player.at(p0.get_position(), scenes.POSE_B_YAW + 0.21 * i + 0.1 * j, scenes.POSE_B_PITCH)) for j in range(2)]
            for j in range(2):      # one launch per stream of the context
                g.trace_strips_device(cams[j], W, H, strip_h, 0, 1, {"bgra8": packed[j].data_ptr()})
            frames = [sharding.gather_frame(packed[j], H, strip_h, dst=0, core=g, pad_rows=H + 8, pixels=pixels) for j in range(2)]
            g.sync()
            torch.cuda.synchronize()
            for j in range(2):
                g.trace_device(cams[j], W, H, 0, H, {"bgra8": whole.data_ptr()})
                g.sync()
                assert torch.equal(frames[j], whole), (i, j)


def test_alpha_is_dropped_and_restored_around_a_transfer():
    """vrt_pack_bgr / vrt_unpack_bgra on a traced frame (alpha 255 everywhere, as the shader stores it) and on random
    bytes (the alpha byte is not carried: it comes back as 255)."""
    W, H, n = 1000, 333, 64   # a pixel count that is a multiple of 4 but not of 256
    p0 = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p0.get_position(), n))
        frame = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
        g.trace_device(scenes.cam16(p0), W, H, 0, H, {"bgra8": frame.data_ptr()})
        noise = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for src in (frame, noise):
            packed = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
            back = torch.zeros((H, W, 4), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            g.pack_bgr(src.data_ptr(), packed.data_ptr(), W * H)
            g.unpack_bgra(packed.data_ptr(), back.data_ptr(), W * H)
            g.sync()
            assert torch.equal(packed, src[..., :3])
            assert torch.equal(back[..., :3], src[..., :3]) and bool((back[..., 3] == 255).all())
        assert torch.equal(frame[..., 3], torch.full((H, W), 255, dtype=torch.uint8, device="cuda"))
        from raytracing_amd import VrtError
        with pytest.raises(VrtError):
            g.pack_bgr(frame.data_ptr(), packed.data_ptr(), W * H - 1)


@pytest.mark.parametrize("w,h,strip_h,parts", [(1920, 1080, 16, 8), (640, 250, 16, 3), (64, 37, 4, 5)])
def test_reassembly_of_three_byte_parts_restores_rows_and_alpha(w, h, strip_h, parts):
    """vrt_assemble_strips_bgr against the two steps it fuses (vrt_assemble_strips on BGR8 rows, then vrt_unpack_bgra):
    random pixels, ragged last strips, padding rows between the parts; and its argument checks."""
    from raytracing_amd import sharding
    with GraphicsCore(grid_n=64, voxel_format=VOXEL_PACKED_U8) as g:
        stride = sharding.max_strip_rows(h, strip_h, parts) + 2
        gen = torch.Generator(device="cuda").manual_seed(w * 31 + h)
        parts3 = torch.randint(0, 256, (parts, stride, w, 3), dtype=torch.uint8, device="cuda", generator=gen)
        rows3 = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        two_steps = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        fused = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        g.assemble_strips(parts3.data_ptr(), rows3.data_ptr(), w * 3, h, strip_h, parts, stride)
        g.unpack_bgra(rows3.data_ptr(), two_steps.data_ptr(), w * h)
        g.assemble_strips_bgr(parts3.data_ptr(), fused.data_ptr(), w, h, strip_h, parts, stride)
        g.sync()
        assert torch.equal(fused, two_steps)
        assert int(fused[..., 3].min()) == 255
        for r in range(parts):          # and row by row against the definition
            rows = sharding.strip_rows(h, strip_h, r, parts)
            assert torch.equal(fused[torch.as_tensor(rows, device="cuda")][..., :3], parts3[r][:len(rows)])
        with pytest.raises(VrtError):
            g.assemble_strips_bgr(parts3.data_ptr(), fused.data_ptr(), w - 1, h, strip_h, parts, stride)
        with pytest.raises(VrtError):
            g.assemble_strips_bgr(parts3.data_ptr(), fused.data_ptr(), w, h, strip_h, parts, stride - 3)
