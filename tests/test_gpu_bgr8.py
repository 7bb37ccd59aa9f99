"""The three-byte pixel plane (vrt_outputs.bgr8) written by the trace kernels themselves: equal to the bgra8 plane
without its alpha byte -- skip and literal kernels, frame widths that are not multiples of four, the start
orientation (whose middle column has a zero direction component and is traced out of line), row bands, strips,
host and device planes, the entry point that allocates nothing -- and a caller built against the struct without the
member (smaller struct_size) is served as before."""
import ctypes as C

import numpy as np
import pytest
import torch

from raytracing_amd import _lib as L
from raytracing_amd import GraphicsCore, Player, TRACE_LITERAL, TRACE_SKIP, VOXEL_PACKED_U8, VOXEL_REF_U16U16, scenes, sharding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [TRACE_SKIP, TRACE_LITERAL])
@pytest.mark.parametrize("w,h", [(640, 480), (333, 217), (62, 9)])
def test_three_byte_plane_equals_the_pixels_without_alpha(mode, w, h):
    n = 160
    poses = [scenes.pose_b(), Player.at((3.5, 30.25, -7.5), 0.0, 0.0), Player.at((70.2, 50.9, -40.4), 2.2, -0.6)]
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_REF_U16U16, trace_mode=mode) as g:
        g.generate_window(scenes.window_min(poses[0].get_position(), n))
        for p in poses:
            cam = scenes.cam16(p)
            out = g.trace(cam, w, h, outputs=("bgra8", "bgr8", "face"))
            assert out["bgr8"].shape == (h, w, 3)
            assert np.array_equal(out["bgr8"], out["bgra8"][..., :3])
            assert int(out["bgra8"][..., 3].min()) == 255
            only = g.trace(cam, w, h, 3, h - 2, outputs=("bgr8",))["bgr8"]     # a band, and no other plane asked for
            assert np.array_equal(only, out["bgr8"][3:h - 2])
        assert (out["face"] >= 0).any()


def test_three_byte_plane_on_device_strips_and_the_static_entry_point():
    n, w, h, strip_h = 256, 1920, 1080, 16
    p = scenes.pose_b()
    cam = scenes.cam16(p)
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8, trace_mode=TRACE_SKIP) as g:
        g.generate_window(scenes.window_min(p.get_position(), n))
        g.reset_distance_field()
        g.set_frame_overlap(2)
        whole = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        three = [torch.full((h, w, 3), 9, dtype=torch.uint8, device="cuda") for _ in range(3)]
        g.trace_device(cam, w, h, 0, h, {"bgra8": whole.data_ptr()})
        g.trace_device(cam, w, h, 0, h, {"bgr8": three[0].data_ptr()})
        g.trace_device(cam, w, h, 0, h, {"bgr8": three[1].data_ptr(), "bgra8": whole.data_ptr()})
        g.sync()
        g.trace_static_device(cam, w, h, 0, h, {"bgr8": three[2].data_ptr()})
        g.sync()
        for t in three:
            assert torch.equal(t, whole[..., :3])
        for parts in (2, 3):
            for part in range(parts):
                rows = sharding.strip_rows(h, strip_h, part, parts)
                packed = torch.zeros((len(rows), w, 3), dtype=torch.uint8, device="cuda")
                g.trace_strips_device(cam, w, h, strip_h, part, parts, {"bgr8": packed.data_ptr()})
                g.sync()
                assert torch.equal(packed, whole[torch.as_tensor(rows, device="cuda")][..., :3])


def test_a_caller_built_against_the_struct_without_the_member_is_served():
    n, w, h = 64, 64, 16
    p = scenes.pose_b()
    with GraphicsCore(grid_n=n, voxel_format=VOXEL_PACKED_U8) as g:
        g.generate_window(scenes.window_min(p.get_position(), n))
        cam = np.ascontiguousarray(scenes.cam16(p), np.float32)
        camp = cam.ctypes.data_as(C.POINTER(C.c_float))
        want = g.trace(cam, w, h, outputs=("bgra8",))["bgra8"]
        got = np.zeros((h, w, 4), np.uint8)
        junk = 0xdeadbeef0   # what lies behind the older struct in the caller's memory must not be read as a pointer
        old_size = C.sizeof(L.VrtOutputs) - C.sizeof(C.c_void_p)
        o = L.VrtOutputs(old_size, L.MEM_HOST, got.ctypes.data, None, None, None, None, None, None, junk)
        L.check(L.lib().vrt_trace(g._ctx, camp, w, h, 0, h, C.byref(o)))
        assert np.array_equal(got, want)
        o.struct_size = old_size - 4
        assert L.lib().vrt_trace(g._ctx, camp, w, h, 0, h, C.byref(o)) == L.VRT_E_INVALID_ARG
