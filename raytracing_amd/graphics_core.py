"""Host-side mirror of the reference's `GraphicsCore` for the trace path, on top of the
C ABI (include/vrt.h).  Same method names and argument meaning as
src/graphics/mod.rs:146-291,426-428,531-634; the window, swapchain and event loop are
not part of this path, so `redraw` renders into a caller-visible frame instead of
presenting it.

This module holds no traversal arithmetic: everything runs in libvrt.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from .player import PushConstants
from .scenes import chunk_of, grid_n_for_render_distance


class GraphicsSettings:
    """src/graphics/graphics_settings.rs:1-17."""

    def __init__(self, render_distance=1, fov=90.0, mouse_sensitivity=(0.002, 0.002)):
        self.render_distance = render_distance
        self.fov = fov  # never read by the reference either
        self.mouse_sensitivity = mouse_sensitivity


_PLANES = {"bgra8": (4, np.uint8), "voxel": (3, np.int32), "face": (1, np.int32), "steps": (1, np.uint32),
           "t": (1, np.float32), "hit_point": (3, np.float32), "rgba": (4, np.float32), "bgr8": (3, np.uint8)}


class GraphicsCore:
    CHUNK_SIZE = 32  # src/graphics/mod.rs:147
    CHUNK_SIZE_3 = 32 * 32 * 32

    def __init__(self, settings=None, width=640, height=480, device=0, grid_n=None,
                 voxel_format=L.VOXEL_REF_U16U16, trace_mode=L.TRACE_AUTO, max_steps=0):
        """≙ GraphicsCore::new (src/graphics/mod.rs:292-425): device + the two world images.
        `grid_n` defaults to the reference's 32*(2*render_distance+3)."""
        self.settings = settings or GraphicsSettings()
        self.width, self.height = width, height
        self.grid_n = grid_n or grid_n_for_render_distance(self.settings.render_distance)
        self.voxel_format = voxel_format
        self.voxel_dtype = np.uint8 if voxel_format == L.VOXEL_PACKED_U8 else np.uint16
        cfg = L.VrtConfig(C.sizeof(L.VrtConfig), device, self.grid_n, voxel_format, max_steps, trace_mode)
        self._ctx = C.c_void_p()
        L.check(L.lib().vrt_create(C.byref(cfg), C.byref(self._ctx)))

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            for addr in getattr(self, "_host_blocks", []):
                L.lib().vrt_free_host(self._ctx, addr)
            self._host_blocks = []
            L.lib().vrt_destroy(self._ctx)
            self._ctx = None

    def host_array(self, shape, dtype):
        """A numpy array over page-locked host memory from the library (`vrt_alloc_host`), for output
        planes that should come back at the bus rate.  Valid until close()."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        addr = C.c_void_p()
        L.check(L.lib().vrt_alloc_host(self._ctx, max(nbytes, 1), C.byref(addr)))
        if not hasattr(self, "_host_blocks"):
            self._host_blocks = []
        self._host_blocks.append(addr.value)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(addr.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def trace_into(self, cam, w, h, planes, y0=0, y1=None):
        """Trace rows [y0,y1) into caller-provided host arrays (dict name -> contiguous numpy array)."""
        y1 = h if y1 is None else y1
        cam, camp = self._cam(cam)
        for k, a in planes.items():
            c, dt = _PLANES[k]
            if a.dtype != dt or not a.flags.c_contiguous or a.size != (y1 - y0) * w * c:
                raise ValueError(f"plane {k}: need a contiguous {np.dtype(dt).name} array of {(y1 - y0) * w * c} elements")
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_HOST,
                         *[planes[k].ctypes.data if k in planes else None for k in _PLANES])
        L.check(L.lib().vrt_trace(self._ctx, camp, w, h, y0, y1, C.byref(o)))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reference surface ------------------------------------------------
    def get_settings(self):  # mod.rs:426-428
        return self.settings

    def upload_chunk_gpu(self, chunk_pos, game_state=None):  # mod.rs:245-291
        cx, cy, cz = (int(v) for v in chunk_pos)
        L.check(L.lib().vrt_generate_chunk(self._ctx, cx, cy, cz))

    def reset_distance_field(self):  # mod.rs:150-183
        L.check(L.lib().vrt_reset_distance_field(self._ctx))

    def redraw(self, push_constants, outputs=("bgra8",)):
        """≙ redraw (mod.rs:531-634) without acquire/present: one compute pass over the
        frame.  Returns the requested planes as numpy arrays (host copies)."""
        cam = push_constants.column_major() if isinstance(push_constants, PushConstants) else push_constants
        return self.trace(cam, self.width, self.height, outputs=outputs)

    def upload_chunks_gpu(self, chunk_positions):
        """vrt_generate_chunks: a batch of chunks, 64 per launch."""
        flat = [int(v) for cp in chunk_positions for v in cp]
        if flat:
            L.check(L.lib().vrt_generate_chunks(self._ctx, (C.c_int32 * len(flat))(*flat), len(flat) // 3))

    def upload_chunks_around_player(self, position, old_chunk=None, batched=False):
        """The caller-side policy of GameState::upload_chunks_around_player
        (src/game_logic/game_state.rs:50-83), provided for tests of the seam: regenerate
        the chunks that entered the (2*gen_dist+1)^3 window, then reset the distance field."""
        new_chunk = chunk_of(position)
        if old_chunk is not None and max(abs(a - b) for a, b in zip(new_chunk, old_chunk)) == 0:
            return 0
        gen_dist = self.settings.render_distance + 1
        count = 0
        batch = []
        for x in range(-gen_dist, gen_dist + 1):
            for y in range(-gen_dist, gen_dist + 1):
                for z in range(-gen_dist, gen_dist + 1):
                    cp = (new_chunk[0] + x, new_chunk[1] + y, new_chunk[2] + z)
                    if old_chunk is not None and max(abs(a - b) for a, b in zip(cp, old_chunk)) <= gen_dist:
                        continue
                    if batched:
                        batch.append(cp)
                    else:
                        self.upload_chunk_gpu(cp)
                    count += 1
        self.upload_chunks_gpu(batch)
        self.sync()
        self.reset_distance_field()
        return count

    # -- C ABI conveniences ----------------------------------------------
    def upload_world(self, block, dist=None):
        block = np.ascontiguousarray(block, dtype=self.voxel_dtype)
        assert block.shape == (self.grid_n,) * 3
        dp = None
        if dist is not None:
            dist = np.ascontiguousarray(dist, dtype=np.uint16)
            assert dist.shape == block.shape
            dp = dist.ctypes.data
        L.check(L.lib().vrt_upload_world(self._ctx, block.ctypes.data, dp, self.grid_n, L.MEM_HOST))

    def upload_world_device(self, block_ptr, dist_ptr=None):
        L.check(L.lib().vrt_upload_world(self._ctx, block_ptr, dist_ptr, self.grid_n, L.MEM_DEVICE))

    def generate_window(self, wmin):
        w = (C.c_int32 * 3)(*[int(v) for v in wmin])
        L.check(L.lib().vrt_generate_window(self._ctx, w))

    def download_world(self):
        out = np.empty((self.grid_n,) * 3, self.voxel_dtype)
        L.check(L.lib().vrt_download_world(self._ctx, out.ctypes.data, L.MEM_HOST))
        return out

    def download_index(self, pyramid=False):
        """The traversal index as the device holds it: cell records, (n/4)^3 x 4 uint32, and optionally the cells'
        column-top words, (n/4)^3 x 16 uint32 (eight 64-bit words per cell).  Brings the index up to date first."""
        nc = self.grid_n // 4
        cells = np.empty((nc ** 3, 4), np.uint32)
        pyr = np.empty((nc ** 3, 16), np.uint32) if pyramid else None
        L.check(L.lib().vrt_download_index(self._ctx, cells.ctypes.data, pyr.ctypes.data if pyramid else None, L.MEM_HOST))
        return (cells, pyr) if pyramid else cells

    def _cam(self, cam):
        cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
        return cam, cam.ctypes.data_as(C.POINTER(C.c_float))

    def trace(self, cam, w, h, y0=0, y1=None, outputs=("bgra8", "voxel", "face", "steps", "t", "hit_point", "rgba")):
        """Trace rows [y0,y1) into fresh host arrays."""
        y1 = h if y1 is None else y1
        rows = max(0, y1 - y0)  # an inverted band is rejected by the library, not here
        cam, camp = self._cam(cam)
        res = {}
        for k in outputs:
            c, dt = _PLANES[k]
            res[k] = np.empty((rows, w) if c == 1 else (rows, w, c), dt)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_HOST,
                         *[res[k].ctypes.data if k in res else None for k in _PLANES])
        L.check(L.lib().vrt_trace(self._ctx, camp, w, h, y0, y1, C.byref(o)))
        return res

    def trace_secondary(self, cam, w, h, y0=0, y1=None):
        """Shadow + 4 ambient-occlusion rays per primary hit (an extension without a reference
        counterpart); returns host planes shadow (u8), ao (u8), sec_steps (u32)."""
        y1 = h if y1 is None else y1
        rows = max(0, y1 - y0)
        cam, camp = self._cam(cam)
        res = {"shadow": np.empty((rows, w), np.uint8), "ao": np.empty((rows, w), np.uint8),
               "sec_steps": np.empty((rows, w), np.uint32)}
        o = L.VrtSecondaryOutputs(C.sizeof(L.VrtSecondaryOutputs), L.MEM_HOST, res["shadow"].ctypes.data,
                                  res["ao"].ctypes.data, res["sec_steps"].ctypes.data)
        L.check(L.lib().vrt_trace_secondary(self._ctx, camp, w, h, y0, y1, C.byref(o)))
        return res

    def trace_secondary_device(self, cam, w, h, y0, y1, shadow=None, ao=None, sec_steps=None):
        cam, camp = self._cam(cam)
        o = L.VrtSecondaryOutputs(C.sizeof(L.VrtSecondaryOutputs), L.MEM_DEVICE, shadow, ao, sec_steps)
        L.check(L.lib().vrt_trace_secondary(self._ctx, camp, w, h, y0, y1, C.byref(o)))

    def trace_strips(self, cam, w, h, strip_h, part, parts, outputs=("bgra8",)):
        rows = L.lib().vrt_strip_rows(h, strip_h, part, parts)
        cam, camp = self._cam(cam)
        res = {}
        for k in outputs:
            c, dt = _PLANES[k]
            res[k] = np.empty((rows, w) if c == 1 else (rows, w, c), dt)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_HOST,
                         *[res[k].ctypes.data if k in res else None for k in _PLANES])
        L.check(L.lib().vrt_trace_strips(self._ctx, camp, w, h, strip_h, part, parts, C.byref(o)))
        return res

    def trace_device(self, cam, w, h, y0, y1, ptrs):
        """Enqueue a trace whose planes are device pointers (dict name -> int address)."""
        cam, camp = self._cam(cam)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_DEVICE, *[ptrs.get(k) for k in _PLANES])
        L.check(L.lib().vrt_trace(self._ctx, camp, w, h, y0, y1, C.byref(o)))

    def bound_trace_device(self, cam, w, h, y0, y1, ptrs):
        """`trace_device` with its arguments converted once: returns a function of no arguments that enqueues the trace
        (a host loop that issues the same launch thousands of times pays the conversion of the camera and of the plane
        table once instead of per call: about 2 instead of 7 microseconds of host time per launch)."""
        cam, camp = self._cam(cam)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_DEVICE, *[ptrs.get(k) for k in _PLANES])
        fn, ctx, ref, check = L.lib().vrt_trace, self._ctx, C.byref(o), L.check
        keep = (cam, o)   # (the closure keeps the converted arguments alive)

        def call(_keep=keep):
            check(fn(ctx, camp, w, h, y0, y1, ref))
        return call

    def bound_trace_strips_device(self, cam, w, h, strip_h, part, parts, ptrs):
        """`trace_strips_device` with its arguments converted once (see bound_trace_device)."""
        cam, camp = self._cam(cam)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_DEVICE, *[ptrs.get(k) for k in _PLANES])
        fn, ctx, ref, check = L.lib().vrt_trace_strips, self._ctx, C.byref(o), L.check
        keep = (cam, o)

        def call(_keep=keep):
            check(fn(ctx, camp, w, h, strip_h, part, parts, ref))
        return call

    def trace_static_device(self, cam, w, h, y0, y1, ptrs):
        """`trace_device` through the entry point that allocates nothing (`vrt_trace_static`): the launch's
        tables and dispatch-order slots live in the library's static device array."""
        cam, camp = self._cam(cam)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_DEVICE, *[ptrs.get(k) for k in _PLANES])
        L.check(L.lib().vrt_trace_static(self._ctx, camp, w, h, y0, y1, C.byref(o)))

    def trace_strips_device(self, cam, w, h, strip_h, part, parts, ptrs):
        cam, camp = self._cam(cam)
        o = L.VrtOutputs(C.sizeof(L.VrtOutputs), L.MEM_DEVICE, *[ptrs.get(k) for k in _PLANES])
        L.check(L.lib().vrt_trace_strips(self._ctx, camp, w, h, strip_h, part, parts, C.byref(o)))

    def assemble_strips(self, gathered, frame, row_bytes, h, strip_h, parts, part_stride_rows):
        """Enqueue the copy of gathered strip buffers (device address; `parts` buffers, each
        `part_stride_rows` packed rows of `row_bytes`) into the `h` rows of `frame` (device address)."""
        L.check(L.lib().vrt_assemble_strips(self._ctx, gathered, frame, row_bytes, h, strip_h, parts,
                                            part_stride_rows))

    def assemble_strips_bgr(self, gathered_bgr, frame_bgra, w, h, strip_h, parts, part_stride_rows):
        """The same for parts that travelled as BGR8 rows (3 * w bytes each): writes the BGRA8 frame, alpha 255."""
        L.check(L.lib().vrt_assemble_strips_bgr(self._ctx, gathered_bgr, frame_bgra, w, h, strip_h, parts, part_stride_rows))

    def pack_bgr(self, bgra_ptr, bgr_ptr, pixels):
        """Enqueue BGRA8 -> BGR8 (device addresses; the frame as it travels between GPUs)."""
        L.check(L.lib().vrt_pack_bgr(self._ctx, bgra_ptr, bgr_ptr, pixels))

    def unpack_bgra(self, bgr_ptr, bgra_ptr, pixels):
        L.check(L.lib().vrt_unpack_bgra(self._ctx, bgr_ptr, bgra_ptr, pixels))

    def sync(self):
        L.check(L.lib().vrt_sync(self._ctx))

    def set_frame_overlap(self, streams):
        """Deal device-plane traces round to `streams` (1..4) streams so that consecutive frames overlap."""
        L.check(L.lib().vrt_set_frame_overlap(self._ctx, int(streams)))

    def fork(self):
        """Every stream of the context waits for what stream() has been given so far."""
        L.check(L.lib().vrt_fork(self._ctx))

    def join(self):
        """The context's first stream (stream()) waits for every trace enqueued so far."""
        L.check(L.lib().vrt_join(self._ctx))

    def kernel_occupancy(self, kind):
        """Workgroups of a trace kernel (0: skip, 1: binned secondary, 2: literal) one compute unit holds at once."""
        n = C.c_int32()
        L.check(L.lib().vrt_kernel_occupancy(self._ctx, int(kind), C.byref(n)))
        return n.value

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        L.check(L.lib().vrt_kernel_time(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_reset(self):
        L.check(L.lib().vrt_kernel_time_reset(self._ctx))

    def kernel_time_sampling(self, every_n):
        """Bracket one trace launch in `every_n` with HIP events (default 1; 0 = none)."""
        L.check(L.lib().vrt_kernel_time_sampling(self._ctx, int(every_n)))

    def last_kernel_name(self):
        return L.lib().vrt_last_kernel_name(self._ctx).decode()

    def stream(self):
        return L.lib().vrt_stream(self._ctx)
