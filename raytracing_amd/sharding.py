"""Multi-GPU sharding of one frame: interleaved row strips + framebuffer gather.

Strip s covers image rows [s*strip_h, (s+1)*strip_h) and belongs to rank s % world, so sky
and ground strips are dealt evenly (contiguous bands would give one rank all the long
rays).  Each rank traces its strips packed in strip order (`vrt_trace_strips`); the only
exchange is the gather of the packed buffers to one rank, followed by a row scatter.
The reference has no counterpart (single device, src/graphics/mod.rs:325-342).
"""
import numpy as np
import torch
import torch.distributed as dist


def strip_rows(h, strip_h, part, parts):
    """Image rows traced by `part`, in packed order (matches vrt_trace_strips)."""
    rows = [np.arange(s * strip_h, min(h, (s + 1) * strip_h)) for s in range(part, -(-h // strip_h), parts)]
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def max_strip_rows(h, strip_h, parts):
    return max(len(strip_rows(h, strip_h, p, parts)) for p in range(parts))


def pose_indices(count, part, parts):
    """Poses of a flythrough traced by `part`: pose k belongs to part k % parts (neighbouring poses look
    alike, so dealing them round keeps the parts' work alike too)."""
    return list(range(part, count, parts))


def gather_frame(packed, h, strip_h, dst=0, group=None, out=None, core=None, pad_rows=None, pixels="bgra"):
    """Gather every rank's packed rows (tensor [rows_r, ...]) to `dst` and return the full
    frame [h, ...] there (None elsewhere).  Ranks own different row counts when the strip
    count is not a multiple of the world size, so buffers are padded to the largest.

    With `core` (the GraphicsCore that traced `packed`, device tensors) the gather is ordered
    behind the trace on the context's stream and the rows are put back by one launch of the
    library's copy kernel (`vrt_assemble_strips`), enqueued on the same stream: call
    `core.sync()` before reading the frame.  Without it (host tensors, the gloo rehearsal)
    the rows are put back with tensor indexing.  `pad_rows`: rows every rank's buffer is padded to (default
    and minimum: the rows of the largest part; the same value on every rank).  `pixels="bgr"` (with `core`, for a
    BGRA8 plane [rows, w, 4] with w a multiple of 4): the rows cross the links without their alpha byte, which is 255
    on every path of the shader -- packed by `vrt_pack_bgr` before the gather, restored by the reorder kernel
    (`vrt_assemble_strips_bgr`) behind it: three quarters of the bytes through rank `dst`'s links."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    most = max_strip_rows(h, strip_h, world)
    pad_rows = most if pad_rows is None else int(pad_rows)
    if pad_rows < most:
        raise ValueError(f"pad_rows {pad_rows} is less than the {most} rows of the largest part")

    def padded():
        if packed.shape[0] == pad_rows:
            return packed
        buf = packed.new_zeros((pad_rows,) + tuple(packed.shape[1:]))
        buf[:packed.shape[0]] = packed
        return buf

    if core is not None:
        if not packed.is_cuda:
            raise ValueError("gather_frame(core=...) takes device tensors")
        # Everything that touches `packed` runs on the context's stream, behind the trace that fills it: the
        # padding copy (ranks own different row counts when the strips do not divide evenly), the gather and
        # the row reorder.  Buffers made here are used on that stream only and are recorded on it.
        bgr = pixels == "bgr"
        if bgr and not (packed.dtype == torch.uint8 and packed.dim() == 3 and packed.shape[2] == 4 and packed.shape[1] % 4 == 0):
            raise ValueError('pixels="bgr" takes a uint8 [rows, w, 4] plane with w a multiple of 4')
        # With frames in flight on several streams of the context (vrt_set_frame_overlap) the trace that fills `packed`
        # may run on another of them: the context's first stream waits for every trace enqueued so far.
        core.join()
        stream = torch.cuda.ExternalStream(core.stream(), device=packed.device)
        with torch.cuda.stream(stream):
            buf = padded()
            buf.record_stream(stream)
            if bgr:   # three bytes per pixel from here to the reorder
                w = buf.shape[1]
                buf3 = buf.new_empty((pad_rows, w, 3))
                buf3.record_stream(stream)
                core.pack_bgr(buf.data_ptr(), buf3.data_ptr(), pad_rows * w)
                buf = buf3
            if rank != dst:
                dist.gather(buf, None, dst=dst, group=group)
                return None
            parts = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
            parts.record_stream(stream)
            dist.gather(buf, list(parts.unbind(0)), dst=dst, group=group)
            frame = out if out is not None else packed.new_empty((h,) + tuple(packed.shape[1:]))
            if not frame.is_contiguous() or frame.shape[0] != h:
                raise ValueError("`out` must be a contiguous [h, ...] tensor")
            if bgr:
                core.assemble_strips_bgr(parts.data_ptr(), frame.data_ptr(), frame.shape[1], h, strip_h, world, pad_rows)
            else:
                row_bytes = frame[0].numel() * frame.element_size()
                core.assemble_strips(parts.data_ptr(), frame.data_ptr(), row_bytes, h, strip_h, world, pad_rows)
            frame.record_stream(stream)
        return frame
    buf = padded()
    if rank == dst:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, parts, dst=dst, group=group)
        frame = out if out is not None else packed.new_empty((h,) + tuple(packed.shape[1:]))
        for r in range(world):
            rows = torch.as_tensor(strip_rows(h, strip_h, r, world), device=packed.device)
            frame[rows] = parts[r][:len(rows)]
        return frame
    dist.gather(buf, None, dst=dst, group=group)
    return None


class Comm:
    """Several GPUs driven by this one process: `cores` are GraphicsCore objects on distinct HIP devices, the
    first one the root.  A thin wrapper of vrt_comm_create / vrt_gather / vrt_gather_strips (RCCL send / receive
    enqueued on the cores' own streams); buffers are device addresses."""

    def __init__(self, cores):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        self.cores = list(cores)
        arr = (C.c_void_p * len(self.cores))(*[c._ctx for c in self.cores])
        self._comm = C.c_void_p()
        L.check(L.lib().vrt_comm_create(arr, len(self.cores), C.byref(self._comm)))

    def size(self):
        return self._L.lib().vrt_comm_size(self._comm)

    def gather(self, send_ptrs, bytes_each, recv_root_ptr):
        arr = (self._C.c_void_p * len(send_ptrs))(*send_ptrs)
        self._L.check(self._L.lib().vrt_gather(self._comm, arr, bytes_each, recv_root_ptr))

    def gather_strips(self, packed_ptrs, gathered_ptr, frame_ptr, row_bytes, h, strip_h, part_stride_rows):
        arr = (self._C.c_void_p * len(packed_ptrs))(*packed_ptrs)
        self._L.check(self._L.lib().vrt_gather_strips(self._comm, arr, gathered_ptr, frame_ptr, row_bytes, h, strip_h,
                                                      part_stride_rows))

    def gather_strips_bgr(self, packed_bgra_ptrs, packed_bgr_ptrs, gathered_bgr_ptr, frame_bgra_ptr, w, h, strip_h, part_stride_rows):
        """A strip-sharded BGRA8 frame with three bytes per pixel on the links: every core packs its rows into its own
        BGR8 buffer, the parts are gathered, the root writes the BGRA8 frame in image order."""
        a = (self._C.c_void_p * len(packed_bgra_ptrs))(*packed_bgra_ptrs)
        b = (self._C.c_void_p * len(packed_bgr_ptrs))(*packed_bgr_ptrs)
        self._L.check(self._L.lib().vrt_gather_strips_bgr(self._comm, a, b, gathered_bgr_ptr, frame_bgra_ptr, w, h, strip_h,
                                                          part_stride_rows))

    def close(self):
        if self._comm:
            self._L.lib().vrt_comm_destroy(self._comm)
            self._comm = self._C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
