"""ctypes binding of the CPU oracle (oracle/libvrt_oracle.so).

Test infrastructure only: imported by tests/, by the smoke check and by the
CPU-baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libvrt_oracle.so")


class OrcWorld(C.Structure):
    _fields_ = [("block", C.c_void_p), ("dist", C.c_void_p), ("voxel_bytes", C.c_int32),
                ("n", C.c_int32), ("max_steps", C.c_uint32)]


class OrcRay(C.Structure):
    _fields_ = [("hit", C.c_int32), ("capped", C.c_int32), ("voxel", C.c_int32 * 3), ("face", C.c_int32),
                ("steps", C.c_uint32), ("block_id", C.c_uint32), ("t", C.c_float),
                ("hit_point", C.c_float * 3), ("normal", C.c_float * 3)]


class OrcOutputs(C.Structure):
    _fields_ = [("bgra8", C.c_void_p), ("voxel", C.c_void_p), ("face", C.c_void_p), ("steps", C.c_void_p),
                ("t", C.c_void_p), ("hit_point", C.c_void_p), ("rgba", C.c_void_p)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", _ORACLE_DIR, "libvrt_oracle.so"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        F3 = C.POINTER(C.c_float)
        L.orc_raygen.argtypes = [F3, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, F3, F3]
        L.orc_raygen.restype = None
        L.orc_trace_ray.argtypes = [C.POINTER(OrcWorld), F3, F3, C.POINTER(OrcRay)]
        L.orc_trace_ray.restype = None
        L.orc_trace.argtypes = [C.POINTER(OrcWorld), F3, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.POINTER(OrcOutputs), C.c_int]
        L.orc_trace.restype = C.c_int
        L.orc_trace_secondary.argtypes = [C.POINTER(OrcWorld), F3, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_trace_secondary.restype = C.c_int
        L.orc_secondary_dir.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, F3]
        L.orc_secondary_dir.restype = None
        L.orc_trace_ray_from.argtypes = [C.POINTER(OrcWorld), F3, F3, F3, C.POINTER(OrcRay)]
        L.orc_trace_ray_from.restype = None
        L.orc_sin.argtypes = [C.c_float]
        L.orc_sin.restype = C.c_float
        L.orc_snoise.argtypes = [C.c_float] * 3
        L.orc_snoise.restype = C.c_float
        L.orc_terrain_fill.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int]
        L.orc_terrain_fill.restype = None
        L.orc_terrain_chunk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        L.orc_terrain_chunk.restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def host_threads():
    """Worker threads for oracle runs: the CPU share of this process, at most 16."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    env = os.environ.get("OMP_NUM_THREADS")
    if env and env.isdigit():
        n = min(n, int(env))
    return max(1, min(n, 16))


class World:
    """Host-side world in the reference image layout (linear, x fastest)."""

    def __init__(self, block, dist=None, max_steps=None):
        block = np.ascontiguousarray(block)
        assert block.ndim == 3 and block.shape[0] == block.shape[1] == block.shape[2]
        assert block.dtype in (np.uint8, np.uint16)
        self.block = block  # indexed [z, y, x]
        self.n = block.shape[0]
        self.dist = None if dist is None else np.ascontiguousarray(dist, dtype=np.uint16)
        self.max_steps = int(max_steps) if max_steps is not None else 4 * self.n
        self._c = OrcWorld(self.block.ctypes.data, None if self.dist is None else self.dist.ctypes.data,
                           self.block.dtype.itemsize, self.n, self.max_steps)

    def c(self):
        return C.byref(self._c)


def raygen(cam, w, h, gx, gy):
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    ro = np.zeros(3, np.float32)
    rd = np.zeros(3, np.float32)
    lib().orc_raygen(_fp(cam), w, h, gx, gy, _fp(ro), _fp(rd))
    return ro, rd


def trace_ray(world, ro, rd):
    ro = np.ascontiguousarray(ro, dtype=np.float32)
    rd = np.ascontiguousarray(rd, dtype=np.float32)
    r = OrcRay()
    lib().orc_trace_ray(world.c(), _fp(ro), _fp(rd), C.byref(r))
    return dict(hit=r.hit, capped=r.capped, voxel=tuple(r.voxel), face=r.face, steps=r.steps,
                block_id=r.block_id, t=np.float32(r.t), hit_point=np.array(r.hit_point, np.float32),
                normal=np.array(r.normal, np.float32))


def trace(world, cam, w, h, y0=0, y1=None, nthreads=None, want=("bgra8", "voxel", "face", "steps", "t", "hit_point", "rgba")):
    """Trace rows [y0,y1) of a w x h frame; returns a dict of numpy planes."""
    y1 = h if y1 is None else y1
    rows = y1 - y0
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    shapes = dict(bgra8=((rows, w, 4), np.uint8), voxel=((rows, w, 3), np.int32), face=((rows, w), np.int32),
                  steps=((rows, w), np.uint32), t=((rows, w), np.float32), hit_point=((rows, w, 3), np.float32),
                  rgba=((rows, w, 4), np.float32))
    out = {k: np.zeros(*shapes[k]) for k in want}
    o = OrcOutputs(*[out[k].ctypes.data if k in out else None for k in
                     ("bgra8", "voxel", "face", "steps", "t", "hit_point", "rgba")])
    rc = lib().orc_trace(world.c(), _fp(cam), w, h, y0, y1, C.byref(o), nthreads or host_threads())
    if rc != 0:
        raise ValueError("orc_trace rejected its arguments")
    return out


def trace_secondary(world, cam, w, h, y0=0, y1=None, nthreads=None):
    y1 = h if y1 is None else y1
    rows = y1 - y0
    cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
    out = {"shadow": np.zeros((rows, w), np.uint8), "ao": np.zeros((rows, w), np.uint8),
           "sec_steps": np.zeros((rows, w), np.uint32)}
    rc = lib().orc_trace_secondary(world.c(), _fp(cam), w, h, y0, y1, out["shadow"].ctypes.data,
                                   out["ao"].ctypes.data, out["sec_steps"].ctypes.data, nthreads or host_threads())
    if rc != 0:
        raise ValueError("orc_trace_secondary rejected its arguments")
    return out


def secondary_dir(pixel_index, index, face):
    rd = np.zeros(3, np.float32)
    lib().orc_secondary_dir(pixel_index, index, face, _fp(rd))
    return rd


def sin(x):
    return np.float32(lib().orc_sin(np.float32(x)))


def snoise(x, y, z):
    return np.float32(lib().orc_snoise(np.float32(x), np.float32(y), np.float32(z)))


def terrain_fill(n, wmin, dtype=np.uint8, nthreads=None):
    """Default scene over the world cube [wmin, wmin+n)^3, stored toroidally; returns block[z,y,x]."""
    block = np.zeros((n, n, n), dtype)
    wm = (C.c_int32 * 3)(*[int(v) for v in wmin])
    lib().orc_terrain_fill(block.ctypes.data, block.dtype.itemsize, n, wm, nthreads or host_threads())
    return block


def terrain_chunk(block, cx, cy, cz):
    lib().orc_terrain_chunk(block.ctypes.data, block.dtype.itemsize, block.shape[0], cx, cy, cz)
