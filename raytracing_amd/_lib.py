"""ctypes binding of libvrt.so -- the C ABI declared in include/vrt.h.

The library is built in-tree (raytracing_amd/libvrt.so) by `make -C raytracing_amd/csrc`
or `__graft_entry__.build()`.  There is no fallback: if the shared object is missing or
cannot be loaded, importing a symbol from it raises.
"""
import ctypes as C
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VRT_LIB", os.path.join(_DIR, "libvrt.so"))  # VRT_LIB: another build of the library, for A/B runs

VRT_OK = 0
VRT_E_INVALID_ARG = -1
VRT_E_NO_DEVICE = -2
VRT_E_OUT_OF_MEMORY = -3
VRT_E_UNSUPPORTED = -4
VRT_E_HIP = -5
VRT_E_NO_WORLD = -6

VOXEL_REF_U16U16 = 0
VOXEL_PACKED_U8 = 1
TRACE_AUTO = 0
TRACE_LITERAL = 1
TRACE_SKIP = 2
MEM_HOST = 0
MEM_DEVICE = 1
CHUNK_SIZE = 32


class VrtConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("grid_n", C.c_uint32),
                ("voxel_format", C.c_uint32), ("max_steps", C.c_uint32), ("trace_mode", C.c_uint32)]


class VrtOutputs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("memory", C.c_uint32), ("bgra8", C.c_void_p), ("voxel", C.c_void_p),
                ("face", C.c_void_p), ("steps", C.c_void_p), ("t", C.c_void_p), ("hit_point", C.c_void_p),
                ("rgba", C.c_void_p), ("bgr8", C.c_void_p)]


class VrtSecondaryOutputs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("memory", C.c_uint32), ("shadow", C.c_void_p), ("ao", C.c_void_p),
                ("sec_steps", C.c_void_p)]


# every symbol include/vrt.h declares: name -> (restype, argtypes)
_F = C.POINTER(C.c_float)
_CTX = C.c_void_p
SYMBOLS = {
    "vrt_create": (C.c_int, [C.POINTER(VrtConfig), C.POINTER(_CTX)]),
    "vrt_destroy": (C.c_int, [_CTX]),
    "vrt_upload_world": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    "vrt_generate_chunk": (C.c_int, [_CTX, C.c_int32, C.c_int32, C.c_int32]),
    "vrt_generate_chunks": (C.c_int, [_CTX, C.POINTER(C.c_int32), C.c_uint32]),
    "vrt_generate_window": (C.c_int, [_CTX, C.POINTER(C.c_int32)]),
    "vrt_reset_distance_field": (C.c_int, [_CTX]),
    "vrt_trace": (C.c_int, [_CTX, _F, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(VrtOutputs)]),
    "vrt_trace_static": (C.c_int, [_CTX, _F, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(VrtOutputs)]),
    "vrt_trace_secondary": (C.c_int, [_CTX, _F, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(VrtSecondaryOutputs)]),
    "vrt_trace_strips": (C.c_int, [_CTX, _F, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.POINTER(VrtOutputs)]),
    "vrt_strip_rows": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vrt_assemble_strips": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_uint32]),
    "vrt_assemble_strips_bgr": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vrt_pack_bgr": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vrt_unpack_bgra": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_size_t]),
    "vrt_alloc_host": (C.c_int, [_CTX, C.c_size_t, C.POINTER(C.c_void_p)]),
    "vrt_free_host": (C.c_int, [_CTX, C.c_void_p]),
    "vrt_sync": (C.c_int, [_CTX]),
    "vrt_set_frame_overlap": (C.c_int, [_CTX, C.c_uint32]),
    "vrt_join": (C.c_int, [_CTX]),
    "vrt_fork": (C.c_int, [_CTX]),
    "vrt_download_world": (C.c_int, [_CTX, C.c_void_p, C.c_uint32]),
    "vrt_download_index": (C.c_int, [_CTX, C.c_void_p, C.c_void_p, C.c_uint32]),
    "vrt_kernel_time": (C.c_int, [_CTX, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "vrt_kernel_time_reset": (C.c_int, [_CTX]),
    "vrt_kernel_time_sampling": (C.c_int, [_CTX, C.c_uint32]),
    "vrt_kernel_occupancy": (C.c_int, [_CTX, C.c_uint32, C.POINTER(C.c_int32)]),
    "vrt_last_kernel_name": (C.c_char_p, [_CTX]),
    "vrt_stream": (C.c_void_p, [_CTX]),
    "vrt_device": (C.c_int, [_CTX]),
    "vrt_comm_create": (C.c_int, [C.POINTER(_CTX), C.c_uint32, C.POINTER(C.c_void_p)]),
    "vrt_comm_destroy": (C.c_int, [C.c_void_p]),
    "vrt_comm_size": (C.c_uint32, [C.c_void_p]),
    "vrt_gather": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]),
    "vrt_gather_strips": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_uint32]),
    "vrt_gather_strips_bgr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                        C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "vrt_last_error": (C.c_char_p, []),
    "vrt_version": (C.c_char_p, []),
}

_lib = None


class VrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"vrt error {code}: {message}")
        self.code = code


def lib():
    """Load libvrt.so (once) and type its entry points."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` "
                              "(there is no CPU fallback for the trace path)")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same soname as
        # the system one).  Loading torch first makes libvrt.so bind to that copy instead of
        # bringing in a second runtime from /opt/rocm, which leaves torch unable to see the GPU.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != VRT_OK:
        raise VrtError(rc, lib().vrt_last_error().decode())
    return rc
