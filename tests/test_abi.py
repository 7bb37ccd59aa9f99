"""The C-ABI library loads and exports every symbol include/vrt.h declares (no GPU needed:
nothing here launches work), and the host-side argument checks reject bad input."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracing_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vrt_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(L.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(L.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_strings():
    lib = L.lib()
    assert b"gfx950" in lib.vrt_version()
    assert isinstance(lib.vrt_last_error(), bytes)


def test_struct_sizes_match_header_layout():
    assert C.sizeof(L.VrtConfig) == 24
    assert C.sizeof(L.VrtOutputs) == 8 + 7 * C.sizeof(C.c_void_p)


def test_argument_validation_without_a_device():
    lib = L.lib()
    ctx = C.c_void_p()
    assert lib.vrt_create(None, C.byref(ctx)) == L.VRT_E_INVALID_ARG
    bad = L.VrtConfig(3, 0, 160, 0, 0, 0)
    assert lib.vrt_create(C.byref(bad), C.byref(ctx)) == L.VRT_E_INVALID_ARG
    assert b"struct_size" in lib.vrt_last_error()
    for n in (0, 4, 100, 8192):
        cfg = L.VrtConfig(C.sizeof(L.VrtConfig), 0, n, 0, 0, 0)
        assert lib.vrt_create(C.byref(cfg), C.byref(ctx)) == L.VRT_E_INVALID_ARG
    cfg = L.VrtConfig(C.sizeof(L.VrtConfig), 0, 160, 7, 0, 0)
    assert lib.vrt_create(C.byref(cfg), C.byref(ctx)) == L.VRT_E_INVALID_ARG
    assert lib.vrt_destroy(None) == L.VRT_OK
    assert lib.vrt_sync(None) == L.VRT_E_INVALID_ARG


def test_strip_rows_partition():
    lib = L.lib()
    for h, sh, parts in [(2160, 16, 8), (1080, 8, 4), (480, 16, 3), (48, 16, 2)]:
        rows = [lib.vrt_strip_rows(h, sh, p, parts) for p in range(parts)]
        assert sum(rows) == h
        assert max(rows) - min(rows) <= sh
    assert lib.vrt_strip_rows(100, 0, 0, 1) == 0 and lib.vrt_strip_rows(100, 16, 2, 2) == 0


def test_no_device_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = L.lib()
    ctx = C.c_void_p()
    cfg = L.VrtConfig(C.sizeof(L.VrtConfig), 0, 160, 0, 0, 0)
    assert lib.vrt_create(C.byref(cfg), C.byref(ctx)) == L.VRT_E_NO_DEVICE
    assert not ctx.value
    from raytracing_amd import GraphicsCore, VrtError
    with pytest.raises(VrtError):
        GraphicsCore()


def test_rust_declarations_name_every_symbol():
    # rust/src/vrt_sys.rs cannot be compiled here; at least its extern block must list the header's functions
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "rust", "src", "vrt_sys.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (vrt_[a-z_0-9]+)\s*\(", text)))
    assert rust == declared_symbols()


def test_selftest_library_loads():
    import ctypes as C
    import os
    from raytracing_amd import _lib as L
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libvrt_selftest.so"))
    assert hasattr(lib, "vrt_selftest_recip")
