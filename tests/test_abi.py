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
    assert C.sizeof(L.VrtOutputs) == 8 + 8 * C.sizeof(C.c_void_p)


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
    assert lib.vrt_set_frame_overlap(None, 2) == L.VRT_E_INVALID_ARG
    assert lib.vrt_join(None) == L.VRT_E_INVALID_ARG and lib.vrt_fork(None) == L.VRT_E_INVALID_ARG
    assert lib.vrt_pack_bgr(None, None, None, 4) == L.VRT_E_INVALID_ARG
    assert lib.vrt_generate_chunks(None, None, 0) == L.VRT_E_INVALID_ARG
    assert lib.vrt_download_index(None, None, None, 0) == L.VRT_E_INVALID_ARG


def test_communicator_argument_validation_without_a_device():
    # every check that precedes RCCL: null arguments, rank counts, null contexts, null communicators
    lib = L.lib()
    comm = C.c_void_p()
    assert lib.vrt_comm_create(None, 1, C.byref(comm)) == L.VRT_E_INVALID_ARG
    one = (C.c_void_p * 1)(None)
    assert lib.vrt_comm_create(one, 1, None) == L.VRT_E_INVALID_ARG
    assert lib.vrt_comm_create(one, 0, C.byref(comm)) == L.VRT_E_INVALID_ARG
    assert lib.vrt_comm_create(one, 65, C.byref(comm)) == L.VRT_E_INVALID_ARG
    assert lib.vrt_comm_create(one, 1, C.byref(comm)) == L.VRT_E_INVALID_ARG and b"context 0 is null" in lib.vrt_last_error()
    assert not comm.value
    assert lib.vrt_comm_destroy(None) == L.VRT_OK
    assert lib.vrt_comm_size(None) == 0
    assert lib.vrt_gather(None, one, 16, None) == L.VRT_E_INVALID_ARG
    assert lib.vrt_gather_strips(None, one, None, None, 16, 16, 8, 16) == L.VRT_E_INVALID_ARG
    assert lib.vrt_device(None) == -1


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
    # rust/src/graphics/vrt_sys.rs cannot be compiled here; at least its extern block must list the header's functions
    text = open(os.path.join(ROOT, "rust", "src", "graphics", "vrt_sys.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (vrt_[a-z_0-9]+)\s*\(", text)))
    assert rust == declared_symbols()


# What main.rs, game_logic and terrain of the reference use of `GraphicsCore`, as regular expressions over source
# text with the whitespace squeezed: (pattern, file of the reference it is taken from).
REFERENCE_SURFACE = [
    (r"pub const CHUNK_SIZE: u32 = 32;", "src/graphics/mod.rs"),                                            # :147
    (r"pub const CHUNK_SIZE_3: u32 = Self::CHUNK_SIZE \* Self::CHUNK_SIZE \* Self::CHUNK_SIZE;", "src/graphics/mod.rs"),  # :148
    (r"pub\(crate\) fn reset_distance_field\(&mut self\)", "src/graphics/mod.rs"),                            # :150
    (r"pub fn upload_chunk_gpu\(&mut self, chunk_pos: Vector3<i32>, _?game_state: &mut GameState\)", "src/graphics/mod.rs"),  # :245
    (r"pub\(crate\) fn new\(\) -> \(EventLoop<\(\)>, GraphicsCore\)", "src/graphics/mod.rs"),                   # :292
    (r"pub fn get_settings\(&self\) -> GraphicsSettings", "src/graphics/mod.rs"),                             # :426
    (r"pub fn run\( ?mut self, event_loop: EventLoop<\(\)>, mut game_state: GameState, "
     r"game_update: fn\(&mut GraphicsCore, &mut GameState\) -> PushConstants,? ?\) -> !", "src/graphics/mod.rs"),  # :430
    (r"fn redraw\(&mut self, (mut )?push_constants: PushConstants\) -> bool", "src/graphics/mod.rs"),         # :531
    (r"pub\(crate\) device: Arc<Device>,", "src/graphics/mod.rs"),                                            # :136
    (r"pub\(crate\) previous_frame_end: Option<Box<dyn GpuFuture>>,", "src/graphics/mod.rs"),                 # :140
    (r"pub struct PushConstants \{ pub transform: Matrix4<f32>, pub image_index: u32, \}", "src/graphics/push_constants.rs"),  # :6-9
    (r"pub struct GraphicsSettings \{ pub\(crate\) render_distance: u8, (#\[allow\(dead_code\)\] )?fov: f64,( //[^\n]*?)? "
     r"pub mouse_sensitivity: \(f32, f32\), \}", "src/graphics/graphics_settings.rs"),                         # :2-6
    (r"pub fn default\(\) -> Self", "src/graphics/graphics_settings.rs"),                                     # :9
]


def squeeze(text):
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", " ", text)


def test_rust_graphics_module_keeps_the_reference_signatures():
    """rust/src/graphics/mod.rs (never compiled: no toolchain) declares every item the reference's callers use with
    the reference's visibility, argument list and return type; where a checkout of the reference is present the
    same patterns are checked against its own source, so that the list cannot drift from what it mirrors."""
    ours = squeeze(open(os.path.join(ROOT, "rust", "src", "graphics", "mod.rs")).read())
    for pattern, _ in REFERENCE_SURFACE:
        assert re.search(pattern, ours), pattern
    ref_root = "/root/reference"
    if os.path.isdir(ref_root):
        for pattern, rel in REFERENCE_SURFACE:
            theirs = squeeze(open(os.path.join(ref_root, rel)).read())
            assert re.search(pattern, theirs), (pattern, rel)
    # the game loop's other lines that name vulkano compile against the stand-in crate
    shim = open(os.path.join(ROOT, "rust", "shims", "vulkano", "src", "lib.rs")).read()
    for name in ("pub trait GpuFuture", "fn then_signal_fence_and_flush", "fn boxed", "pub fn now", "pub fn wait", "pub trait Pipeline"):
        assert name in shim, name


def integration_snippet():
    """The Rust block of INTEGRATION.md section 3."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = md[md.index("## 3. `GraphicsCore` keeps its signatures"):md.index("## 4. ")]
    blocks = re.findall(r"```rust\n(.*?)```", section, re.S)
    assert len(blocks) == 1
    return blocks[0]


def test_integration_text_is_the_rust_module():
    """One text for the Rust boundary: the snippet a maintainer reads in INTEGRATION.md is rust/src/graphics/mod.rs with
    the event loop's body elided -- every line of it is a line of that file, in order -- and it matches each of the
    reference's signatures by itself."""
    snippet = integration_snippet()
    module = open(os.path.join(ROOT, "rust", "src", "graphics", "mod.rs")).read().split("\n")
    at = 0
    for line in snippet.split("\n"):
        if not line.strip() or line.strip().startswith("// ...") or line.strip().startswith("//     "):
            continue
        while at < len(module) and module[at] != line:
            at += 1
        assert at < len(module), f"INTEGRATION.md line not in rust/src/graphics/mod.rs (or out of order): {line!r}"
        at += 1
    squeezed = squeeze(snippet)
    for pattern, _ in REFERENCE_SURFACE:
        assert re.search(pattern, squeezed), pattern
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "FrameToken" not in md and "fresh_frame_token" not in md


def test_selftest_library_loads():
    import ctypes as C
    import os
    from raytracing_amd import _lib as L
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(L.__file__)), "libvrt_selftest.so"))
    for name in ("vrt_selftest_recip", "vrt_selftest_valu_rate", "vrt_selftest_kernarg", "vrt_selftest_normalize",
                 "vrt_selftest_unorm8", "vrt_selftest_floor_convert"):
        assert hasattr(lib, name), name
