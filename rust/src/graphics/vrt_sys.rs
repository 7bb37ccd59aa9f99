//! UNVERIFIED (never compiled).  Declarations of include/vrt.h, one to one.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct VrtCtx {
    _private: [u8; 0],
}

#[repr(C)]
pub struct VrtComm {
    _private: [u8; 0],
}

#[repr(C)]
pub struct VrtConfig {
    pub struct_size: u32,
    pub device: i32,
    pub grid_n: u32,
    pub voxel_format: u32,
    pub max_steps: u32,
    pub trace_mode: u32,
}

#[repr(C)]
pub struct VrtOutputs {
    pub struct_size: u32,
    pub memory: u32,
    pub bgra8: *mut u8,
    pub voxel: *mut i32,
    pub face: *mut i32,
    pub steps: *mut u32,
    pub t: *mut f32,
    pub hit_point: *mut f32,
    pub rgba: *mut f32,
    pub bgr8: *mut u8,
}

#[repr(C)]
pub struct VrtSecondaryOutputs {
    pub struct_size: u32,
    pub memory: u32,
    pub shadow: *mut u8,
    pub ao: *mut u8,
    pub sec_steps: *mut u32,
}

pub const VRT_OK: c_int = 0;
pub const VRT_VOXEL_REF_U16U16: u32 = 0;
pub const VRT_VOXEL_PACKED_U8: u32 = 1;
pub const VRT_TRACE_AUTO: u32 = 0;
pub const VRT_MEM_HOST: u32 = 0;
pub const VRT_MEM_DEVICE: u32 = 1;

extern "C" {
    pub fn vrt_create(cfg: *const VrtConfig, out: *mut *mut VrtCtx) -> c_int;
    pub fn vrt_destroy(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_upload_world(ctx: *mut VrtCtx, block: *const c_void, dist: *const c_void, n: u32, memory: u32) -> c_int;
    pub fn vrt_generate_chunk(ctx: *mut VrtCtx, cx: i32, cy: i32, cz: i32) -> c_int;
    pub fn vrt_generate_chunks(ctx: *mut VrtCtx, xyz: *const i32, count: u32) -> c_int;
    pub fn vrt_generate_window(ctx: *mut VrtCtx, wmin: *const i32) -> c_int;
    pub fn vrt_reset_distance_field(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_trace(ctx: *mut VrtCtx, cam_col_major: *const f32, w: u32, h: u32, y0: u32, y1: u32, out: *const VrtOutputs) -> c_int;
    pub fn vrt_trace_static(ctx: *mut VrtCtx, cam_col_major: *const f32, w: u32, h: u32, y0: u32, y1: u32, out: *const VrtOutputs) -> c_int;
    pub fn vrt_trace_secondary(ctx: *mut VrtCtx, cam_col_major: *const f32, w: u32, h: u32, y0: u32, y1: u32, out: *const VrtSecondaryOutputs) -> c_int;
    pub fn vrt_trace_strips(ctx: *mut VrtCtx, cam_col_major: *const f32, w: u32, h: u32, strip_h: u32, part: u32, parts: u32, out: *const VrtOutputs) -> c_int;
    pub fn vrt_strip_rows(h: u32, strip_h: u32, part: u32, parts: u32) -> u32;
    pub fn vrt_assemble_strips(ctx: *mut VrtCtx, gathered: *const c_void, frame: *mut c_void, row_bytes: u32, h: u32, strip_h: u32, parts: u32, part_stride_rows: u32) -> c_int;
    pub fn vrt_assemble_strips_bgr(ctx: *mut VrtCtx, gathered_bgr: *const c_void, frame_bgra: *mut c_void, w: u32, h: u32, strip_h: u32, parts: u32, part_stride_rows: u32) -> c_int;
    pub fn vrt_pack_bgr(ctx: *mut VrtCtx, bgra: *const c_void, bgr: *mut c_void, pixels: usize) -> c_int;
    pub fn vrt_unpack_bgra(ctx: *mut VrtCtx, bgr: *const c_void, bgra: *mut c_void, pixels: usize) -> c_int;
    pub fn vrt_alloc_host(ctx: *mut VrtCtx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn vrt_free_host(ctx: *mut VrtCtx, ptr: *mut c_void) -> c_int;
    pub fn vrt_sync(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_set_frame_overlap(ctx: *mut VrtCtx, streams: u32) -> c_int;
    pub fn vrt_join(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_fork(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_download_world(ctx: *mut VrtCtx, block_out: *mut c_void, memory: u32) -> c_int;
    pub fn vrt_download_index(ctx: *mut VrtCtx, cells_out: *mut c_void, pyr_out: *mut c_void, memory: u32) -> c_int;
    pub fn vrt_kernel_time(ctx: *mut VrtCtx, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn vrt_kernel_time_reset(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_kernel_time_sampling(ctx: *mut VrtCtx, every_n: u32) -> c_int;
    pub fn vrt_kernel_occupancy(ctx: *mut VrtCtx, kind: u32, workgroups_per_cu: *mut i32) -> c_int;
    pub fn vrt_last_kernel_name(ctx: *mut VrtCtx) -> *const c_char;
    pub fn vrt_stream(ctx: *mut VrtCtx) -> *mut c_void;
    pub fn vrt_device(ctx: *mut VrtCtx) -> c_int;
    pub fn vrt_comm_create(ctxs: *const *mut VrtCtx, n: u32, out: *mut *mut VrtComm) -> c_int;
    pub fn vrt_comm_destroy(comm: *mut VrtComm) -> c_int;
    pub fn vrt_comm_size(comm: *const VrtComm) -> u32;
    pub fn vrt_gather(comm: *mut VrtComm, send: *const *const c_void, bytes_each: usize, recv_root: *mut c_void) -> c_int;
    pub fn vrt_gather_strips(comm: *mut VrtComm, packed: *const *const c_void, gathered: *mut c_void, frame: *mut c_void, row_bytes: u32, h: u32, strip_h: u32, part_stride_rows: u32) -> c_int;
    pub fn vrt_gather_strips_bgr(comm: *mut VrtComm, packed_bgra: *const *const c_void, packed_bgr: *const *mut c_void, gathered_bgr: *mut c_void, frame_bgra: *mut c_void, w: u32, h: u32, strip_h: u32, part_stride_rows: u32) -> c_int;
    pub fn vrt_last_error() -> *const c_char;
    pub fn vrt_version() -> *const c_char;
}
