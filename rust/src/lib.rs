//! UNVERIFIED (never compiled: the build image has no Rust toolchain).
//!
//! `GraphicsCore` with the names `main.rs` and `game_logic` of the reference use
//! (src/graphics/mod.rs:147-148, 150, 245, 292, 426, 430, 531; src/game_logic/game_state.rs:50-83),
//! backed by libvrt.so instead of vulkano.  Window creation, input and presentation stay with the
//! caller: `redraw` returns the B8G8R8A8 frame.  The verified counterpart of this file is the C++
//! mirror include/graphics_core.hpp, which tests/test_gpu_cpp_host.py checks against the oracle.
pub mod vrt_sys;

use nalgebra::{Matrix4, Vector3};
use std::ffi::CStr;
use std::ptr::null_mut;

fn last_error() -> String {
    unsafe { CStr::from_ptr(vrt_sys::vrt_last_error()).to_string_lossy().into_owned() }
}

fn check(rc: i32) {
    // the reference unwraps every Vulkan error
    assert_eq!(rc, vrt_sys::VRT_OK, "{}", last_error());
}

/// src/graphics/push_constants.rs:4-9
#[repr(C)]
#[derive(Clone, Copy)]
pub struct PushConstants {
    pub transform: Matrix4<f32>, // column major, like GLSL's mat4
    pub image_index: u32,
}

/// src/graphics/mod.rs:27-33
#[derive(Clone, Copy)]
pub struct GraphicsSettings {
    pub fov: f32,
    pub render_distance: u8,
}

/// Stands in for `Box<dyn GpuFuture>` where game_state.rs:77 writes
/// `core.previous_frame_end.take().unwrap().then_signal_fence_and_flush().unwrap().wait(None).unwrap()`.
pub struct FrameToken(*mut vrt_sys::VrtCtx);
impl FrameToken {
    pub fn then_signal_fence_and_flush(self) -> Result<Self, ()> {
        Ok(self)
    }
    pub fn wait(&self, _timeout: Option<std::time::Duration>) -> Result<(), ()> {
        if unsafe { vrt_sys::vrt_sync(self.0) } == vrt_sys::VRT_OK { Ok(()) } else { Err(()) }
    }
}

/// Stands in for `Arc<Device>` (game_state.rs:79 only clones it to make a fresh future).
#[derive(Clone)]
pub struct DeviceHandle;

pub struct GraphicsCore {
    ctx: *mut vrt_sys::VrtCtx,
    settings: GraphicsSettings,
    width: u32,
    height: u32,
    frame: Vec<u8>,
    pub previous_frame_end: Option<FrameToken>,
    pub device: DeviceHandle,
}

impl GraphicsCore {
    pub const CHUNK_SIZE: u32 = 32;
    pub const CHUNK_SIZE_3: u32 = 32 * 32 * 32;

    /// src/graphics/mod.rs:292 (without the event loop, which the caller keeps)
    pub fn new(width: u32, height: u32) -> GraphicsCore {
        let settings = GraphicsSettings { fov: 90.0, render_distance: 1 };
        let grid_n = Self::CHUNK_SIZE * (2 * settings.render_distance as u32 + 3); // mod.rs:359
        let cfg = vrt_sys::VrtConfig {
            struct_size: std::mem::size_of::<vrt_sys::VrtConfig>() as u32,
            device: 0,
            grid_n,
            voxel_format: vrt_sys::VRT_VOXEL_REF_U16U16,
            max_steps: 0,
            trace_mode: vrt_sys::VRT_TRACE_AUTO,
        };
        let mut ctx = null_mut();
        check(unsafe { vrt_sys::vrt_create(&cfg, &mut ctx) });
        // a game loop does not read kernel times: no events around the launches
        check(unsafe { vrt_sys::vrt_kernel_time_sampling(ctx, 0) });
        GraphicsCore {
            ctx,
            settings,
            width,
            height,
            frame: vec![0u8; (width * height * 4) as usize],
            previous_frame_end: Some(FrameToken(ctx)),
            device: DeviceHandle,
        }
    }

    pub fn get_settings(&self) -> GraphicsSettings {
        self.settings
    }

    /// src/graphics/mod.rs:245 -- the terrain generator for one chunk, stored at its toroidal texels
    /// (the second argument is the reference's `&mut GameState`, which it does not use either)
    pub fn upload_chunk_gpu<G>(&mut self, chunk_pos: Vector3<i32>, _game_state: &mut G) {
        check(unsafe { vrt_sys::vrt_generate_chunk(self.ctx, chunk_pos.x, chunk_pos.y, chunk_pos.z) });
    }

    /// src/graphics/mod.rs:150
    pub fn reset_distance_field(&mut self) {
        check(unsafe { vrt_sys::vrt_reset_distance_field(self.ctx) });
        self.previous_frame_end = Some(FrameToken(self.ctx));
    }

    /// What game_state.rs:79 builds with `vulkano::sync::now(core.device.clone()).boxed()`: that line names
    /// vulkano itself and is the one line of game_logic that has to change, to
    /// `core.previous_frame_end = Some(core.fresh_frame_token());`
    pub fn fresh_frame_token(&self) -> FrameToken {
        FrameToken(self.ctx)
    }

    /// src/graphics/mod.rs:531 -- returns the frame (rows top to bottom, bytes B, G, R, A)
    pub fn redraw(&mut self, push_constants: PushConstants) -> &[u8] {
        let out = vrt_sys::VrtOutputs {
            struct_size: std::mem::size_of::<vrt_sys::VrtOutputs>() as u32,
            memory: vrt_sys::VRT_MEM_HOST,
            bgra8: self.frame.as_mut_ptr(),
            voxel: null_mut(),
            face: null_mut(),
            steps: null_mut(),
            t: null_mut(),
            hit_point: null_mut(),
            rgba: null_mut(),
        };
        // nalgebra's Matrix4<f32> is column major: as_slice() is the 16 floats the ABI wants
        check(unsafe {
            vrt_sys::vrt_trace(self.ctx, push_constants.transform.as_slice().as_ptr(), self.width, self.height, 0,
                               self.height, &out)
        });
        self.previous_frame_end = Some(FrameToken(self.ctx));
        &self.frame
    }
}

impl Drop for GraphicsCore {
    fn drop(&mut self) {
        unsafe { vrt_sys::vrt_destroy(self.ctx) };
    }
}
