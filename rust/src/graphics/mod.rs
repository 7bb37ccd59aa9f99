//! UNVERIFIED (never compiled: the build image has no Rust toolchain).
//!
//! Replacement for `src/graphics/mod.rs` of Solvir777/Raytracing with the trace path on libvrt.so instead of
//! vulkano.  Everything `src/main.rs`, `src/game_logic` and `src/terrain` use keeps its name, visibility
//! and signature (reference lines in brackets):
//!
//!   pub const CHUNK_SIZE: u32, CHUNK_SIZE_3: u32                                         [mod.rs:147-148]
//!   pub(crate) fn new() -> (EventLoop<()>, GraphicsCore)                                 [mod.rs:292]
//!   pub fn run(mut self, EventLoop<()>, GameState, fn(&mut GraphicsCore, &mut GameState) -> PushConstants) -> !   [mod.rs:430]
//!   fn redraw(&mut self, PushConstants) -> bool            (true = frame skipped)        [mod.rs:531]
//!   pub fn upload_chunk_gpu(&mut self, Vector3<i32>, &mut GameState)                     [mod.rs:245]
//!   pub(crate) fn reset_distance_field(&mut self)                                        [mod.rs:150]
//!   pub fn get_settings(&self) -> GraphicsSettings                                       [mod.rs:426]
//!   pub(crate) device: Arc<Device>, pub(crate) previous_frame_end: Option<Box<dyn GpuFuture>>   [mod.rs:136,140]
//!   push_constants::PushConstants { transform: Matrix4<f32>, image_index: u32 }          [push_constants.rs:4-9]
//!   graphics_settings::GraphicsSettings { render_distance: u8, fov: f64, mouse_sensitivity: (f32, f32) }   [graphics_settings.rs:2-6]
//!
//! `Device` and `GpuFuture` are the stand-ins of shims/vulkano (a device is something whose queue can be
//! waited for: here `vrt_sync`).  The window and the event routing are winit's, as upstream.  Putting the
//! B8G8R8A8 frame on the screen is not part of the trace path: `Presenter` is the hook for it, and the default
//! one drops the frame.
pub mod vrt_sys;

use crate::game_logic::game_state::GameState;
use nalgebra::Vector3;
use std::ffi::CStr;
use std::os::raw::c_void;
use std::ptr::null_mut;
use std::sync::Arc;
use vulkano::device::Device;
use vulkano::sync::{self, GpuFuture};
use winit::dpi::PhysicalSize;
use winit::event::{DeviceEvent, Event, MouseScrollDelta, VirtualKeyCode, WindowEvent};
use winit::event_loop::{ControlFlow, EventLoop};
use winit::window::{CursorGrabMode, Window, WindowBuilder};

use graphics_settings::GraphicsSettings;
use push_constants::PushConstants;

pub mod push_constants {
    use nalgebra::Matrix4;

    /// The shader's push-constant block (raytrace.glsl:5-8): 64 bytes of column-major mat4, then a u32.
    #[repr(C)]
    pub struct PushConstants {
        pub transform: Matrix4<f32>,
        pub image_index: u32,
    }
}

mod graphics_settings {
    #[derive(Copy, Clone)]
    pub struct GraphicsSettings {
        pub(crate) render_distance: u8,
        #[allow(dead_code)]
        fov: f64, // the reference never reads it either
        pub mouse_sensitivity: (f32, f32),
    }

    impl GraphicsSettings {
        pub fn default() -> Self {
            Self { render_distance: 1, fov: 90.0, mouse_sensitivity: (0.002, 0.002) }
        }
    }
}

/// Receives every finished frame: `width * height * 4` bytes, rows top to bottom, bytes B, G, R, A.
pub trait Presenter {
    fn present(&mut self, window: &Window, frame: &[u8], width: u32, height: u32);
}

struct DropFrames;
impl Presenter for DropFrames {
    fn present(&mut self, _window: &Window, _frame: &[u8], _width: u32, _height: u32) {}
}

/// The context pointer as something a `Send + Sync` closure may hold (the library serialises nothing: one
/// context is driven from one thread, as the reference drives its queue).
struct CtxPtr(*mut vrt_sys::VrtCtx);
unsafe impl Send for CtxPtr {}
unsafe impl Sync for CtxPtr {}

fn last_error() -> String {
    unsafe { CStr::from_ptr(vrt_sys::vrt_last_error()).to_string_lossy().into_owned() }
}

/// The reference unwraps every Vulkan error; so does this.
fn check(rc: i32) {
    assert_eq!(rc, vrt_sys::VRT_OK, "{}", last_error());
}

pub struct GraphicsCore {
    ctx: *mut vrt_sys::VrtCtx,
    settings: GraphicsSettings,
    window: Arc<Window>,
    /// The B8G8R8A8 frame in page-locked host memory of the library (vrt_alloc_host): the copy behind the kernel
    /// runs at the bus rate, 0.23 ms per 1080p frame against 0.54 ms into a pageable Vec.
    frame: *mut u8,
    frame_bytes: usize,
    presenter: Box<dyn Presenter>,
    pub(crate) device: Arc<Device>,
    pub(crate) previous_frame_end: Option<Box<dyn GpuFuture>>,
}

impl GraphicsCore {
    pub const CHUNK_SIZE: u32 = 32;
    pub const CHUNK_SIZE_3: u32 = Self::CHUNK_SIZE * Self::CHUNK_SIZE * Self::CHUNK_SIZE;

    pub(crate) fn new() -> (EventLoop<()>, GraphicsCore) {
        let settings = GraphicsSettings::default();
        let event_loop = EventLoop::new();
        let window = Arc::new(WindowBuilder::new().build(&event_loop).unwrap());

        // the world images: extent 32 * (2 * render_distance + 3) per axis, two r16ui images (mod.rs:354-382)
        let cfg = vrt_sys::VrtConfig {
            struct_size: std::mem::size_of::<vrt_sys::VrtConfig>() as u32,
            device: 0,
            grid_n: Self::CHUNK_SIZE * (2 * settings.render_distance as u32 + 3),
            voxel_format: vrt_sys::VRT_VOXEL_REF_U16U16,
            max_steps: 0,
            trace_mode: vrt_sys::VRT_TRACE_AUTO,
        };
        let mut ctx = null_mut();
        check(unsafe { vrt_sys::vrt_create(&cfg, &mut ctx) });
        // a game loop does not read kernel times: no events around the launches
        check(unsafe { vrt_sys::vrt_kernel_time_sampling(ctx, 0) });

        let queue = CtxPtr(ctx);
        let device = Device::from_wait(Box::new(move || {
            if unsafe { vrt_sys::vrt_sync(queue.0) } == vrt_sys::VRT_OK { Ok(()) } else { Err(last_error()) }
        }));
        let previous_frame_end = Some(sync::now(device.clone()).boxed());
        (
            event_loop,
            GraphicsCore {
                ctx, settings, window, frame: null_mut(), frame_bytes: 0, presenter: Box::new(DropFrames), device, previous_frame_end,
            },
        )
    }

    /// Where finished frames go (not in the reference, which presents through its swapchain).
    pub fn set_presenter(&mut self, presenter: Box<dyn Presenter>) {
        self.presenter = presenter;
    }

    pub fn get_settings(&self) -> GraphicsSettings {
        self.settings
    }

    /// Clears the distance image; the library also brings its traversal index up to date with the chunks
    /// uploaded since the last call (the reference calls this after every batch, game_state.rs:82).
    pub(crate) fn reset_distance_field(&mut self) {
        check(unsafe { vrt_sys::vrt_reset_distance_field(self.ctx) });
        self.previous_frame_end = Some(sync::now(self.device.clone()).boxed());
    }

    /// Runs the terrain generator for the 32^3 voxels of one chunk and stores them at their toroidal texels.
    pub fn upload_chunk_gpu(&mut self, chunk_pos: Vector3<i32>, _game_state: &mut GameState) {
        check(unsafe { vrt_sys::vrt_generate_chunk(self.ctx, chunk_pos.x, chunk_pos.y, chunk_pos.z) });
        self.previous_frame_end = Some(sync::now(self.device.clone()).boxed());
    }

    pub fn run(
        mut self,
        event_loop: EventLoop<()>,
        mut game_state: GameState,
        game_update: fn(&mut GraphicsCore, &mut GameState) -> PushConstants,
    ) -> ! {
        event_loop.run(move |event, _, control_flow| match event {
            Event::WindowEvent { event: window_event, .. } => match window_event {
                WindowEvent::CloseRequested => *control_flow = ControlFlow::Exit,
                WindowEvent::CursorEntered { .. } => {
                    self.window.set_cursor_grab(CursorGrabMode::Confined).expect("cursor grab refused");
                    self.window.set_cursor_visible(false);
                }
                WindowEvent::KeyboardInput { input, .. } => {
                    if input.virtual_keycode == Some(VirtualKeyCode::Escape) {
                        *control_flow = ControlFlow::Exit;
                    } else {
                        game_state.key_states.update(input);
                    }
                }
                WindowEvent::MouseWheel { delta, .. } => {
                    let notches = match delta {
                        MouseScrollDelta::LineDelta(_, y) => y,
                        MouseScrollDelta::PixelDelta(position) => position.y as f32,
                    };
                    game_state.speed_modifier = (game_state.speed_modifier + notches * 0.3).clamp(0.1, 100.);
                }
                // a resize needs nothing here: redraw sizes its frame from the window every time
                _ => (),
            },
            Event::DeviceEvent { event: DeviceEvent::MouseMotion { delta }, .. } => {
                game_state.player_look_dir(delta, self.settings.mouse_sensitivity);
            }
            Event::RedrawEventsCleared => {
                let push_constants = game_update(&mut self, &mut game_state);
                if !self.redraw(push_constants) {
                    game_state.key_states.reset();
                }
            }
            _ => (),
        })
    }

    /// One compute pass over the window's pixels.  Returns true when the frame was skipped (a zero-sized
    /// window), as the reference does.
    fn redraw(&mut self, push_constants: PushConstants) -> bool {
        let PhysicalSize { width, height } = self.window.inner_size();
        if width == 0 || height == 0 {
            return true;
        }
        let bytes = width as usize * height as usize * 4;
        if self.frame_bytes != bytes {
            // a new window size: the old frame goes back to the library (which waits for copies into it first)
            check(unsafe { vrt_sys::vrt_free_host(self.ctx, self.frame as *mut c_void) });
            let mut fresh: *mut c_void = null_mut();
            check(unsafe { vrt_sys::vrt_alloc_host(self.ctx, bytes, &mut fresh) });
            self.frame = fresh as *mut u8;
            self.frame_bytes = bytes;
        }
        let out = vrt_sys::VrtOutputs {
            struct_size: std::mem::size_of::<vrt_sys::VrtOutputs>() as u32,
            memory: vrt_sys::VRT_MEM_HOST,
            bgra8: self.frame,
            voxel: null_mut(),
            face: null_mut(),
            steps: null_mut(),
            t: null_mut(),
            hit_point: null_mut(),
            rgba: null_mut(),
            bgr8: null_mut(),
        };
        // nalgebra's Matrix4<f32> is column major: as_slice() is the 16 floats the library takes
        check(unsafe {
            vrt_sys::vrt_trace(self.ctx, push_constants.transform.as_slice().as_ptr(), width, height, 0, height, &out)
        });
        self.previous_frame_end = Some(sync::now(self.device.clone()).boxed());
        // (HOST planes: vrt_trace has returned with the frame complete)
        let frame = unsafe { std::slice::from_raw_parts(self.frame, bytes) };
        self.presenter.present(&self.window, frame, width, height);
        false
    }
}

impl Drop for GraphicsCore {
    fn drop(&mut self) {
        unsafe {
            vrt_sys::vrt_free_host(self.ctx, self.frame as *mut c_void);
            vrt_sys::vrt_destroy(self.ctx);
        }
    }
}
