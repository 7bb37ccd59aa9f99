//! UNVERIFIED (never compiled).
//!
//! The handful of vulkano names that code OUTSIDE src/graphics uses in Solvir777/Raytracing:
//!   src/game_logic/game_state.rs:3     use vulkano::sync::GpuFuture;
//!   src/game_logic/game_state.rs:77    previous_frame_end.take().unwrap().then_signal_fence_and_flush().unwrap().wait(None).unwrap()
//!   src/game_logic/game_state.rs:79    Some(vulkano::sync::now(core.device.clone()).boxed())
//!   src/terrain/mod.rs:5               use vulkano::pipeline::Pipeline;
//! Here a "device" is whatever can wait for the work queued on it: the graphics module builds one around
//! `vrt_sync`, and every future made from it completes when that wait returns.
use std::sync::Arc;

pub mod device {
    /// What `GraphicsCore::device` holds (the reference: `Arc<vulkano::device::Device>`).
    pub struct Device {
        wait_idle: Box<dyn Fn() -> Result<(), String> + Send + Sync>,
    }

    impl Device {
        pub fn from_wait(wait_idle: Box<dyn Fn() -> Result<(), String> + Send + Sync>) -> std::sync::Arc<Device> {
            std::sync::Arc::new(Device { wait_idle })
        }
        pub fn wait_idle(&self) -> Result<(), String> {
            (self.wait_idle)()
        }
    }
}

pub mod pipeline {
    /// Imported, never used, by src/terrain/mod.rs:5.
    pub trait Pipeline {}
}

pub mod sync {
    use super::device::Device;
    use super::Arc;
    use std::time::Duration;

    #[derive(Debug)]
    pub struct FlushError(pub String);

    pub trait GpuFuture {
        /// Blocks until everything submitted before this future has run.
        fn wait_for_queue(&self) -> Result<(), FlushError>;

        fn then_signal_fence_and_flush(self) -> Result<FenceSignalFuture<Self>, FlushError>
        where
            Self: Sized,
        {
            Ok(FenceSignalFuture { previous: self })
        }

        fn boxed(self) -> Box<dyn GpuFuture>
        where
            Self: Sized + 'static,
        {
            Box::new(self)
        }
    }

    impl GpuFuture for Box<dyn GpuFuture> {
        fn wait_for_queue(&self) -> Result<(), FlushError> {
            (**self).wait_for_queue()
        }
    }

    pub struct FenceSignalFuture<F: GpuFuture> {
        previous: F,
    }

    impl<F: GpuFuture> FenceSignalFuture<F> {
        pub fn wait(&self, _timeout: Option<Duration>) -> Result<(), FlushError> {
            self.previous.wait_for_queue()
        }
    }

    pub struct NowFuture {
        device: Arc<Device>,
    }

    /// A future that is complete once the device's queue has drained.
    pub fn now(device: Arc<Device>) -> NowFuture {
        NowFuture { device }
    }

    impl GpuFuture for NowFuture {
        fn wait_for_queue(&self) -> Result<(), FlushError> {
            self.device.wait_idle().map_err(FlushError)
        }
    }
}
