//! render_gpu.rs — `raytracer/src/render_gpu.rs`: the `gpu` branch of `RaytracerState::new`, replacing the worker pool of
//! main.rs:1326-1409 with one thread that drives librtiow_hip.so.
//!
//! UNVERIFIED SOURCE (no rustc in the build image).  integration/c_host.c makes the same calls in the same order from C and
//! is run on the GPU box by tests/test_gpu_c_host.py.
//!
//! In `RaytracerState::new`, after `scene_*()` returned `(world, lights)` (main.rs:1301-1312) and the sampler was built (:1324):
//!
//! ```ignore
//! let workers = match GpuRenderer::try_new(&world, &lights, &s, &params, &cam) {
//!     Ok(gpu) => vec![gpu.spawn(tx.clone(), Arc::clone(&cancel_token), Arc::clone(&workblocks_done), total_workblocks)],
//!     Err(why) => { eprintln!("GPU backend not used: {:?}", why); /* the thread pool of :1326-1409, unchanged */ }
//! };
//! ```
use crate::flatten::{FlatError, SceneBuilder};
use crate::gpu_ffi::*;
use std::os::raw::c_void;
use std::sync::atomic::{AtomicBool, AtomicI32, Ordering};
use std::sync::Arc;

#[derive(Debug)]
pub enum GpuError {
    NoDevice,
    Flatten(FlatError),
    Library(String),
}

/// One context per GPU.  `ctxs[0]` is the root: its accumulation buffer receives the RCCL reduce.
pub struct GpuRenderer {
    group: *mut rt_group,
    ctxs: Vec<*mut rt_ctx>,
    sums: Vec<*mut f32>, // W*H*4 f32 on each device (rt_device_alloc)
    params: rt_render_params,
    seed: u32,
}
unsafe impl Send for GpuRenderer {} // moved into the render thread once; a context is driven by one thread at a time

impl GpuRenderer {
    pub fn try_new(
        world: &crate::hittable_list::HittableList,
        lights: &crate::hittable_list::HittableList,
        sampler: &crate::sampling::MultiJitteredSampler,
        params: &crate::RaytracerParams,
        cam: &crate::camera::Camera,
    ) -> Result<GpuRenderer, GpuError> {
        let n_gpus = unsafe { rt_device_count() };
        if n_gpus < 1 {
            return Err(GpuError::NoDevice);
        }
        // (world, lights) -> flat arrays + BVH (flatten.rs); an object without a flatten sends the host back to its thread pool
        let flat = SceneBuilder::new().finish(world, lights, false).map_err(GpuError::Flatten)?;

        let ids: Vec<i32> = (0..n_gpus).collect();
        let mut group: *mut rt_group = std::ptr::null_mut();
        // one GPU: no communicator is needed, but the group API is the same (rt_group_reduce_to_root with n = 1 is a no-op)
        lib(std::ptr::null(), unsafe { rt_group_create(ids.as_ptr(), n_gpus, &mut group) })?;
        let ctxs: Vec<*mut rt_ctx> = (0..n_gpus).map(|i| unsafe { rt_group_ctx(group, i) }).collect();

        // SamplerBase keeps Vec<Vec2> (sampling.rs:15-18); Vec2 is #[repr(C)] { x, y } (math/src/vec2.rs:8), i.e. pairs of f32
        let smp = rt_sampler_desc {
            sets: sampler.sets(),
            samples_in_set: sampler.samples_in_set(),
            samples: sampler.samples().as_ptr() as *const f32,
            disk_samples: sampler.disk_samples().as_ptr() as *const f32,
            shuffled_indices: sampler.shuffled_indices().as_ptr(),
        };
        let n_bytes = (params.image_width as usize) * (params.image_height as usize) * 16;
        let mut sums = Vec::new();
        for &ctx in &ctxs {
            lib(ctx, unsafe { rt_scene_upload(ctx, &flat.desc) })?;
            lib(ctx, unsafe { rt_sampler_upload(ctx, &smp) })?;
            let mut p: *mut c_void = std::ptr::null_mut();
            lib(ctx, unsafe { rt_device_alloc(ctx, n_bytes, &mut p) })?;
            lib(ctx, unsafe { rt_device_zero(ctx, p, n_bytes, std::ptr::null_mut()) })?;
            sums.push(p as *mut f32);
        }
        drop(flat); // the library copied what it keeps

        let seed = rand::random::<u32>(); // the reference is unseeded (thread_rng); the backend's counter RNG wants a key
        let rp = rt_render_params {
            struct_size: std::mem::size_of::<rt_render_params>() as u32,
            image_width: params.image_width,
            image_height: params.image_height,
            samples_per_pixel: params.samples_per_pixel,
            spp_begin: 0,
            spp_count: 0,
            max_ray_depth: params.max_ray_depth,
            background: params.background,
            seed,
            tile_x0: 0,
            tile_y0: 0,
            tile_x1: params.image_width,
            tile_y1: params.image_height,
            camera: cam.to_ffi(), // camera.rs:25-36 field for field; `fn to_ffi(&self) -> rt_camera` goes into camera.rs (private fields)
            out_location: RT_MEM_DEVICE,
            stream: std::ptr::null_mut(),
            cancel: std::ptr::null(),
            collect_stats: 0,
            pixel_rays: std::ptr::null_mut(),
            projection: params.projection as i32, // Perspective, Orthographic, FishEye, SphericalPanoramic = 0..3 (main.rs:1140-1146)
            psi_max: params.psi_max,
            lambda_max: params.lambda_max,
        };
        Ok(GpuRenderer { group, ctxs, sums, params: rp, seed })
    }

    /// The render thread: progressive batches of samples; after each batch the frame so far goes to the viewer through the
    /// same channel the workers use (main.rs:1390-1395), so `RaytracerState::raytrace` (:1436-1467) needs no change.
    pub fn spawn(
        self,
        tx: std::sync::mpsc::Sender<crate::RaytracedPixel>,
        cancel_token: Arc<AtomicBool>,
        workblocks_done: Arc<AtomicI32>,
        total_workblocks: u32,
    ) -> std::thread::JoinHandle<()> {
        std::thread::spawn(move || {
            let (w, h, spp) = (self.params.image_width, self.params.image_height, self.params.samples_per_pixel);
            let n_px = (w * h) as usize;
            let n_gpus = self.ctxs.len() as i32;
            let batch = 16 * n_gpus; // samples per progressive update, split over the GPUs
            let mut rgba = vec![0f32; n_px * 4];
            let mut shown: *mut c_void = std::ptr::null_mut(); // resolved image on the root device
            if unsafe { rt_device_alloc(self.ctxs[0], n_px * 16, &mut shown) } != RT_OK {
                return;
            }
            let mut done = 0;
            while done < spp && !cancel_token.load(Ordering::SeqCst) {
                let n = batch.min(spp - done);
                // GPU g renders samples [done + g*n/G, done + (g+1)*n/G): the RNG is keyed by the absolute sample index, so the
                // frame does not depend on the number of GPUs.  One host thread per context (rt_render feeds its pipeline
                // from the calling thread for the length of the render).
                let status: Vec<i32> = std::thread::scope(|sc| {
                    let hs: Vec<_> = (0..n_gpus).map(|g| {
                        let (ctx, sum, mut p) = (SendPtr(self.ctxs[g as usize]), SendPtr(self.sums[g as usize]), self.params);
                        p.spp_begin = done + g * n / n_gpus;
                        p.spp_count = done + (g + 1) * n / n_gpus - p.spp_begin;
                        sc.spawn(move || if p.spp_count > 0 { unsafe { rt_render(ctx.0, &p, sum.0, std::ptr::null_mut()) } } else { RT_OK })
                    }).collect();
                    hs.into_iter().map(|h| h.join().unwrap_or(RT_E_HIP)).collect()
                });
                if status.iter().any(|&s| s != RT_OK) {
                    eprintln!("rt_render: {:?}", status);
                    break;
                }
                done += n;
                // the sum of all GPUs' buffers lands on GPU 0 (one RCCL reduce over xGMI), the others start again from zero
                if n_gpus > 1 {
                    let ptrs: Vec<*mut f32> = self.sums.clone();
                    if unsafe { rt_group_reduce_to_root(self.group, ptrs.as_ptr(), (n_px * 4) as i64, 0, std::ptr::null()) } != RT_OK {
                        break;
                    }
                    for g in 1..n_gpus as usize {
                        unsafe { rt_device_zero(self.ctxs[g], self.sums[g] as *mut c_void, n_px * 16, std::ptr::null_mut()) };
                    }
                }
                // gamma / validity pass of main.rs:1363-1388 with the samples so far, then the copy the viewer reads
                unsafe {
                    rt_resolve(self.ctxs[0], self.sums[0], n_px as i32, done, shown as *mut f32, RT_MEM_DEVICE, std::ptr::null_mut());
                    rt_device_read(self.ctxs[0], rgba.as_mut_ptr() as *mut c_void, shown, n_px * 16, std::ptr::null_mut());
                }
                for y in 0..h {
                    for x in 0..w {
                        // row 0 of the library's image is the TOP row; RaytracedPixel.y counts from the bottom and
                        // `raytrace` flips it back (main.rs:1461-1466)
                        let i = ((y * w + x) * 4) as usize;
                        let color = crate::types::Color { r: rgba[i], g: rgba[i + 1], b: rgba[i + 2], a: 1f32 };
                        if tx.send(crate::RaytracedPixel { x: x as u32, y: (h - 1 - y) as u32, color }).is_err() {
                            return;
                        }
                    }
                }
                workblocks_done.store((total_workblocks as i64 * done as i64 / spp as i64) as i32, Ordering::SeqCst);
            }
            unsafe {
                rt_device_free(self.ctxs[0], shown);
                for (g, &p) in self.sums.iter().enumerate() {
                    rt_device_free(self.ctxs[g], p as *mut c_void);
                }
                rt_group_destroy(self.group); // destroys the contexts it owns
            }
        })
    }

    pub fn seed(&self) -> u32 { self.seed }
}

#[derive(Clone, Copy)]
struct SendPtr<T>(*mut T);
unsafe impl<T> Send for SendPtr<T> {}

fn lib(ctx: *const rt_ctx, status: i32) -> Result<(), GpuError> {
    check(ctx, status).map_err(GpuError::Library)
}
