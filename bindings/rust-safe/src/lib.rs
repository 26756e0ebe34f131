//! Safe layer over `sar_sys` (the raw `extern "C"` bindings of `include/sar.h`) with the names and call shapes of
//! Icelk/strange-attractor-renderer's own surface, so that `src/bin/main.rs:483-517` keeps its structure:
//!
//! | reference (`src/lib.rs`)                                  | here                                   |
//! |-----------------------------------------------------------|----------------------------------------|
//! | `Runtime::new(&config)` (:660), `reset` (:682), `merge` (:708) | `GpuRuntime::new / reset / merge`  |
//! | `render(&config, &mut runtime)` (:747)                    | `render(&config, &mut gpu_runtime)`    |
//! | `colorize(&config, &runtime) -> FinalImage` (:841)        | `colorize(&config, &gpu_runtime)`      |
//! | `ParallelRenderer::new()` (:919), `shutdown` (:1020)      | `GpuRenderer::new / new_multi / shutdown` |
//! | `render_parallel(&mut renderer, config, jobs_per_thread)` (:1051) | `render_parallel(&mut gpu_renderer, config, jobs_per_thread)` |
//!
//! SOURCE ONLY: the image this repository is built in has no Rust toolchain, so this crate has never been compiled
//! there; the same call sequence is exercised from a compiled C program (`tests/test_c_program.py`) and through
//! ctypes by the whole `-m gpu` suite.
//!
//! What cannot cross a C ABI: arbitrary `impl Attractor` / closure colour transforms. The layer is implemented for
//! `Config<PolynomialSprott2Degree, T>` with `T` one of the two built-in transforms (`Mi355xTransform`).
//! Error behaviour is the reference's: a failed call panics (`assert_eq!` / `unwrap` at :709-710, :990, :1024) with
//! the library's message.

use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

use image::ImageBuffer;
use sar_sys as sys;
use strange_attractor_renderer::attractors::PolynomialSprott2Degree;
use strange_attractor_renderer::config::color_transforms;
use strange_attractor_renderer::{ColorTransform, Config, FinalImage, RenderKind, Vec3, View};

/// Panics with the library's message on a non-zero status — the reference's error behaviour.
#[track_caller]
pub fn check(status: c_int) {
    if status != sys::SAR_OK {
        let (name, msg) = unsafe {
            (
                CStr::from_ptr(sys::sar_status_string(status)).to_string_lossy().into_owned(),
                CStr::from_ptr(sys::sar_last_error()).to_string_lossy().into_owned(),
            )
        };
        panic!("libsar_hip: {name} ({status}): {msg}");
    }
}

/// The built-in colour transforms, named for the ABI (`SAR_CT_*`, offset, factor).
pub trait Mi355xTransform: ColorTransform {
    fn abi(&self) -> (i32, f64, f64);
}
impl Mi355xTransform for color_transforms::AdjustedVelocity {
    fn abi(&self) -> (i32, f64, f64) {
        (sys::SAR_CT_ADJUSTED_VELOCITY, self.offset, self.factor)
    }
}
/// `color_transforms::Function` is a bare `fn` pointer: the only one the device implements is
/// `color_transforms::poisson_saturne` (:520-558). Any other function is refused (probed on a few points, bit for bit).
impl Mi355xTransform for color_transforms::Function {
    fn abi(&self) -> (i32, f64, f64) {
        let view = View {
            center_camera: Vec3::new(-0.005, 0.262, -0.246),
            rotation: strange_attractor_renderer::EulerAxisRotation { axis: Vec3::new(0.3, 0.76, 0.57), rotation: 1.78 },
            scale: 1.0,
        };
        let probes = [
            (Vec3::new(0.01, -0.02, 0.03), Vec3::new(0.1, 0.2, -0.3)),
            (Vec3::new(-0.4, 0.1, 0.0), Vec3::new(-0.2, 0.45, 0.05)),
            (Vec3::new(0.0, 0.0, 0.0), Vec3::new(0.3, -0.01, -0.6)),
        ];
        for (d, s) in probes {
            let a = self(d, s, &view);
            let b = color_transforms::poisson_saturne(d, s, &view);
            assert!(a.to_bits() == b.to_bits(), "only color_transforms::poisson_saturne can run on the device");
        }
        (sys::SAR_CT_POISSON_SATURNE, 0.0, 0.0)
    }
}

/// The reference's default palette (`Colors::default`, :480-492). `Palette` keeps its list private, so a custom
/// palette has to be handed over explicitly (`GpuOptions::palette`); the default one is recognised by value.
const DEFAULT_PALETTE: [[f64; 3]; 6] =
    [[1., 1., 0.5], [0.5, 1., 0.5], [1., 0.5, 0.5], [0.5, 1., 1.], [0.5, 0.5, 1.], [1., 0.5, 1.]];

/// What the reference hides and the device needs: the seed of the start-point stream (the reference seeds from OS
/// entropy, :656), the device, and the palette entries when they are not the default ones.
#[derive(Clone, Debug, Default)]
pub struct GpuOptions {
    pub device: i32,
    pub seed: u64,
    pub palette: Option<Vec<[f64; 3]>>,
}

fn to_abi<T: Mi355xTransform>(c: &Config<PolynomialSprott2Degree, T>, o: &GpuOptions) -> sys::SarConfig {
    let (kind, off, fac) = c.color_transform.abi();
    let mut s: sys::SarConfig = unsafe { std::mem::zeroed() };
    s.iterations = c.iterations as u64;
    s.width = c.width;
    s.height = c.height;
    s.render_kind = match c.render {
        RenderKind::Gas => sys::SAR_RENDER_GAS,
        RenderKind::Depth => sys::SAR_RENDER_DEPTH,
    };
    s.transparent = c.transparent as i32;
    s.angle = c.angle;
    s.silent = c.silent as i32;
    s.attractor_kind = 0;
    s.coeff_x = c.attractor.x;
    s.coeff_y = c.attractor.y;
    s.coeff_z = c.attractor.z;
    // palette: the user's entries; the library duplicates the last one itself (Palette::new, :416-418)
    let entries: Vec<[f64; 3]> = match &o.palette {
        Some(p) => p.clone(),
        None => {
            // the config must carry the default palette: check it through the one public view of it, `interpolate`
            let n = c.colors.palette.count();
            assert!(n == DEFAULT_PALETTE.len(), "custom palette: pass its entries in GpuOptions::palette");
            for (k, e) in DEFAULT_PALETTE.iter().enumerate() {
                let got = c.colors.palette.interpolate((k as f64 + 0.5) / n as f64).0;
                let nxt = DEFAULT_PALETTE[(k + 1).min(n - 1)];
                for ch in 0..3 {
                    let want = (nxt[ch] * 0.5 + e[ch] * 0.5).sqrt();
                    assert!((got[ch] - want).abs() < 1e-9, "custom palette: pass its entries in GpuOptions::palette");
                }
            }
            DEFAULT_PALETTE.to_vec()
        }
    };
    assert!(!entries.is_empty() && entries.len() <= sys::SAR_PALETTE_MAX, "palette: 1..=15 entries");
    s.palette_len = entries.len() as u32;
    for (i, e) in entries.iter().enumerate() {
        s.palette_rgb[i] = *e;
    }
    s.brightness_offset = c.colors.brighness.offset;
    s.brightness_factor = c.colors.brighness.factor;
    let v = &c.view;
    s.center_camera = [v.center_camera.x, v.center_camera.y, v.center_camera.z];
    s.rotation_axis = [v.rotation.axis.x, v.rotation.axis.y, v.rotation.axis.z];
    s.rotation_angle = v.rotation.rotation;
    s.scale = v.scale;
    s.color_transform = kind;
    s.ct_offset = off;
    s.ct_factor = fac;
    s.seed = o.seed;
    s.jobs_total = 1;
    s
}

/// `Runtime` (:631-646) living in the HBM of one GPU.
pub struct GpuRuntime {
    raw: *mut sys::SarRuntime,
    owned: bool,
    opts: GpuOptions,
}
// same contract as `&mut Runtime`: one user at a time, may move between threads
unsafe impl Send for GpuRuntime {}

impl GpuRuntime {
    /// `Runtime::new(&config)` (:660-665).
    pub fn new<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>) -> Self {
        Self::with_options(config, GpuOptions::default())
    }
    pub fn with_options<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, opts: GpuOptions) -> Self {
        let abi = to_abi(config, &opts);
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::sar_runtime_new(&abi, opts.device, &mut raw) });
        Self { raw, owned: true, opts }
    }
    /// `Runtime::reset` (:682-699).
    pub fn reset(&mut self) {
        check(unsafe { sys::sar_runtime_reset(self.raw) });
    }
    /// `Runtime::merge` (:708-738). Panics on a size mismatch like the reference's `assert_eq!` (:709-710).
    pub fn merge(&mut self, other: &Self) {
        check(unsafe { sys::sar_runtime_merge(self.raw, other.raw) });
    }
    /// The private buffers of `Runtime` (:633-643), read back for inspection.
    pub fn count(&mut self) -> Vec<u32> {
        let (w, h) = self.dims();
        let mut out = vec![0u32; (w as usize) * (h as usize)];
        check(unsafe { sys::sar_runtime_count(self.raw, out.as_mut_ptr()) });
        out
    }
    pub fn max(&mut self) -> u32 {
        let mut m = 0u32;
        check(unsafe { sys::sar_runtime_max(self.raw, &mut m) });
        m
    }
    pub fn dims(&self) -> (u32, u32) {
        let (mut w, mut h) = (0u32, 0u32);
        check(unsafe { sys::sar_runtime_dims(self.raw, &mut w, &mut h) });
        (w, h)
    }
    pub fn as_raw(&self) -> *mut sys::SarRuntime {
        self.raw
    }
}
impl Drop for GpuRuntime {
    fn drop(&mut self) {
        if self.owned && !self.raw.is_null() {
            unsafe { sys::sar_runtime_free(self.raw) };
        }
    }
}

/// `render(&config, &mut runtime)` (:747-838): ONE trajectory of `config.iterations` counted iterations from the
/// next start point of the runtime's stream, accumulated into the runtime (no reset).
pub fn render<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, runtime: &mut GpuRuntime) {
    let abi = to_abi(config, &runtime.opts);
    check(unsafe { sys::sar_render(&abi, runtime.raw) });
}

/// The data-parallel form of calling `render` `jobs` times on one un-reset runtime (what one reference worker does,
/// :956-988): `jobs` trajectories of `config.iterations / jobs` iterations, sequential (job-major) semantics.
pub fn render_jobs<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, runtime: &mut GpuRuntime, jobs: u32) {
    let mut abi = to_abi(config, &runtime.opts);
    abi.jobs_total = jobs;
    check(unsafe { sys::sar_render_jobs(&abi, runtime.raw, std::ptr::null()) });
}

/// The frames of a `sequence` sweep (src/bin/main.rs:493-517 renders them one after the other, each a reset and a
/// render of fresh jobs) through ONE set of launches: frame i is `render_jobs(&configs[i], &mut runtimes[i], jobs)`, bit for
/// bit — a frame of 65 536 jobs fills a third of an MI355X, eight or sixteen of them fill it. The frames must share the
/// image size, the job split and the scale; anything else runs frame after frame inside the call.
pub fn render_jobs_batch<T: Mi355xTransform>(configs: &[Config<PolynomialSprott2Degree, T>], runtimes: &mut [GpuRuntime], jobs: u32) {
    assert!(configs.len() == runtimes.len(), "one runtime per frame");
    let abis: Vec<sys::SarConfig> = configs
        .iter()
        .zip(runtimes.iter())
        .map(|(c, rt)| {
            let mut abi = to_abi(c, &rt.opts);
            abi.jobs_total = jobs;
            abi
        })
        .collect();
    let cfg_ptrs: Vec<*const sys::SarConfig> = abis.iter().map(|a| a as *const _).collect();
    let rt_ptrs: Vec<*mut sys::SarRuntime> = runtimes.iter().map(|rt| rt.raw).collect();
    // start points NULL: every frame draws from its own runtime's stream, as `render` does (:748)
    check(unsafe { sys::sar_render_jobs_batch(abis.len() as u32, cfg_ptrs.as_ptr(), rt_ptrs.as_ptr(), std::ptr::null()) });
}

/// How many frames of this shape the library would put into one `render_jobs_batch` (a multiple of eight: the chip has eight XCDs).
pub fn batch_frames<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, runtime: &mut GpuRuntime, jobs: u32) -> usize {
    let mut abi = to_abi(config, &runtime.opts);
    abi.jobs_total = jobs;
    let mut n = 0u32;
    check(unsafe { sys::sar_runtime_batch_frames(&abi, runtime.raw, &mut n) });
    n as usize
}

/// The id of the sources `libsar_hip.so` was built from (16 hex digits): what a deployment logs next to its results.
pub fn build_id() -> String {
    unsafe { std::ffi::CStr::from_ptr(sys::sar_build_id()) }.to_string_lossy().into_owned()
}

/// `colorize(&config, &runtime) -> FinalImage` (:841-904).
pub fn colorize<T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, runtime: &GpuRuntime) -> FinalImage {
    let abi = to_abi(config, &runtime.opts);
    let mut buf = vec![0u16; config.width as usize * config.height as usize * 4];
    check(unsafe { sys::sar_colorize(&abi, runtime.raw, buf.as_mut_ptr()) });
    // FinalImage = ImageBuffer<Rgba<u16>, Vec<u16>> (:625)
    ImageBuffer::from_raw(config.width, config.height, buf).expect("buffer has width*height*4 samples")
}

/// A page-locked host image in one of the CLI's export formats (`sar_image_format`): the target of an asynchronous
/// read-back, what the `sequence` loop (src/bin/main.rs:493-517) hands to its writer threads.
pub struct PinnedImage {
    ptr: *mut c_void,
    bytes: usize,
    format: c_int,
}
unsafe impl Send for PinnedImage {}

impl PinnedImage {
    pub fn new(format: c_int, width: u32, height: u32) -> Self {
        let bytes = unsafe { sys::sar_image_bytes(format, width, height) };
        assert!(bytes > 0, "unknown image format");
        let mut ptr = std::ptr::null_mut();
        check(unsafe { sys::sar_host_alloc(bytes, &mut ptr) });
        Self { ptr, bytes, format }
    }
    /// The samples (host-endian, `format`'s layout). Only reachable while no read-back into the image is pending: a
    /// `PendingImage` owns the image until it has been waited for.
    pub fn bytes(&self) -> &[u8] {
        unsafe { std::slice::from_raw_parts(self.ptr as *const u8, self.bytes) }
    }
}
impl Drop for PinnedImage {
    fn drop(&mut self) {
        unsafe { sys::sar_host_free(self.ptr) };
    }
}

/// A read-back in flight. It OWNS the image until `wait` hands it back: `std::mem::forget` of a handle that merely
/// borrowed the image would end the borrow without waiting (the image could then be read, or freed, under the DMA);
/// forgetting an owning handle only leaks the page-locked memory, which nobody can touch any more.
pub struct PendingImage<'a> {
    runtime: &'a GpuRuntime,
    ticket: u64,
    image: Option<PinnedImage>,
}
impl<'a> PendingImage<'a> {
    /// Blocks until the frame is in the image and gives the image back.
    pub fn wait(mut self) -> PinnedImage {
        check(unsafe { sys::sar_runtime_wait_image(self.runtime.raw, self.ticket) });
        self.image.take().expect("a pending image holds its image until it is waited for")
    }
}
impl Drop for PendingImage<'_> {
    fn drop(&mut self) {
        if self.image.is_some() {
            // dropped without `wait`: the copy must have landed before the image is freed (PinnedImage::drop runs next)
            unsafe { sys::sar_runtime_wait_image(self.runtime.raw, self.ticket) };
        }
    }
}

/// `colorize` + the CLI's format conversion, only ENQUEUED (`sar_colorize_format_async`). The handle borrows the runtime
/// (shared) and takes the image until the frame has landed: a `sequence` loop keeps the GPU busy by rendering frame k+1
/// on a SECOND runtime meanwhile (the C ABI would also allow resetting this one — the borrow is the safe subset).
pub fn colorize_format_async<'a, T: Mi355xTransform>(config: &Config<PolynomialSprott2Degree, T>, runtime: &'a GpuRuntime,
                                                     image: PinnedImage) -> PendingImage<'a> {
    let abi = to_abi(config, &runtime.opts);
    let mut ticket = 0u64;
    check(unsafe { sys::sar_colorize_format_async(&abi, runtime.raw, image.format, image.ptr, &mut ticket) });
    PendingImage { runtime, ticket, image: Some(image) }
}

/// `ParallelRenderer` (:908-915): owns the execution units the job split divides by — here the lanes of one or
/// several GPUs instead of OS threads.
pub struct GpuRenderer {
    raw: *mut sys::SarRenderer,
    opts: GpuOptions,
}
unsafe impl Send for GpuRenderer {}

impl GpuRenderer {
    /// `ParallelRenderer::new()` (:919-1004) on one GPU; `units == 0` selects the device default (64 per CU).
    pub fn new(opts: GpuOptions, units: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::sar_renderer_new(opts.device, units, opts.seed, &mut raw) });
        Self { raw, opts }
    }
    /// The same over several GPUs of one node: jobs sharded over the devices, partial buffers exchanged
    /// point-to-point over xGMI and folded with `Runtime::merge` in device order — all behind the C ABI.
    pub fn new_multi(devices: &[i32], opts: GpuOptions, units: u32) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { sys::sar_renderer_new_multi(devices.as_ptr(), devices.len() as u32, units, opts.seed, &mut raw) });
        Self { raw, opts }
    }
    /// How the devices exchange their partial buffers before the fold (:1068-1076): 0 automatic, 1 whole image slices by peer
    /// copies, 2 only the records of the 64-pixel granules a device has touched (a quarter of the bytes for the presets).
    pub fn set_exchange(&mut self, mode: u32) {
        check(unsafe { sys::sar_renderer_set_exchange(self.raw, mode) });
    }
    /// `num_threads()` (:1016-1018): the divisor of the job split.
    pub fn num_threads(&self) -> usize {
        let mut n = 0u32;
        check(unsafe { sys::sar_renderer_num_units(self.raw, &mut n) });
        n as usize
    }
    /// `ParallelRenderer::shutdown` (:1020-1025).
    pub fn shutdown(mut self) {
        self.close();
    }
    fn close(&mut self) {
        if !self.raw.is_null() {
            unsafe { sys::sar_renderer_shutdown(self.raw) };
            self.raw = std::ptr::null_mut();
        }
    }
    /// The renderer's merged runtime, e.g. to read `count` after a frame. It lives inside the renderer: the borrow keeps
    /// the renderer from being shut down, dropped or rendered into again while the view is alive.
    pub fn runtime(&mut self) -> BorrowedRuntime<'_> {
        let mut rt = std::ptr::null_mut();
        check(unsafe { sys::sar_renderer_runtime(self.raw, &mut rt) });
        BorrowedRuntime { rt: GpuRuntime { raw: rt, owned: false, opts: self.opts.clone() }, _renderer: std::marker::PhantomData }
    }
}

/// A `GpuRuntime` that belongs to a `GpuRenderer` (`GpuRenderer::runtime`): usable like `&GpuRuntime` for as long as the
/// renderer stays mutably borrowed, never freed through this handle.
pub struct BorrowedRuntime<'a> {
    rt: GpuRuntime,
    _renderer: std::marker::PhantomData<&'a mut GpuRenderer>,
}
impl<'a> std::ops::Deref for BorrowedRuntime<'a> {
    type Target = GpuRuntime;
    fn deref(&self) -> &GpuRuntime {
        &self.rt
    }
}
impl Drop for GpuRenderer {
    fn drop(&mut self) {
        self.close();
    }
}

/// `render_parallel(&mut renderer, config, jobs_per_thread) -> FinalImage` (:1051-1082): `N / T / J` iterations per
/// job (:1058), `T * J` jobs (:1062), reset, render, merge, colorize.
pub fn render_parallel<T: Mi355xTransform>(
    renderer: &mut GpuRenderer,
    config: Config<PolynomialSprott2Degree, T>,
    jobs_per_thread: usize,
) -> FinalImage {
    let abi = to_abi(&config, &renderer.opts);
    let mut buf = vec![0u16; config.width as usize * config.height as usize * 4];
    check(unsafe { sys::sar_render_parallel(renderer.raw, &abi, jobs_per_thread as u32, buf.as_mut_ptr()) });
    ImageBuffer::from_raw(config.width, config.height, buf).expect("buffer has width*height*4 samples")
}

/// The CLI's two code paths (`src/bin/main.rs:483-517`) with this layer in place of the CPU one.
///
/// ```ignore
/// // --single-thread (main.rs:483-491)
/// let mut runtime = GpuRuntime::new(&config);
/// render(&config, &mut runtime);
/// let image = colorize(&config, &runtime);
/// runtime.reset();
/// // default (main.rs:493-517)
/// let mut renderer = GpuRenderer::new(GpuOptions::default(), 0);
/// let image = render_parallel(&mut renderer, config.clone(), jobs_per_thread);
/// renderer.shutdown();
/// ```
pub mod cli_shape {}
