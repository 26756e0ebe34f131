//! Raw bindings to `include/sar.h` (the C ABI of the MI355X iterate/accumulate path).
//!
//! SOURCE ONLY: the image this repository is built in has no Rust toolchain, so this crate has never been
//! compiled there. `#[repr(C)] SarConfig` mirrors `struct sar_config` field by field; the layout the C side
//! expects is pinned by `tests/test_abi_and_host.py::test_struct_layout_matches_c`.
//! Each function names the item of Icelk/strange-attractor-renderer (`src/lib.rs`) it replaces.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const SAR_OK: c_int = 0;
pub const SAR_ERR_INVALID: c_int = 1;
pub const SAR_ERR_DIM_MISMATCH: c_int = 2;
pub const SAR_ERR_NO_DEVICE: c_int = 3;
pub const SAR_ERR_HIP: c_int = 4;
pub const SAR_ERR_OOM: c_int = 5;
pub const SAR_ERR_RANGE: c_int = 6;
pub const SAR_ERR_IO: c_int = 7;

pub const SAR_RENDER_GAS: i32 = 0; // RenderKind::Gas   (:233-239)
pub const SAR_RENDER_DEPTH: i32 = 1; // RenderKind::Depth
pub const SAR_CT_POISSON_SATURNE: i32 = 0; // color_transforms::poisson_saturne (:520)
pub const SAR_CT_ADJUSTED_VELOCITY: i32 = 1; // color_transforms::AdjustedVelocity (:507)
pub const SAR_PALETTE_MAX: usize = 15;
pub const SAR_FMT_RGBA16: c_int = 0;
pub const SAR_FMT_RGB16: c_int = 1;
pub const SAR_FMT_RGBA8: c_int = 2;
pub const SAR_FMT_RGB8: c_int = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct SarConfig {
    pub iterations: u64,
    pub width: u32,
    pub height: u32,
    pub render_kind: i32,
    pub transparent: i32,
    pub angle: f64,
    pub silent: i32,
    pub attractor_kind: i32,
    pub coeff_x: [f64; 10],
    pub coeff_y: [f64; 10],
    pub coeff_z: [f64; 10],
    pub palette_len: u32,
    pub _pad0: u32,
    pub palette_rgb: [[f64; 3]; SAR_PALETTE_MAX],
    pub brightness_offset: f64,
    pub brightness_factor: f64,
    pub center_camera: [f64; 3],
    pub rotation_axis: [f64; 3],
    pub rotation_angle: f64,
    pub scale: f64,
    pub color_transform: i32,
    pub _pad1: i32,
    pub ct_offset: f64,
    pub ct_factor: f64,
    pub seed: u64,
    pub jobs_total: u32,
    pub _pad2: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SarTiming {
    pub iterate_ms: f32,
    pub resolve_ms: f32,
    pub colorize_ms: f32,
    pub merge_ms: f32,
    pub iterate_launches: u32,
    pub warmup_ms: f32,
    pub iterations_counted: u64,
    pub depth_atomics: u64,
    pub depth_candidates: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SarParallelTiming {
    pub total_ms: f32,
    pub render_ms: f32,
    pub exchange_ms: f32,
    pub colorize_ms: f32,
    pub n_devices: u32,
    pub peer_access_failures: u32,
    pub exchange_bytes_per_device: u64,
    pub host_ms_before_exchange: f32,
    pub host_ms_enqueue: f32,
    pub draw_ahead_ms: f32,
    pub _pad: f32,
}

#[repr(C)]
pub struct SarRuntime {
    _private: [u8; 0],
}
#[repr(C)]
pub struct SarRenderer {
    _private: [u8; 0],
}
#[repr(C)]
pub struct SarExchange {
    _private: [u8; 0],
}
/// Geometry of the sliced exchange (sar_exchange_new).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SarExchangeLayout {
    pub world: u32,
    pub rank: u32,
    pub slice_pixels: u32,
    pub first_px: u32,
    pub n_px: u32,
    pub granules: u32,
    pub block_bytes: u64,
}

extern "C" {
    pub fn sar_abi_version() -> c_int;
    pub fn sar_build_id() -> *const c_char;
    pub fn sar_device_pci_bus_id(device: c_int, out: *mut c_char, cap: usize) -> c_int;
    pub fn sar_checksum_fnv1a64(data_host: *const c_void, nbytes: usize, out: *mut u64) -> c_int;
    pub fn sar_status_string(status: c_int) -> *const c_char;
    pub fn sar_last_error() -> *const c_char;
    pub fn sar_device_count(out_count: *mut c_int) -> c_int;

    pub fn sar_config_poisson_saturne(out: *mut SarConfig) -> c_int; // Config::poisson_saturne (:310)
    pub fn sar_config_solar_sail(out: *mut SarConfig) -> c_int; // Config::solar_sail (:355)
    pub fn sar_config_validate(cfg: *const SarConfig) -> c_int;
    pub fn sar_rotation_matrix(cfg: *const SarConfig, m_out: *mut f64) -> c_int; // to_rotation_matrix (:176)
    pub fn sar_start_points(seed: u64, first_job: u64, n_jobs: u32, xyz_out_host: *mut f64) -> c_int;

    pub fn sar_runtime_new(cfg: *const SarConfig, device: c_int, out: *mut *mut SarRuntime) -> c_int; // Runtime::new (:660)
    pub fn sar_runtime_free(rt: *mut SarRuntime) -> c_int;
    pub fn sar_runtime_reset(rt: *mut SarRuntime) -> c_int; // Runtime::reset (:682)
    pub fn sar_runtime_set_width_height(rt: *mut SarRuntime, width: u32, height: u32) -> c_int; // (:667)
    pub fn sar_runtime_seed(rt: *mut SarRuntime, seed: u64) -> c_int;
    pub fn sar_runtime_merge(dst: *mut SarRuntime, src: *const SarRuntime) -> c_int; // Runtime::merge (:708)
    pub fn sar_runtime_synchronize(rt: *mut SarRuntime) -> c_int;
    pub fn sar_runtime_dims(rt: *const SarRuntime, width: *mut u32, height: *mut u32) -> c_int;
    pub fn sar_runtime_set_stream(rt: *mut SarRuntime, hip_stream: *mut c_void) -> c_int;
    pub fn sar_runtime_get_stream(rt: *const SarRuntime, hip_stream_out: *mut *mut c_void) -> c_int;

    pub fn sar_render(cfg: *const SarConfig, rt: *mut SarRuntime) -> c_int; // render (:747)
    pub fn sar_render_jobs(cfg: *const SarConfig, rt: *mut SarRuntime, starts_xyz_host: *const f64) -> c_int;
    pub fn sar_render_job_range(cfg: *const SarConfig, rt: *mut SarRuntime, n_jobs: u32, iters_per_job: u64,
                                starts_xyz_host: *const f64) -> c_int;
    pub fn sar_colorize(cfg: *const SarConfig, rt: *mut SarRuntime, rgba_out_host: *mut u16) -> c_int; // colorize (:841)
    pub fn sar_render_job_range_device(cfg: *const SarConfig, rt: *mut SarRuntime, n_jobs: u32, iters_per_job: u64,
                                       starts_xyz_dev: *const f64) -> c_int;
    pub fn sar_runtime_prefetch_device(cfg: *const SarConfig, rt: *mut SarRuntime, n_jobs: u32, iters_per_job: u64,
                                       starts_xyz_dev: *const f64) -> c_int;
    /// F frames of a sweep (src/bin/main.rs:493-517) through one set of launches: frame i == sar_render_jobs(cfgs[i], rts[i], starts[i]).
    pub fn sar_render_jobs_batch(n_frames: u32, cfgs: *const *const SarConfig, rts: *const *mut SarRuntime,
                                 starts_xyz_host: *const *const f64) -> c_int;
    /// n runtimes for the frames of one batch: one stream, one device and one page-locked allocation for all of them.
    pub fn sar_runtime_new_group(cfg: *const SarConfig, device: c_int, n: u32, out: *mut *mut SarRuntime) -> c_int;
    pub fn sar_runtime_reset_batch(n: u32, rts: *const *mut SarRuntime) -> c_int;
    pub fn sar_colorize_device_batch(n: u32, cfgs: *const *const SarConfig, rts: *const *mut SarRuntime, rgba_out_dev: *const *mut c_void) -> c_int;
    pub fn sar_runtime_batch_frames(cfg: *const SarConfig, rt: *mut SarRuntime, out_frames: *mut u32) -> c_int;
    pub fn sar_renderer_set_exchange(r: *mut SarRenderer, mode: u32) -> c_int;
    pub fn sar_runtime_get_copy_stream(rt: *mut SarRuntime, hip_stream_out: *mut *mut c_void) -> c_int;
    pub fn sar_runtime_set_copy_stream(rt: *mut SarRuntime, hip_stream: *mut c_void) -> c_int;
    pub fn sar_colorize_device(cfg: *const SarConfig, rt: *mut SarRuntime, rgba_out_dev: *mut c_void) -> c_int;
    pub fn sar_runtime_describe_last_launch(rt: *const SarRuntime, out: *mut c_char, cap: usize) -> c_int;

    pub fn sar_runtime_extent(cfg: *const SarConfig, rt: *mut SarRuntime, n_jobs: u32, iters_per_job: u64,
                              starts_xyz_host: *const f64, out12: *mut f64) -> c_int;
    // image export (src/bin/main.rs:40-100)
    pub fn sar_image_format(transparent: c_int, eight_bit: c_int) -> c_int;
    pub fn sar_image_bytes(format: c_int, width: u32, height: u32) -> usize;
    pub fn sar_image_convert_device(rt: *mut SarRuntime, rgba16_dev: *const c_void, format: c_int, out_dev: *mut c_void) -> c_int;
    pub fn sar_colorize_format(cfg: *const SarConfig, rt: *mut SarRuntime, format: c_int, out_host: *mut c_void) -> c_int;
    pub fn sar_colorize_format_async(cfg: *const SarConfig, rt: *mut SarRuntime, format: c_int, out_host: *mut c_void, ticket_out: *mut u64) -> c_int;
    pub fn sar_runtime_wait_image(rt: *mut SarRuntime, ticket: u64) -> c_int;
    pub fn sar_runtime_read_image_async(rt: *mut SarRuntime, out_host: *mut c_void, ticket_out: *mut u64) -> c_int;
    pub fn sar_runtime_image_done(rt: *mut SarRuntime, ticket: u64, done_out: *mut c_int) -> c_int;
    pub fn sar_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn sar_host_free(p: *mut c_void) -> c_int;
    pub fn sar_host_reserve(bytes: usize, count: u32) -> c_int;
    pub fn sar_write_png(path: *const c_char, format: c_int, width: u32, height: u32, pixels: *const c_void) -> c_int;
    pub fn sar_write_bmp(path: *const c_char, format: c_int, width: u32, height: u32, pixels: *const c_void) -> c_int;
    pub fn sar_write_pam(path: *const c_char, format: c_int, width: u32, height: u32, pixels: *const c_void) -> c_int;
    pub fn sar_runtime_count(rt: *mut SarRuntime, out_host: *mut u32) -> c_int;
    pub fn sar_runtime_steps(rt: *mut SarRuntime, out_host: *mut f64) -> c_int;
    pub fn sar_runtime_zbuf(rt: *mut SarRuntime, out_host: *mut f32) -> c_int;
    pub fn sar_runtime_max(rt: *mut SarRuntime, out_max: *mut u32) -> c_int;
    pub fn sar_runtime_load(rt: *mut SarRuntime, count_host: *const u32, steps_host: *const f64,
                            zbuf_host: *const f32, max: u32) -> c_int;

    // The ONE exchange step before colorize of a one-process-per-GPU host (Runtime::merge folded in rank order, :708-738, :1068-1076)
    // behind one context object; the collectives between the steps are the caller's.
    pub fn sar_exchange_slice_pixels(npix: u32, world: u32, out_slice_pixels: *mut u32) -> c_int;
    pub fn sar_exchange_new(rt: *mut SarRuntime, world: u32, rank: u32, out: *mut *mut SarExchange, layout_out: *mut SarExchangeLayout) -> c_int;
    pub fn sar_exchange_free(ex: *mut SarExchange) -> c_int;
    pub fn sar_exchange_flags(ex: *mut SarExchange, flags_out_dev: *mut u8) -> c_int;
    pub fn sar_exchange_pack(ex: *mut SarExchange, flags_all_dev: *const u8, dense_above: f64, send_dev: *mut c_void,
                             send_bytes: *mut u64, recv_bytes: *mut u64, sparse_out: *mut c_int) -> c_int;
    pub fn sar_exchange_merge(ex: *mut SarExchange, recv_dev: *const c_void, scalars_out_dev: *mut i64) -> c_int;
    pub fn sar_exchange_finish(ex: *mut SarExchange, scalars_reduced_dev: *const i64) -> c_int;
    pub fn sar_exchange_rooted(ex: *mut SarExchange, step: u32, key_i64_dev: *mut c_void, sum_i32_dev: *mut c_void) -> c_int;
    pub fn sar_colorize_range_device(cfg: *const SarConfig, rt: *mut SarRuntime, first_px: u32, n_px: u32,
                                     rgba_out_dev: *mut c_void) -> c_int;

    pub fn sar_renderer_new_multi(devices: *const c_int, n_devices: u32, units: u32, seed: u64,
                                  out: *mut *mut SarRenderer) -> c_int; // ParallelRenderer::new over several GPUs (:919)
    pub fn sar_renderer_num_devices(r: *const SarRenderer, out_devices: *mut u32) -> c_int;
    pub fn sar_renderer_last_timing(r: *const SarRenderer, out: *mut SarParallelTiming) -> c_int;
    pub fn sar_renderer_new(device: c_int, units: u32, seed: u64, out: *mut *mut SarRenderer) -> c_int; // ParallelRenderer::new (:919)
    pub fn sar_renderer_num_units(r: *const SarRenderer, out_units: *mut u32) -> c_int;
    pub fn sar_renderer_shutdown(r: *mut SarRenderer) -> c_int; // ParallelRenderer::shutdown (:1020)
    pub fn sar_render_parallel(r: *mut SarRenderer, cfg: *const SarConfig, jobs_per_unit: u32,
                               rgba_out_host: *mut u16) -> c_int; // render_parallel (:1051)
    pub fn sar_renderer_runtime(r: *mut SarRenderer, out_borrowed: *mut *mut SarRuntime) -> c_int;

    pub fn sar_runtime_enable_timing(rt: *mut SarRuntime, enabled: c_int) -> c_int;
    pub fn sar_runtime_last_timing(rt: *mut SarRuntime, out: *mut SarTiming) -> c_int;
    pub fn sar_runtime_set_option(rt: *mut SarRuntime, name: *const c_char, value: u64) -> c_int;
    pub fn sar_bin_geometry(width: u32, height: u32, bin_shift: u32, bin_interleave: u32, out: *mut u32) -> c_int;
}
