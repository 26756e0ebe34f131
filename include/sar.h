/*
 * sar.h — C ABI of the MI355X-native strange-attractor iterate/accumulate path.
 *
 * This is the drop-in boundary for ONE hot path of Icelk/strange-attractor-renderer:
 *   iterate the polynomial-Sprott map -> scatter-accumulate count / depth / colour index
 *   -> merge partial buffers -> tone-map (colorize).
 *
 * The reference has no FFI: its boundary is the Rust crate surface
 *   Config / View / Colors / RenderKind          (reference src/lib.rs:228-492)
 *   Runtime::{new, reset, merge}                 (src/lib.rs:631-739)
 *   render(&Config, &mut Runtime)                (src/lib.rs:747-838)
 *   colorize(&Config, &Runtime) -> FinalImage    (src/lib.rs:841-904)
 *   ParallelRenderer::{new, shutdown}            (src/lib.rs:908-1031)
 *   render_parallel(&mut ParallelRenderer, Config, jobs_per_thread) (src/lib.rs:1051-1082)
 * Each entry point below names the reference item it replaces. A Rust `extern "C"` block a
 * maintainer would add to bind these is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int status (SAR_OK == 0); nothing unwinds across this boundary
 *     (the reference panics instead: src/lib.rs:709-710, 990, 1024);
 *   - plain pointers and sizes only; no C++/torch types;
 *   - a sar_runtime is bound to one HIP device + one HIP stream and is NOT thread-safe
 *     (same contract as `&mut Runtime`);
 *   - pointers named *_host are host memory, *_dev are device memory on the runtime's device;
 *   - image buffers are row-major, index = y*width + x (image::ImageBuffer Luma layout,
 *     src/lib.rs:808).
 */
#ifndef SAR_H
#define SAR_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAR_ABI_VERSION 6  /* 6: sar_runtime_new_group, sar_exchange_* (one context object for the multi-process exchange) */

/* ---- status codes ------------------------------------------------------------------------
 * Every function that can fail returns one of these (the reference panics instead: assert_eq! / unwrap / expect); the text is
 * in sar_last_error(). No C++ exception leaves the library: one thrown inside an entry point comes back as SAR_ERR_OOM
 * (std::bad_alloc) or SAR_ERR_INVALID (anything else). */
enum {
    SAR_OK = 0,
    SAR_ERR_INVALID = 1,      /* NULL pointer / bad enum / empty palette (ref: Palette::new panics, :413-418) */
    SAR_ERR_DIM_MISMATCH = 2, /* merge of runtimes with different sizes (ref: assert_eq!, :709-710) */
    SAR_ERR_NO_DEVICE = 3,    /* no HIP device / HIP runtime unavailable */
    SAR_ERR_HIP = 4,          /* a HIP call failed; see sar_last_error() */
    SAR_ERR_OOM = 5,          /* device or host memory exhausted */
    SAR_ERR_RANGE = 6,        /* a size is out of range: width*height > 2^31-1, units*jobs_per_unit > 2^32-1 */
    SAR_ERR_IO = 7            /* an image file could not be created or written (ref: File::create(..).unwrap(), main.rs:103) */
};

/* ---- closed enums (Rust generics / closures cannot cross a C ABI) ------------------------- */
enum { SAR_RENDER_GAS = 0, SAR_RENDER_DEPTH = 1 };            /* RenderKind, src/lib.rs:233-239 */
enum { SAR_ATTRACTOR_SPROTT2 = 0 };                           /* PolynomialSprott2Degree, :575-580 */
enum { SAR_CT_POISSON_SATURNE = 0, SAR_CT_ADJUSTED_VELOCITY = 1 }; /* color_transforms, :498-559 */

#define SAR_PALETTE_MAX 15   /* user entries; the duplicated last entry (:416-418) is added internally */

/*
 * POD mirror of Config<A,T> + View + Colors (src/lib.rs:253-308, 389-492), plus the two things
 * the reference hides: `seed` (it seeds SmallRng from OS entropy, :656) and `jobs_total`
 * (it derives T*J from the thread pool, :1058-1062).
 */
typedef struct sar_config {
    uint64_t iterations;          /* Config::iterations (:267) */
    uint32_t width;               /* :269 */
    uint32_t height;              /* :271 */
    int32_t  render_kind;         /* SAR_RENDER_*  (:273) */
    int32_t  transparent;         /* bool (:275) */
    double   angle;               /* radians (:277) */
    int32_t  silent;              /* bool (:280); this library never prints */
    int32_t  attractor_kind;      /* SAR_ATTRACTOR_* */
    double   coeff_x[10];         /* PolynomialSprott2Degree::x (:577) */
    double   coeff_y[10];         /* :578 */
    double   coeff_z[10];         /* :579 */
    uint32_t palette_len;         /* number of entries in palette_rgb, 1..SAR_PALETTE_MAX */
    uint32_t _pad0;
    double   palette_rgb[SAR_PALETTE_MAX][3]; /* Palette list (:409), linear r,g,b */
    double   brightness_offset;   /* BrighnessConstants::offset (:394) */
    double   brightness_factor;   /* BrighnessConstants::factor (:395) */
    double   center_camera[3];    /* View::center_camera (:257) */
    double   rotation_axis[3];    /* EulerAxisRotation::axis (:172) — NOT normalised (release build, :181-183) */
    double   rotation_angle;      /* EulerAxisRotation::rotation (:174) */
    double   scale;               /* View::scale (:260) */
    int32_t  color_transform;     /* SAR_CT_* */
    int32_t  _pad1;
    double   ct_offset;           /* AdjustedVelocity::offset (:508) */
    double   ct_factor;           /* AdjustedVelocity::factor (:509) */
    uint64_t seed;                /* seed of the runtime's start-point stream (see sar_start_points) */
    uint32_t jobs_total;          /* number of independent trajectories ("jobs", :1062) sar_render_jobs runs */
    uint32_t _pad2;
} sar_config;

typedef struct sar_runtime sar_runtime;     /* opaque; Runtime, src/lib.rs:631-646 */
typedef struct sar_renderer sar_renderer;   /* opaque; ParallelRenderer, src/lib.rs:908-915 */

/* Per-call device timings (HIP events on the runtime's stream), filled when timing is enabled. */
typedef struct sar_timing {
    float    iterate_ms;    /* sum over launch chunks of the iterate kernel (k_iterate_split / k_iterate_lean) alone */
    float    resolve_ms;    /* depth-winner payload resolve + max reduction */
    float    colorize_ms;   /* last colorize */
    float    merge_ms;      /* last merge */
    uint32_t iterate_launches;
    float    warmup_ms;     /* sum over launch chunks of the warm-up + packing kernel (was padding before ABI 2) */
    uint64_t iterations_counted; /* jobs * iterations-per-job executed by the last render call */
    uint64_t depth_atomics;      /* binned path: global depth atomics issued since the last query (statistic) */
    uint64_t depth_candidates;   /* binned path: visits that passed the depth-hint filter (one chip-wide key load each) since the last query */
} sar_timing;

/* ---- misc ---------------------------------------------------------------------------------- */
int         sar_abi_version(void);
/* Which sources this binary was built from: the first 16 hex digits of the SHA-256 over the files under csrc, this header and the compiler
 * flags (strange_attractor_renderer_amd/build.py: source_id). The Python loader recomputes it from the tree and refuses a
 * library that was built from other sources. */
const char* sar_build_id(void);
const char* sar_status_string(int status);
const char* sar_last_error(void);           /* thread-local, human readable */
int         sar_device_count(int* out_count);
/* "0000:c5:00.0" of a HIP device ordinal (cap >= 16): which physical GPU a rank really runs on, for run records. */
int         sar_device_pci_bus_id(int device, char* out, size_t cap);
/* FNV-1a (64 bit) over a host buffer: the checksum tests/golden/fullsize_checksums.json freezes the full-size frames with,
 * so that a run record can say "this frame's count / zbuf / steps / RGBA16 are the committed ones" without shipping them. */
int         sar_checksum_fnv1a64(const void* data_host, size_t nbytes, uint64_t* out);

/* ---- Config presets (data only) -------------------------------------------------------------- */
/* Config::new defaults (:289-307) + poisson_saturne() values (:310-352). */
int sar_config_poisson_saturne(sar_config* out);
/* Config::new defaults + solar_sail() values (:354-387) (library scale 1.7; the CLI overrides to 1). */
int sar_config_solar_sail(sar_config* out);
/* Checks enums, palette_len, non-zero dimensions. */
int sar_config_validate(const sar_config* cfg);

/* ---- host-side setup math (no device needed) -------------------------------------------------- */
/* EulerAxisRotation::to_rotation_matrix, release semantics (:176-196). m is row-major 3x3. */
int sar_rotation_matrix(const sar_config* cfg, double m_out[9]);
/*
 * The start-point stream the reference leaves to OS entropy (:656, :748). SplitMix64(seed) seeds xoshiro256++ (four outputs
 * = the state: what rand 0.9 documents for SmallRng::seed_from_u64 on 64-bit targets); every f64 is (next_u64 >> 11) * 2^-53,
 * multiplied by 0.1 (`rng.random::<Vec3>() * 0.1`). Jobs come in BLOCKS of 4096: block b draws from the generator after b
 * applications of xoshiro256's published jump() (2^128 steps each), and job k takes draws 3i..3i+2 of block k / 4096 as
 * x, y, z, with i = k % 4096 — so the first 4096 jobs are the plain stream, and any job's point is found without drawing
 * its predecessors' (a multi-device render_parallel draws every device's job slice on its own host thread). Writes n_jobs*3
 * doubles for jobs [first_job, first_job+n_jobs) (first_job <= 2^36: reaching a job costs first_job / 4096 jumps of about a
 * microsecond each). The published vectors of both generators and the jump polynomial are held by tests/test_oracle_kat.py.
 */
int sar_start_points(uint64_t seed, uint64_t first_job, uint32_t n_jobs, double* xyz_out_host);

/* ---- Runtime (src/lib.rs:631-739) -------------------------------------------------------------- */
/* Runtime::new (:660-665): allocates count/steps/zbuf for cfg->width x cfg->height on `device`,
 * resets them, seeds the start-point stream with cfg->seed. */
int sar_runtime_new(const sar_config* cfg, int device, sar_runtime** out);
int sar_runtime_free(sar_runtime* rt);
/* n runtimes (1..32) for the frames of ONE batch (sar_render_jobs_batch; the `sequence` loop, src/bin/main.rs:493-517, keeps one
 * renderer for all its frames): one stream, one read-back stream, every runtime's buffers — sized for frames like cfg in batches
 * of n — carved from ONE device allocation instead of some twenty. Each out[i] is an ordinary runtime, freed in any order. */
int sar_runtime_new_group(const sar_config* cfg, int device, uint32_t n, sar_runtime** out /* [n] */);
/* Runtime::reset (:682-699): count<-0, steps<-0.0, zbuf<--1.0, max<-0. The RNG stream is NOT reseeded. */
int sar_runtime_reset(sar_runtime* rt);
/* n resets at once — the frames of a batch of a sweep (:950-951 per frame): ONE launch for the runtimes that share a stream. */
int sar_runtime_reset_batch(uint32_t n, sar_runtime* const* rts);
/* Runtime::set_width_height (:667-675): reallocates + resets only when the size changes. */
int sar_runtime_set_width_height(sar_runtime* rt, uint32_t width, uint32_t height);
/* Reseed the start-point stream (the reference has no equivalent; needed for reproducibility). */
int sar_runtime_seed(sar_runtime* rt, uint64_t seed);
/* Runtime::merge (:708-738): dst.count += src.count (wrapping); dst.max = max(dst.max, merged counts);
 * where src.zbuf > dst.zbuf (strict; dst wins ties) take src's steps and zbuf. Same device required. */
int sar_runtime_merge(sar_runtime* dst, const sar_runtime* src);
int sar_runtime_synchronize(sar_runtime* rt);
int sar_runtime_dims(const sar_runtime* rt, uint32_t* width, uint32_t* height);
/* Use an existing hipStream_t (passed as void*) instead of the runtime's own stream. */
int sar_runtime_set_stream(sar_runtime* rt, void* hip_stream);
int sar_runtime_get_stream(const sar_runtime* rt, void** hip_stream_out);
/* The stream the read-backs of sar_colorize_format_async run on (made on first use). Runtimes that share a launch stream —
 * the frames of a batch — should share this one too (set it on the others; whoever made it must outlive them): a process
 * has few hardware queues, and every further stream shares one with somebody's kernels. */
int sar_runtime_get_copy_stream(sar_runtime* rt, void** hip_stream_out);
int sar_runtime_set_copy_stream(sar_runtime* rt, void* hip_stream);

/* ---- render (src/lib.rs:747-838) ------------------------------------------------------------------ */
/* Exactly `render`: ONE trajectory of cfg->iterations counted iterations after a start point drawn
 * from the runtime's stream and 1000 uncounted warm-up iterations. Accumulates into rt (no reset). */
int sar_render(const sar_config* cfg, sar_runtime* rt);
/*
 * The data-parallel form: equivalent to calling `render` cfg->jobs_total times on this un-reset
 * runtime (what one reference worker does, :956-988) with iterations = cfg->iterations / jobs_total
 * (floor; the job split of :1056-1058). Results are defined as the SEQUENTIAL result in job order
 * (job-major, iteration-minor ties). starts_xyz_host: jobs_total*3 doubles (pre-warm-up start points,
 * already scaled) or NULL to draw them from the runtime's stream.
 */
int sar_render_jobs(const sar_config* cfg, sar_runtime* rt, const double* starts_xyz_host);
/* Shard form: run only jobs [first_job, first_job+n_jobs) of the split above, each with
 * iters_per_job counted iterations. starts_xyz_host holds n_jobs*3 doubles for THIS slice (required). */
int sar_render_job_range(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs,
                         uint64_t iters_per_job, const double* starts_xyz_host);

/* The same with the start points already in device memory (n_jobs*3 doubles, same [job][xyz] layout, on the
 * runtime's device; read in stream order): nothing crosses PCIe, the call only enqueues. */
int sar_render_job_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs,
                                uint64_t iters_per_job, const double* starts_xyz_dev);

/* F frames of a sweep in ONE set of launches — F iterations of the CLI's frame loop (src/bin/main.rs:493-517: every frame a reset,
 * src/lib.rs:950-951, and a render_parallel of fresh jobs) at a time. Frame i is sar_render_jobs(cfgs[i], rts[i],
 * starts_xyz_host[i]), bit for bit; what changes is how the chip is filled: a frame of 65 536 jobs occupies a third of an MI355X,
 * so the frames' workgroups share ONE launch of each kernel (workgroup -> frame -> its argument block and buffers), the frames
 * dealt to the XCDs. It applies to distinct runtimes on one device with one image size (make them with sar_runtime_new_group) and
 * configs with the same jobs_total, iterations per job and scale whose jobs are resident at once (one launch chunk, at most
 * 4 Mpx); anything else — and n_frames == 1 — runs frame after frame, same result. The work is enqueued on rts[0]'s stream (a
 * runtime on another stream is ordered with it through events); the launch options (sar_runtime_set_option) are rts[0]'s.
 * starts_xyz_host[i] == NULL (or starts_xyz_host == NULL) draws frame i's points from rts[i]'s own stream. */
int sar_render_jobs_batch(uint32_t n_frames, const sar_config* const* cfgs, sar_runtime* const* rts,
                          const double* const* starts_xyz_host);
/* How many frames like cfg to render per batch: 1 when frames of this shape cannot share launches (the test sar_render_jobs_batch
 * makes — a caller then builds ONE runtime per lane, not a batch of them); otherwise a multiple of eight, 8..32, the smallest
 * whose last round of equally long wave pairs fills an XCD (eight pairs per CU; a frame takes one per 64 jobs that survive the
 * warm-up, by this runtime's last launch — before any has reported: 16). rt may be NULL: the answer for a runtime yet to be made. */
int sar_runtime_batch_frames(const sar_config* cfg, sar_runtime* rt, uint32_t* out_frames);

/* Announces the NEXT sar_render_job_range_device call on this runtime — these start points, job count and iterations per
 * job, and the attractor's 30 coefficients; the view, render kind and colours of cfg may differ in the announced call (the
 * warm-up is the map alone: a sweep's next frame only turns the view) — so that the 1000 uncounted
 * warm-up iterations of its jobs (src/lib.rs:750-752) can run ahead, on a second stream, under the accumulate / fold /
 * colorize tail of the frame in flight (the CLI's frame loop, src/bin/main.rs:493-517, knows the next frame while it
 * renders this one; sar_render_parallel announces its own next frame this way). Results do not depend on it: a call that
 * does not match the announcement simply runs its own warm-up. The start points must be in place in device memory when this is called and stay unchanged until the announced
 * call; a reset in between is fine. Enqueues only. */
int sar_runtime_prefetch_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs,
                                uint64_t iters_per_job, const double* starts_xyz_dev);

/* ---- colorize (src/lib.rs:841-904) ---------------------------------------------------------------- */
/* Writes width*height*4 uint16 (RGBA16, FinalImage layout :625) to host memory. */
int sar_colorize(const sar_config* cfg, sar_runtime* rt, uint16_t* rgba_out_host);
/* Same, leaving the image in device memory (width*height*8 bytes); stream-ordered, no host sync. */
int sar_colorize_device(const sar_config* cfg, sar_runtime* rt, void* rgba_out_dev);
/* n of them at once (frame i: cfgs[i], rts[i] -> rgba_out_dev[i]): ONE launch for Gas frames of one palette on one stream. */
int sar_colorize_device_batch(uint32_t n, const sar_config* const* cfgs, sar_runtime* const* rts, void* const* rgba_out_dev);

/* ---- attractor extent: the "first pass" the reference leaves as a TODO (src/lib.rs:326-333) ----------------- *
 * n_jobs trajectories (start points from starts_xyz_host[n_jobs*3], or from the runtime's stream when NULL), each
 * 1000 warm-up iterations (:750-752) then iters_per_job iterations. out12[0..6) = xmin,xmax,ymin,ymax,zmin,zmax of the
 * screen-space points (cfg's rotation matrix applied, :773: the quantities the comment at :329-333 lists and from
 * which View::center_camera is chosen), out12[6..12) the same for the raw points. Bounds move through `<` / `>` only
 * (NaN never moves one), so the result is independent of the execution order. Does not touch the runtime's buffers. */
int sar_runtime_extent(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                       const double* starts_xyz_host, double* out12);

/* ---- image export (src/bin/main.rs:40-100, write_image_matches) ------------------------------------ *
 * The CLI converts FinalImage (RGBA16) by (--transparent, --8bit) before it encodes (:52-57):
 *   (true,false) RGBA16 as is | (false,false) to_rgb16 | (true,true) to_rgba8 | (false,true) to_rgb8
 * and writes PNG (default compression, adaptive filter, :84-92), BMP (:71-77) or PAM (:64-70; both need --8bit,
 * :256-258). The conversions and encoders live in the `image` crate (Cargo.toml: image = "0.25", no lockfile,
 * not vendored): restated here from its published algorithm — 16 -> 8 bit is ((c + 128) / 257), alpha is
 * dropped without pre-multiplication — PARITY UNPINNED beyond "the file decodes to these samples".
 */
#define SAR_FMT_RGBA16 0
#define SAR_FMT_RGB16  1
#define SAR_FMT_RGBA8  2
#define SAR_FMT_RGB8   3
/* The format write_image_matches picks for (--transparent, --8bit) (:52-57). */
int sar_image_format(int transparent, int eight_bit);
/* Bytes of a width x height image in `format` (0 for an unknown format). */
size_t sar_image_bytes(int format, uint32_t width, uint32_t height);
/* RGBA16 (device) -> format (device), stream-ordered on rt's stream; in and out must not overlap. */
int sar_image_convert_device(sar_runtime* rt, const void* rgba16_dev, int format, void* out_dev);
/* colorize + conversion on the device, then ONE device-to-host copy of the converted image
 * (sar_image_bytes(format) bytes: 12 MiB instead of 32 MiB for RGB8 at 2048x2048). Samples are host-endian. */
int sar_colorize_format(const sar_config* cfg, sar_runtime* rt, int format, void* out_host);
/* The same, returning as soon as the work is ENQUEUED: the image is in out_host once sar_runtime_wait_image(rt, ticket) has returned
 * (sar_runtime_image_done asks without waiting). A `sequence` sweep (src/bin/main.rs:493-517 hands frame k to its writer threads
 * and goes on with frame k+1) reads frame k back while frame k+1 renders. out_host should be page-locked (sar_host_alloc; pageable
 * memory is staged by the HIP runtime and the call may block) and stay untouched until the ticket is done; the runtime may be reset
 * and rendered into again before that. out_host == NULL: colorize + conversion only — the image stays in device memory until
 * sar_runtime_read_image_async fetches it (page-locking a host image takes 1-3 ms, a frame renders in 0.7). */
int sar_colorize_format_async(const sar_config* cfg, sar_runtime* rt, int format, void* out_host, uint64_t* ticket_out);
int sar_runtime_read_image_async(sar_runtime* rt, void* out_host, uint64_t* ticket_out);
int sar_runtime_image_done(sar_runtime* rt, uint64_t ticket, int* done_out);
int sar_runtime_wait_image(sar_runtime* rt, uint64_t ticket);
/* Page-locked host memory for those read-backs (4 MiB and more: mapped with huge pages, touched, hipHostRegister'ed). */
int sar_host_alloc(size_t bytes, void** out);
int sar_host_free(void* p);
/* Announces `count` sar_host_alloc(bytes) calls to come (a sweep's ring of images): helper threads map and zero the blocks ahead —
 * most of what page-locking costs, and no HIP call — and sar_host_alloc only locks them. One announcement per process at a time; a
 * new one, or count 0, releases what the last one left. */
int sar_host_reserve(size_t bytes, uint32_t count);
/* Encoders (host only; no device needed). `pixels` is a host image in `format`, host-endian samples.
 * PNG: 8/16-bit RGB(A), zlib default compression, per-row adaptive filter (minimum sum of absolute differences).
 * BMP / PAM: SAR_FMT_RGBA8 or SAR_FMT_RGB8 only (the CLI requires --8bit for them); BMP 24 bpp BI_RGB or
 * 32 bpp BI_BITFIELDS (V4 header) bottom-up; PAM "P7" with TUPLTYPE RGB / RGB_ALPHA. */
int sar_write_png(const char* path, int format, uint32_t width, uint32_t height, const void* pixels);
int sar_write_bmp(const char* path, int format, uint32_t width, uint32_t height, const void* pixels);
int sar_write_pam(const char* path, int format, uint32_t width, uint32_t height, const void* pixels);

/* ---- read-back accessors (the reference keeps these fields private, :633-643) ---------------------- */
int sar_runtime_count(sar_runtime* rt, uint32_t* out_host);   /* width*height */
int sar_runtime_steps(sar_runtime* rt, double* out_host);     /* width*height */
int sar_runtime_zbuf(sar_runtime* rt, float* out_host);       /* width*height */
int sar_runtime_max(sar_runtime* rt, uint32_t* out_max);
/* Upload a full state (used to move a partial render between processes / devices). */
int sar_runtime_load(sar_runtime* rt, const uint32_t* count_host, const double* steps_host,
                     const float* zbuf_host, uint32_t max);

/* ---- multi-GPU exchange, one process per GPU: Runtime::merge (:708-738) folded in rank order (:1068-1076) -----------------------
 * ONE context object per runtime and world; the collectives (RCCL through torch.distributed, or anything else) are the caller's,
 * on buffers the caller owns (device memory the collective library can address); everything between them happens here.
 * Every rank OWNS the slice [rank*S, min(npix, (rank+1)*S)) of the image, S a multiple of 2048 (sar_exchange_slice_pixels):
 *
 *   sar_exchange_flags   flags[g] = 1 for every 64-pixel GRANULE of this rank's partial buffers with a count or a depth  -> all-gather (1 B / granule)
 *   sar_exchange_pack    from every rank's flags the library PLANS the exchange on the device (two block scans: where each of my
 *                        records goes, where each record I receive arrives) and packs the send buffer — records of 1 KiB
 *                        [count u32 x 64 | sortable(zbuf) u32 x 64 | steps f64 x 64], owner by owner (a frame touches a fifth of
 *                        its pixels: 21 % of the granules of BASELINE configs[1]); a rank without a record holds the reset state
 *                        there, the fold skips it, same result bit for bit. A frame whose flags cover more than dense_above of
 *                        the image — or flags_all == NULL — goes DENSE: `world` blocks of S*16 bytes [count x S | zbuf x S |
 *                        steps x S]. send_bytes / recv_bytes[world] are the split sizes of the all-to-all: the call waits for
 *                        them (2 world + 1 numbers), the ONE host wait of a frame's exchange                    -> all-to-all (split sizes)
 *   sar_exchange_merge   the owner folds what arrived in rank order (rank 0 = the accumulator of :1070, the earlier rank wins depth
 *                        ties, `max` follows every intermediate sum) into rt's own buffers at its slice, and writes {max, wrap
 *                        flag, depth range} as 4 x int64                                                         -> all-reduce MAX (32 B)
 *   sar_exchange_finish  the reduced scalars become the runtime's; sar_colorize_range_device on my slice         -> gather (8 B/px)
 *
 * After merge a runtime holds the merged frame only inside its own slice. strange_attractor_renderer_amd/distributed.py
 * (SlicedExchange) is this sequence around torch.distributed. */
typedef struct sar_exchange sar_exchange;
typedef struct sar_exchange_layout {
    uint32_t world, rank;
    uint32_t slice_pixels;    /* S */
    uint32_t first_px, n_px;  /* this rank's slice */
    uint32_t granules;        /* ceil(npix / 64): the bytes of one rank's flags */
    uint64_t block_bytes;     /* world * S * 16: the size of the send and of the receive buffer */
} sar_exchange_layout;
#define SAR_EXCHANGE_GRANULE 64
int sar_exchange_slice_pixels(uint32_t npix, uint32_t world, uint32_t* out_slice_pixels);   /* host arithmetic only */
/* rt is borrowed: every call on the exchange but sar_exchange_free needs it alive, at the image size it had here (a resized
 * runtime: SAR_ERR_DIM_MISMATCH, make a new context); layout_out may be NULL. */
int sar_exchange_new(sar_runtime* rt, uint32_t world, uint32_t rank, sar_exchange** out, sar_exchange_layout* layout_out);
int sar_exchange_free(sar_exchange* ex);
int sar_exchange_flags(sar_exchange* ex, uint8_t* flags_out_dev /* [granules] */);
int sar_exchange_pack(sar_exchange* ex, const uint8_t* flags_all_dev /* [world][granules], or NULL */, double dense_above,
                      void* send_dev /* block_bytes */, uint64_t* send_bytes /* [world] */, uint64_t* recv_bytes /* [world] */, int* sparse_out);
int sar_exchange_merge(sar_exchange* ex, const void* recv_dev /* block_bytes */, int64_t* scalars_out_dev /* [4] */);
int sar_exchange_finish(sar_exchange* ex, const int64_t* scalars_reduced_dev /* [4] */);
/* The ROOTED form, for a caller that wants the whole merged Runtime on one rank (20 B/px through two ring collectives):
 *   step 0  key[p] = sortable int64 of (zbuf[p], lowest-rank-wins)                                        -> all-reduce MAX
 *   step 1  sum[0..npix) = count[p] as int32 (wrapping == u32 add), sum[npix..3 npix) = the two int32 halves of steps[p] where this
 *           rank holds the winning key, else 0                                                            -> reduce SUM (int32)
 *   step 2  (the root) count / zbuf / steps / max of rt replaced by the reduced buffers. */
int sar_exchange_rooted(sar_exchange* ex, uint32_t step, void* key_i64_dev /* [npix] */, void* sum_i32_dev /* [3 npix]; step 0: NULL */);
/* colorize (:841-904) of the pixel range [first_px, first_px + n_px) into out_dev (n_px*8 bytes, RGBA16), using the
 * max / depth range the runtime's scalars hold (made global by step 3); stream-ordered. */
int sar_colorize_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t first_px, uint32_t n_px, void* rgba_out_dev);

/* ---- ParallelRenderer / render_parallel (src/lib.rs:908-1082) -------------------------------------- */
/* ParallelRenderer::new (:919-1004). `units` plays the role of num_threads (:920-922): the number of
 * execution units the job split divides by; 0 selects the device default, 64 per CU (16 384 on MI355X), so that the
 * CLI's default of 12 jobs per thread (src/bin/main.rs:305) becomes 196 608 trajectories = three waves per SIMD.
 * The renderer owns one runtime on `device`, seeded with `seed`. */
int sar_renderer_new(int device, uint32_t units, uint64_t seed, sar_renderer** out);
/* The same over SEVERAL GPUs of one node, behind this ABI alone (no Python, no RCCL): ParallelRenderer::new owns every execution
 * unit of the machine (:919-1004). devices[n_devices] are HIP device ordinals in FOLD ORDER (device 0 is the accumulator of :1070;
 * a device may be listed more than once — each entry is its own shard). units == 0: 64 per CU summed over the devices.
 * render_parallel then cuts the units*jobs_per_unit jobs into contiguous slices, one per device (one host thread + one stream
 * each), lets every device own one slice of the image, exchanges the partial buffers point-to-point (sar_renderer_set_exchange),
 * folds them with Runtime::merge in device order, colorizes each slice where it lives and copies it into rgba_out_host. The
 * result is bit-identical to the single-device renderer's for the same units. (Validated with one physical GPU listed several
 * times; a node with several GPUs has not been available to this build.) */
int sar_renderer_new_multi(const int* devices, uint32_t n_devices, uint32_t units, uint64_t seed, sar_renderer** out);
int sar_renderer_num_devices(const sar_renderer* r, uint32_t* out_devices);
int sar_renderer_num_units(const sar_renderer* r, uint32_t* out_units);
/* ParallelRenderer::shutdown (:1020-1025). */
int sar_renderer_shutdown(sar_renderer* r);
/* render_parallel (:1051-1082): iterations/units/jobs_per_unit per job (:1058), units*jobs_per_unit
 * jobs (:1062), reset, render, colorize. rgba_out_host: width*height*4 uint16. */
int sar_render_parallel(sar_renderer* r, const sar_config* cfg, uint32_t jobs_per_unit,
                        uint16_t* rgba_out_host);
/* The renderer's runtime (borrowed; device 0's), e.g. to read the count buffer after render_parallel. With several
 * devices the merged slices are first gathered into it (20 B/px over xGMI, once per frame, only when asked). */
int sar_renderer_runtime(sar_renderer* r, sar_runtime** out_borrowed);
/* Phases of the last sar_render_parallel on a multi-device renderer: the slowest device's stream time per phase. */
typedef struct sar_parallel_timing {
    float    total_ms;      /* host wall time of the call */
    float    render_ms;     /* reset + warm-up + iterate + accumulate + fold + pack */
    float    exchange_ms;   /* records / peer copies + merge of the owned slice + the scalar reduce (includes waiting for the slowest peer) */
    float    colorize_ms;   /* colorize of the own slice + its copy to the host image (from behind the scalar reduce) */
    uint32_t n_devices;
    uint32_t peer_access_failures;  /* ordered pairs of distinct devices WITHOUT direct peer access (hipDeviceCanAccessPeer said
                                       no, or hipDeviceEnablePeerAccess failed; sar_last_error keeps the last reason): their
                                       copies are staged through host memory by the HIP runtime */
    uint64_t exchange_bytes_per_device;  /* bytes every device pulls over xGMI per frame */
    float    host_ms_before_exchange;    /* host time between the last device's render being enqueued and the first pull of the
                                            exchange being enqueued (one device: entry to render enqueued) — the next frame's
                                            start points are drawn on helper threads meanwhile, off this path */
    float    host_ms_enqueue;            /* host time from entry until the whole frame (render, exchange, colorize, copies) is enqueued */
    float    draw_ahead_ms;              /* a helper thread's time to draw one device's slice of the next frame's start points */
    float    _pad;
} sar_parallel_timing;
int sar_renderer_last_timing(const sar_renderer* r, sar_parallel_timing* out);
/* How a multi-device renderer exchanges its partial buffers before colorize. 2 = sparse: every device writes the records of the
 * 64-pixel granules it has touched (1 KiB each: count, zbuf, steps) straight into their owners' buffers — kernels storing to
 * peer memory over xGMI — and the owners fold what arrived (a fifth of a frame at the BASELINE shapes); 1 = dense: whole slices,
 * 16 B/px, by hipMemcpyPeerAsync; 0 (default) = sparse when every pair of devices has direct peer access AND every device could
 * give its receive buffers fine-grained (device-coherent) memory, dense otherwise — where either is missing, a render call under
 * mode 2 fails with SAR_ERR_INVALID instead of folding what may be stale. The merged frame is the same bit for bit. */
int sar_renderer_set_exchange(sar_renderer* r, uint32_t mode);

/* ---- measurement ----------------------------------------------------------------------------------- */
int sar_runtime_enable_timing(sar_runtime* rt, int enabled);
int sar_runtime_last_timing(sar_runtime* rt, sar_timing* out);
/* What the last render call launched, as one line of text for logs and bench records — e.g.
 * "k_iterate_split R=60 bins=128x32768px interleaved hints=f32 pipe=2 | k_bin_accumulate splits=4 lists=4 counters=u32 |
 * chunks=1 warmup_ahead=19": the iterate kernel the library really chose (not a guess of the caller), its chunk size and
 * bin geometry, the accumulate mode, the launch chunks of the call and how many render calls so far found their warm-up
 * already done (sar_runtime_prefetch_device). Writes at most cap bytes including the terminating 0. */
int sar_runtime_describe_last_launch(const sar_runtime* rt, char* out, size_t cap);
/* Options by name (value 0 restores the default):
 *   "block_threads"      lanes per workgroup of the iterate kernel (64, 128, 192, 256)
 *   "checkpoint_stride"  iterations between trajectory checkpoints used by the payload resolve (default 32)
 *   "hint_bits"          per-XCD depth hints of the iterate kernel: 16 (fixed point over the depth range the warm-up saw) or
 *                        32 (the depth itself as f32); 0 = by image size
 *   "split_waves"        the iterate kernel as producer / consumer wave pairs: 1 never, 2 wherever the kernel exists; 0 = 2 for
 *                        launches whose jobs are all resident at once (512 per CU), 1 for larger ones
 *   "timing_accumulate"  1: the spans of successive render calls add up (sar_timing sums, iterate_launches counts
 *                        them) until sar_runtime_last_timing reads and clears them; 0: last render call only
 * Everything else a laboratory wants to turn — accumulate path, bin geometry, chunk sizes, hint layout, launch-chunk caps,
 * the batched launch's variants — is NOT in this library: include/sar_test_hooks.h declares sar_runtime_set_test_option, which
 * only the hooks build of the test-suite links (tests/hooks/libsar_hip_hooks.so: the same object files plus that one function).
 *
 * Jobs of more than 2^32-2 iterations (Config::iterations is a usize, :267): a launch orders its visits with a 32-bit
 * ordinal, so such a job runs as successive launches that hand its state on — and it runs them ALONE, one lane of the
 * chip at ~1e6 iterations per second: the reference's tie rule is job-major (an earlier JOB wins an exact depth tie
 * whatever the iteration), and two jobs advancing through their segments side by side would fold a later job's early
 * visit before an earlier job's late one. Correct, but ~1e5 times slower than the same iterations cut into more jobs;
 * use jobs_total / jobs_per_unit so that a job stays below 2^32-2 iterations. */
int sar_runtime_set_option(sar_runtime* rt, const char* name, uint64_t value);
/* Diagnostic (host arithmetic only, no device needed): the pixel -> (bin, 16-bit record) map the LDS-binned path uses for
 * a width x height image with "bin_shift" / "bin_interleave" as given (0 = the defaults of a runtime without forced
 * options and 131072 jobs). out = {ok, bins, bin_shift, interleaved, seg_shift, bin_bits, hi_shift, low_mask}:
 *   bin = (idx >> seg_shift) & ((1 << bin_bits) - 1);  record = (idx & low_mask) | ((idx >> hi_shift) & ~low_mask)
 *   idx = (record & low_mask) | (bin << seg_shift) | ((record & ~low_mask) << hi_shift)
 * ok = 0: the image has no binned geometry (the one-atomic-per-visit path renders it). */
int sar_bin_geometry(uint32_t width, uint32_t height, uint32_t bin_shift, uint32_t bin_interleave, uint32_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* SAR_H */
