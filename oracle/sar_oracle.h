/*
 * sar_oracle.h — CPU oracle for the iterate/accumulate path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (strange_attractor_renderer_amd/) never links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's arithmetic (Icelk/strange-attractor-renderer,
 * src/lib.rs), op for op, to be built with -ffp-contract=off (Rust/LLVM never contracts a*b+c).
 *
 * PARITY STATUS: the reference is Rust and cannot be built here (no cargo/rustc, no lockfile, no
 * vendored crates), and its own test-suite holds no vectors for this path (the only `cargo test`
 * item is a doc-test that constructs a Config, src/lib.rs:9-15). Bit-level parity is therefore
 * UNPINNED; the oracle is pinned (a) by the only numbers the reference's source holds for this
 * path, the screen-space extent of poisson-saturne in the comment at src/lib.rs:329-333, which
 * sar_oracle_extent reproduces to < 5e-5 (tests/test_oracle_kat.py) — they cover the coefficients,
 * the rotation matrix and screen_space; (b) statistically against the one artefact the reference
 * ships, media/poisson-saturne.png (tests/golden/ref_png_stats.json,
 * tests/test_oracle_reference_png.py); (c) by known-answer vectors produced independently
 *  and frozen under tests/golden/.
 */
#ifndef SAR_ORACLE_H
#define SAR_ORACLE_H

#include <stdint.h>
#include "../include/sar.h" /* sar_config POD only */

#ifdef __cplusplus
extern "C" {
#endif

/* Runtime, src/lib.rs:631-646 (rng kept outside: start points are explicit here). */
typedef struct sar_oracle_runtime {
    uint32_t width, height;
    uint32_t max;
    uint32_t _pad;
    uint32_t* count; /* width*height */
    double*   steps;
    float*    zbuf;
} sar_oracle_runtime;

/* PolynomialSprott2Degree::next_point, src/lib.rs:583-621 */
void sar_oracle_next_point(const sar_config* cfg, const double p[3], double out[3]);
/* EulerAxisRotation::to_rotation_matrix (release: axis not normalised), src/lib.rs:176-196 */
void sar_oracle_rotation_matrix(const sar_config* cfg, double m[9]);
/* color transforms, src/lib.rs:507-516 and 520-558 */
double sar_oracle_color_transform(const sar_config* cfg, const double delta[3], const double ss[3]);

/* Runtime::new / reset / merge, src/lib.rs:660-665, 682-699, 708-738 */
sar_oracle_runtime* sar_oracle_runtime_new(uint32_t width, uint32_t height);
void sar_oracle_runtime_free(sar_oracle_runtime* rt);
void sar_oracle_runtime_reset(sar_oracle_runtime* rt);
int  sar_oracle_runtime_merge(sar_oracle_runtime* dst, const sar_oracle_runtime* src);

/* render, src/lib.rs:747-838, with the start point (pre-warm-up, already scaled by 0.1) explicit. */
void sar_oracle_render(const sar_config* cfg, sar_oracle_runtime* rt, const double p0[3],
                       uint64_t iterations);
/* `jobs` sequential render calls on one un-reset runtime (one worker's loop, src/lib.rs:956-988). */
void sar_oracle_render_jobs(const sar_config* cfg, sar_oracle_runtime* rt, const double* starts_xyz,
                            uint32_t jobs, uint64_t iters_per_job);
/* Iterates only (no accumulation): writes the point after `n` applications of next_point. */
void sar_oracle_iterate(const sar_config* cfg, const double p0[3], uint64_t n, double out[3]);

/* The "first pass" the reference leaves as a TODO (src/lib.rs:326-333): extent of the attractor. For every job:
 * 1000 warm-up iterations (:750-752), then `iters_per_job` iterations; out[0..6) = xmin,xmax,ymin,ymax,zmin,zmax of
 * the SCREEN-SPACE points (rotation matrix applied, :773 — the numbers the comment at :329-333 lists), out[6..12) the
 * same for the raw points. A coordinate updates a bound through `<` / `>` only, so NaN never does. */
void sar_oracle_extent(const sar_config* cfg, const double* starts_xyz, uint32_t jobs, uint64_t iters_per_job,
                       double out[12]);

/* Palette::interpolate, src/lib.rs:442-472 */
void sar_oracle_palette(const sar_config* cfg, double value, double rgb[3]);
/* colorize, src/lib.rs:841-904; rgba: width*height*4 uint16 */
void sar_oracle_colorize(const sar_config* cfg, const sar_oracle_runtime* rt, uint16_t* rgba);

/* write_image_matches' format conversion, src/bin/main.rs:52-57 (DynamicImage::to_rgb16 / to_rgba8 / to_rgb8 of the
 * `image` crate 0.25 — not vendored, version unpinned: its published u16 -> u8 rule ((c + 128) / 257) is restated;
 * parity unpinned). format: SAR_FMT_*; out holds npix * channels samples of 1 or 2 bytes, host-endian. */
void sar_oracle_convert(int format, uint64_t npix, const uint16_t* rgba16, void* out);

/* Start-point stream (defined by this project, see include/sar.h sar_start_points). */
void sar_oracle_start_points(uint64_t seed, uint64_t first_job, uint32_t n_jobs, double* xyz);
/* the stream's pieces, for the tests that hold them to their published vectors */
void sar_oracle_splitmix64(uint64_t seed, uint32_t n, uint64_t* out);          /* n outputs of SplitMix64 from state `seed` */
void sar_oracle_xoshiro256pp(uint64_t state[4], uint32_t n, uint64_t* out);    /* n outputs of xoshiro256++; state advanced */
void sar_oracle_xoshiro256_jump(uint64_t state[4]);                            /* the published jump(): 2^128 steps */
double sar_oracle_unit_f64(uint64_t raw);                                      /* (raw >> 11) * 2^-53 */

/* FNV-1a 64 over raw bytes (fixture hashing). */
uint64_t sar_oracle_fnv1a64(const void* data, uint64_t nbytes);

/*
 * CPU baseline shaped like render_parallel (src/lib.rs:1051-1082, pool :919-1004): `threads`
 * workers with private runtimes, a shared atomic job counter of threads*jobs_per_thread jobs of
 * iterations/threads/jobs_per_thread iterations each, then SERIAL merge and SERIAL colorize.
 * Returns wall seconds; counted iterations in *iters_done. rgba may be NULL.
 */
double sar_oracle_render_parallel(const sar_config* cfg, uint32_t threads, uint32_t jobs_per_thread,
                                  uint64_t seed, uint16_t* rgba, uint64_t* iters_done,
                                  sar_oracle_runtime* out_merged /* nullable, pre-allocated */);

/* sar_oracle_render_jobs on `threads` host threads with the SAME result bit for bit: contiguous job slices into
 * private runtimes, folded with Runtime::merge in slice order (earlier job wins depth ties, like the sequential
 * strict `>`). Lets the full-size BASELINE frames (1e9 iterations) meet the oracle in seconds. 0 on success. */
int sar_oracle_render_jobs_mt(const sar_config* cfg, sar_oracle_runtime* rt, const double* starts_xyz, uint32_t jobs,
                              uint64_t iters_per_job, uint32_t threads);

#ifdef __cplusplus
}
#endif
#endif
