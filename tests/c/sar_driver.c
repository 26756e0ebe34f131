/* sar_driver.c — the call sequence of bindings/rust-safe (GpuRuntime / render / colorize / GpuRenderer /
 * render_parallel, i.e. the reference CLI's two code paths, src/bin/main.rs:483-517) from a COMPILED C99 program over
 * include/sar.h: what a cgo / Rust / JNI host does, with no Python in the process. tests/test_c_program.py builds it
 * with gcc, runs it and compares the files it writes with the oracle.
 *
 *   sar_driver <out_dir> <width> <height> <jobs> <iters_per_job> <seed> <n_shards>
 * exit code: 0 ok, 3 no HIP device (SAR_ERR_NO_DEVICE surfaced as a status, nothing crashed), 1 anything else. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sar.h"

static int fail(const char* what, int st) {
    fprintf(stderr, "%s: %s (%d): %s\n", what, sar_status_string(st), st, sar_last_error());
    return st == SAR_ERR_NO_DEVICE ? 3 : 1;
}
#define CHECK(call) do { int st_ = (call); if (st_ != SAR_OK) return fail(#call, st_); } while (0)

static int dump(const char* dir, const char* name, const void* p, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    const size_t w = fwrite(p, 1, bytes, f);
    fclose(f);
    return w == bytes ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: %s out_dir width height jobs iters_per_job seed n_shards\n", argv[0]); return 1; }
    const char* dir = argv[1];
    const uint32_t W = (uint32_t)atoi(argv[2]), H = (uint32_t)atoi(argv[3]), jobs = (uint32_t)atoi(argv[4]);
    const uint64_t n = (uint64_t)atoll(argv[5]), seed = (uint64_t)atoll(argv[6]);
    const uint32_t shards = (uint32_t)atoi(argv[7]);
    if (sar_abi_version() != SAR_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (strlen(sar_build_id()) != 16) { fprintf(stderr, "no build id\n"); return 1; }
    const size_t npix = (size_t)W * H;

    /* Config { iterations, width, height, ..Config::poisson_saturne() } (src/lib.rs:9-15) */
    sar_config cfg;
    CHECK(sar_config_poisson_saturne(&cfg));
    cfg.width = W; cfg.height = H; cfg.transparent = 0; cfg.seed = seed;
    cfg.jobs_total = jobs; cfg.iterations = (uint64_t)jobs * n;
    CHECK(sar_config_validate(&cfg));

    /* --single-thread shape (main.rs:483-491): Runtime::new, render, colorize, reset — with `jobs` trajectories */
    sar_runtime* rt = NULL;
    CHECK(sar_runtime_new(&cfg, 0, &rt));
    CHECK(sar_render_jobs(&cfg, rt, NULL));            /* start points from the runtime's stream (seed) */
    uint32_t* count = malloc(npix * 4);
    uint16_t* rgba = malloc(npix * 8);
    double* steps = malloc(npix * 8);
    float* zbuf = malloc(npix * 4);
    uint32_t mx = 0;
    if (!count || !rgba || !steps || !zbuf) return 1;
    CHECK(sar_colorize(&cfg, rt, rgba));
    CHECK(sar_runtime_count(rt, count));
    CHECK(sar_runtime_steps(rt, steps));
    CHECK(sar_runtime_zbuf(rt, zbuf));
    CHECK(sar_runtime_max(rt, &mx));
    if (dump(dir, "count.bin", count, npix * 4) || dump(dir, "rgba.bin", rgba, npix * 8) || dump(dir, "steps.bin", steps, npix * 8) ||
        dump(dir, "zbuf.bin", zbuf, npix * 4) || dump(dir, "max.bin", &mx, 4)) return 1;
    CHECK(sar_runtime_reset(rt));
    CHECK(sar_runtime_count(rt, count));
    for (size_t k = 0; k < npix; ++k) if (count[k]) { fprintf(stderr, "reset left a count\n"); return 1; }
    CHECK(sar_runtime_free(rt));

    /* a `sequence` sweep's frames (main.rs:493-517 renders them one after the other) through ONE set of launches:
     * sar_render_jobs_batch — three frames, each with its own Runtime (of ONE frame group), view angle and start-point stream */
    enum { FRAMES = 3 };
    sar_config fc[FRAMES];
    const sar_config* fcp[FRAMES];
    sar_runtime* fr[FRAMES];
    for (int i = 0; i < FRAMES; ++i) {
        fc[i] = cfg;
        fc[i].angle = 0.3 * i;
        fc[i].seed = seed + 1u + (uint64_t)i;
        fcp[i] = &fc[i];
    }
    uint32_t advice = 0;
    CHECK(sar_runtime_batch_frames(&fc[0], NULL, &advice));       /* before any runtime exists: 1 = frame by frame */
    if (advice < 1u) { fprintf(stderr, "sar_runtime_batch_frames said %u\n", advice); return 1; }
    CHECK(sar_runtime_new_group(&fc[0], 0, FRAMES, fr));           /* one stream, one allocation per runtime */
    for (int i = 0; i < FRAMES; ++i) CHECK(sar_runtime_seed(fr[i], fc[i].seed));
    CHECK(sar_render_jobs_batch(FRAMES, fcp, (sar_runtime* const*)fr, NULL));
    /* the sweep's read-back in two steps: the batch's conversions at once (device memory only), then every frame into a
     * page-locked image, polled without waiting (sar_runtime_image_done) and waited for at the end */
    const int fmt = sar_image_format(0, 1);                        /* --8bit, not transparent: RGB8 (main.rs:52-57) */
    const size_t img_bytes = sar_image_bytes(fmt, W, H);
    void* img[FRAMES];
    uint64_t ticket[FRAMES];
    for (int i = 0; i < FRAMES; ++i) CHECK(sar_colorize_format_async(&fc[i], fr[i], fmt, NULL, NULL));
    for (int i = 0; i < FRAMES; ++i) {
        CHECK(sar_host_alloc(img_bytes, &img[i]));
        CHECK(sar_runtime_read_image_async(fr[i], img[i], &ticket[i]));
    }
    for (int i = 0; i < FRAMES; ++i) {
        char name[64];
        int done = 0;
        CHECK(sar_runtime_image_done(fr[i], ticket[i], &done));    /* 0 or 1: never blocks */
        CHECK(sar_runtime_wait_image(fr[i], ticket[i]));
        CHECK(sar_runtime_image_done(fr[i], ticket[i], &done));
        if (!done) { fprintf(stderr, "a waited-for image is not done\n"); return 1; }
        snprintf(name, sizeof name, "rgb8_batch_%d.bin", i);
        if (dump(dir, name, img[i], img_bytes)) return 1;
        CHECK(sar_host_free(img[i]));
    }
    uint64_t sums[FRAMES];
    for (int i = 0; i < FRAMES; ++i) {
        char name[64];
        CHECK(sar_runtime_count(fr[i], count));
        CHECK(sar_checksum_fnv1a64(count, npix * 4, &sums[i]));
        CHECK(sar_colorize(&fc[i], fr[i], rgba));
        snprintf(name, sizeof name, "count_batch_%d.bin", i);
        if (dump(dir, name, count, npix * 4)) return 1;
        snprintf(name, sizeof name, "rgba_batch_%d.bin", i);
        if (dump(dir, name, rgba, npix * 8)) return 1;
        CHECK(sar_runtime_free(fr[i]));
    }
    if (dump(dir, "count_batch_fnv.bin", sums, sizeof sums)) return 1;

    /* default shape (main.rs:493-517): ParallelRenderer::new, render_parallel, shutdown — over `shards` shards */
    int devices[16];
    for (uint32_t k = 0; k < shards && k < 16; ++k) devices[k] = 0;
    sar_renderer* r = NULL;
    if (shards > 1) CHECK(sar_renderer_new_multi(devices, shards, jobs, seed, &r));   /* units = jobs, 1 job per unit */
    else CHECK(sar_renderer_new(0, jobs, seed, &r));
    CHECK(sar_render_parallel(r, &cfg, 1, rgba));
    if (dump(dir, "rgba_parallel.bin", rgba, npix * 8)) return 1;
    sar_runtime* borrowed = NULL;
    CHECK(sar_renderer_runtime(r, &borrowed));
    CHECK(sar_runtime_count(borrowed, count));
    if (dump(dir, "count_parallel.bin", count, npix * 4)) return 1;
    if (shards > 1) {   /* the next frame of the renderer through the dense exchange (the first went through the default: sparse records) */
        uint16_t* again = malloc(npix * 8);
        if (!again) return 1;
        CHECK(sar_renderer_set_exchange(r, 1u));
        CHECK(sar_render_parallel(r, &cfg, 1, again));
        if (dump(dir, "rgba_parallel_dense.bin", again, npix * 8)) return 1;
        free(again);
    }
    CHECK(sar_renderer_shutdown(r));
    free(count); free(rgba); free(steps); free(zbuf);
    puts("ok");
    return 0;
}
