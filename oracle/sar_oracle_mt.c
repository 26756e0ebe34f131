/*
 * sar_oracle_mt.c — CPU baseline shaped like the reference's render_parallel. TEST/BENCH ONLY.
 *
 * Restates the structure of ParallelRenderer + render_parallel (src/lib.rs:908-1082) with pthreads:
 *   - `threads` workers, each owning a private runtime (16 B/pixel, :938) that it resets (:950-951);
 *   - iterations per job = N / threads / jobs_per_thread (two floor divisions, :1058);
 *   - a shared atomic counter of threads*jobs_per_thread jobs (:1062) that workers decrement
 *     (:962-982), each job being one `render` with a fresh start point and 1000 warm-ups (:987);
 *   - finished runtimes are handed to the caller in ARRIVAL order (mpsc, :990, :1011-1013); the
 *     caller takes the first as accumulator (:1070) and merges the rest serially (:1072-1076);
 *   - serial colorize (:1080).
 * The reference binary itself cannot be built here (no Rust toolchain); this is the "port" CPU
 * baseline bench.py reports, never a product path.
 */
#define _POSIX_C_SOURCE 200809L /* pthread_barrier_t, clock_gettime under -std=c11 */
#include "sar_oracle.h"

#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <time.h>

typedef struct mt_shared {
    const sar_config* cfg;
    uint64_t iters_per_job;
    atomic_long jobs_left;
    atomic_ulong next_job; /* index into the start-point table (the reference draws per-thread RNG) */
    const double* starts;
    /* arrival queue */
    pthread_mutex_t mu;
    pthread_cond_t cv;
    uint32_t arrived;
    uint32_t* arrival_order;
} mt_shared;

typedef struct mt_worker {
    mt_shared* sh;
    sar_oracle_runtime* rt;
    uint32_t index;
} mt_worker;

static void* worker_main(void* arg) {
    mt_worker* w = (mt_worker*)arg;
    mt_shared* sh = w->sh;
    sar_oracle_runtime_reset(w->rt); /* :950-951 */
    for (;;) {
        long left = atomic_load(&sh->jobs_left);
        int took = 0;
        while (left > 0) { /* fetch_update, :962-982 */
            if (atomic_compare_exchange_weak(&sh->jobs_left, &left, left - 1)) {
                took = 1;
                break;
            }
        }
        if (!took) break;
        const unsigned long job = atomic_fetch_add(&sh->next_job, 1ul);
        sar_oracle_render(sh->cfg, w->rt, sh->starts + 3 * job, sh->iters_per_job); /* :987 */
    }
    pthread_mutex_lock(&sh->mu); /* sender.send(...), :990 */
    sh->arrival_order[sh->arrived++] = w->index;
    pthread_cond_signal(&sh->cv);
    pthread_mutex_unlock(&sh->mu);
    return NULL;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double sar_oracle_render_parallel(const sar_config* cfg, uint32_t threads, uint32_t jobs_per_thread,
                                  uint64_t seed, uint16_t* rgba, uint64_t* iters_done,
                                  sar_oracle_runtime* out_merged) {
    if (threads == 0 || jobs_per_thread == 0) return -1.0;
    const uint64_t total_jobs = (uint64_t)threads * jobs_per_thread;
    mt_shared sh;
    sh.cfg = cfg;
    sh.iters_per_job = cfg->iterations / threads / jobs_per_thread; /* :1058 */
    atomic_init(&sh.jobs_left, (long)total_jobs);                   /* :1062 */
    atomic_init(&sh.next_job, 0ul);
    double* starts = (double*)malloc(sizeof(double) * 3 * total_jobs);
    sar_oracle_start_points(seed, 0, (uint32_t)total_jobs, starts);
    sh.starts = starts;
    pthread_mutex_init(&sh.mu, NULL);
    pthread_cond_init(&sh.cv, NULL);
    sh.arrived = 0;
    sh.arrival_order = (uint32_t*)malloc(sizeof(uint32_t) * threads);

    /* The pool and its per-thread textures exist before render_parallel is called in the reference
     * (ParallelRenderer::new, :919); allocation is therefore outside the timed region here too. */
    mt_worker* ws = (mt_worker*)malloc(sizeof(mt_worker) * threads);
    pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (uint32_t t = 0; t < threads; ++t) {
        ws[t].sh = &sh;
        ws[t].index = t;
        ws[t].rt = sar_oracle_runtime_new(cfg->width, cfg->height);
    }

    const double t0 = now_s();
    for (uint32_t t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, worker_main, &ws[t]);

    /* receive in arrival order and fold (:1068-1076) */
    sar_oracle_runtime* acc = NULL;
    for (uint32_t got = 0; got < threads; ++got) {
        pthread_mutex_lock(&sh.mu);
        while (sh.arrived <= got) pthread_cond_wait(&sh.cv, &sh.mu);
        const uint32_t who = sh.arrival_order[got];
        pthread_mutex_unlock(&sh.mu);
        if (!acc) acc = ws[who].rt;
        else sar_oracle_runtime_merge(acc, ws[who].rt);
    }
    if (rgba) sar_oracle_colorize(cfg, acc, rgba); /* :1080 */
    const double t1 = now_s();

    for (uint32_t t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
    if (iters_done) *iters_done = sh.iters_per_job * total_jobs;
    if (out_merged && out_merged->width == acc->width && out_merged->height == acc->height) {
        const size_t n = (size_t)acc->width * acc->height;
        for (size_t k = 0; k < n; ++k) {
            out_merged->count[k] = acc->count[k];
            out_merged->steps[k] = acc->steps[k];
            out_merged->zbuf[k] = acc->zbuf[k];
        }
        out_merged->max = acc->max;
    }
    for (uint32_t t = 0; t < threads; ++t) sar_oracle_runtime_free(ws[t].rt);
    free(ws);
    free(tids);
    free(starts);
    free(sh.arrival_order);
    pthread_mutex_destroy(&sh.mu);
    pthread_cond_destroy(&sh.cv);
    return t1 - t0;
}

/* ---- deterministic multi-threaded form of sar_oracle_render_jobs (full-size parity tests) -------------------------
 * The result of `jobs` sequential render calls on one runtime (src/lib.rs:956-988) equals: cut the job list into
 * `threads` CONTIGUOUS slices, render slice t into its own runtime (one reference worker's loop), then fold
 * Runtime::merge (:708-738) over the slices in slice order — count is a sum, zbuf a max, and `self wins ties` (:728)
 * keeps the payload of the EARLIER job exactly like the strict `>` of the sequential depth test (:821). The fold is
 * done per pixel range on all threads (the per-pixel rule is :716-735 verbatim; the walk order does not enter). */
typedef struct det_worker {
    const sar_config* cfg;
    sar_oracle_runtime* rt;     /* this slice's runtime (slice 0 renders straight into the output) */
    sar_oracle_runtime** all;   /* every slice's runtime, in slice order */
    const double* starts;
    uint32_t first, count, threads, index;
    uint64_t iters;
    pthread_barrier_t* bar;
    uint32_t max_seen;
} det_worker;

static void* det_main(void* arg) {
    det_worker* w = (det_worker*)arg;
    sar_oracle_render_jobs(w->cfg, w->rt, w->starts + 3 * (size_t)w->first, w->count, w->iters);
    pthread_barrier_wait(w->bar);
    sar_oracle_runtime* dst = w->all[0];
    const size_t npix = (size_t)dst->width * dst->height;
    const size_t lo = npix * w->index / w->threads, hi = npix * (w->index + 1) / w->threads;
    uint32_t mx = 0;
    for (uint32_t t = 1; t < w->threads; ++t) {
        const sar_oracle_runtime* src = w->all[t];
        for (size_t k = lo; k < hi; ++k) {
            dst->count[k] += src->count[k];                    /* :719, wrapping */
            if (dst->count[k] > mx) mx = dst->count[k];        /* :721-723 */
            if (src->zbuf[k] > dst->zbuf[k]) {                 /* :728, strict: dst (earlier slice) wins ties */
                dst->steps[k] = src->steps[k];
                dst->zbuf[k] = src->zbuf[k];
            }
        }
    }
    w->max_seen = mx;
    return NULL;
}

int sar_oracle_render_jobs_mt(const sar_config* cfg, sar_oracle_runtime* rt, const double* starts_xyz, uint32_t jobs,
                              uint64_t iters_per_job, uint32_t threads) {
    if (threads == 0) threads = 1;
    if (threads > jobs) threads = jobs ? jobs : 1;
    if (threads == 1) {
        sar_oracle_render_jobs(cfg, rt, starts_xyz, jobs, iters_per_job);
        return 0;
    }
    det_worker* ws = (det_worker*)calloc(threads, sizeof(det_worker));
    pthread_t* tids = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    sar_oracle_runtime** all = (sar_oracle_runtime**)calloc(threads, sizeof(*all));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, threads);
    int ok = 1;
    all[0] = rt;
    for (uint32_t t = 1; t < threads && ok; ++t) {
        all[t] = sar_oracle_runtime_new(rt->width, rt->height);
        if (!all[t]) ok = 0;
        else sar_oracle_runtime_reset(all[t]);
    }
    if (ok) {
        const uint32_t base = jobs / threads, rem = jobs % threads;
        uint32_t first = 0;
        for (uint32_t t = 0; t < threads; ++t) {
            ws[t].cfg = cfg; ws[t].rt = all[t]; ws[t].all = all; ws[t].starts = starts_xyz;
            ws[t].first = first; ws[t].count = base + (t < rem ? 1u : 0u); ws[t].threads = threads; ws[t].index = t;
            ws[t].iters = iters_per_job; ws[t].bar = &bar;
            first += ws[t].count;
        }
        for (uint32_t t = 0; t < threads; ++t) pthread_create(&tids[t], NULL, det_main, &ws[t]);
        for (uint32_t t = 0; t < threads; ++t) pthread_join(tids[t], NULL);
        for (uint32_t t = 0; t < threads; ++t) if (ws[t].max_seen > rt->max) rt->max = ws[t].max_seen;
    }
    for (uint32_t t = 1; t < threads; ++t) if (all[t]) sar_oracle_runtime_free(all[t]);
    pthread_barrier_destroy(&bar);
    free(all); free(tids); free(ws);
    return ok ? 0 : -1;
}
