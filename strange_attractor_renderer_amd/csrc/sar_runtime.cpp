// sar_runtime.cpp — the C ABI over the HIP kernels: Runtime, render, merge, colorize, the
// ParallelRenderer mirror and the multi-GPU exchange helpers (include/sar.h).
//
// Host logic only (allocation, chunking, argument blocks, stream ordering); all arithmetic on image
// data happens in the kernel files (sar_iterate.hip, sar_accumulate.hip, sar_image.hip). There is no CPU fallback: without a HIP device every entry point
// that touches a runtime returns SAR_ERR_NO_DEVICE.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "sar_launch.hpp"
#include "sar_runtime_impl.hpp"

using namespace sar;

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? SAR_ERR_OOM : SAR_ERR_HIP;           \
        }                                                                             \
    } while (0)

#define SAR_TRY(expr)                    \
    do {                                 \
        int s_ = (expr);                 \
        if (s_ != SAR_OK) return s_;     \
    } while (0)

namespace {

// ln(k+1) for k < kLnLutEntries, computed once per process with the host libm — the same function
// the oracle (and the reference, through Rust's f64::ln) calls on this machine.
const double* host_ln_lut() {
    static std::vector<double> lut;
    static std::once_flag once;
    std::call_once(once, [] {
        lut.resize(kLnLutEntries);
        for (uint32_t k = 0; k < kLnLutEntries; ++k) lut[k] = std::log(static_cast<double>(k + 1u));
    });
    return lut.data();
}

}  // namespace

namespace {

int free_device_buffers(sar_runtime* rt) {
    if (rt->d_count) hipFree(rt->d_count);
    if (rt->d_key) hipFree(rt->d_key);
    if (rt->d_steps) hipFree(rt->d_steps);
    if (rt->d_scratch_count) hipFree(rt->d_scratch_count);
    if (rt->d_scratch_key) hipFree(rt->d_scratch_key);
    if (rt->d_rgba) hipFree(rt->d_rgba);
    if (rt->d_export) hipFree(rt->d_export);
    rt->d_export = nullptr;
    if (rt->d_ztmp) hipFree(rt->d_ztmp);
    if (rt->d_zhint) hipFree(rt->d_zhint);
    rt->d_zhint = nullptr;
    rt->d_count = nullptr;
    rt->d_key = nullptr;
    rt->d_steps = nullptr;
    rt->d_scratch_count = nullptr;
    rt->d_scratch_key = nullptr;
    rt->d_rgba = nullptr;
    rt->d_ztmp = nullptr;
    rt->copies = 0;
    rt->key_copies = 0;
    return SAR_OK;
}

int alloc_image_buffers(sar_runtime* rt, uint32_t w, uint32_t h) {
    const uint64_t npix64 = static_cast<uint64_t>(w) * h;
    if (w == 0 || h == 0) { set_error("zero image dimension"); return SAR_ERR_INVALID; }
    if (npix64 > 0x7fffffffull) { set_error("width*height exceeds 2^31-1"); return SAR_ERR_RANGE; }
    // allocate first, commit on full success: a failed grow leaves the runtime as it was (old size, old buffers)
    uint32_t* count = nullptr;
    unsigned long long* key = nullptr;
    double* steps = nullptr;
    hipError_t e = hipMalloc(&count, npix64 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&key, npix64 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&steps, npix64 * sizeof(double));
    if (e != hipSuccess) {
        if (count) hipFree(count);
        if (key) hipFree(key);
        if (steps) hipFree(steps);
        set_error("image buffers for %ux%u: %s", w, h, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? SAR_ERR_OOM : SAR_ERR_HIP;
    }
    free_device_buffers(rt);
    rt->W = w;
    rt->H = h;
    rt->npix = static_cast<uint32_t>(npix64);
    rt->d_count = count;
    rt->d_key = key;
    rt->d_steps = steps;
    return SAR_OK;
}

int ensure_scratch(sar_runtime* rt, uint32_t copies, uint32_t key_copies) {
    if (rt->copies != copies || !rt->d_scratch_count) {
        if (rt->d_scratch_count) hipFree(rt->d_scratch_count);
        rt->d_scratch_count = nullptr;
        rt->copies = 0;
        const size_t n = static_cast<size_t>(copies) * rt->npix;
        HIP_TRY(hipMalloc(&rt->d_scratch_count, n * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(rt->d_scratch_count, 0, n * sizeof(uint32_t), rt->stream));
        rt->copies = copies;
    }
    if (rt->key_copies != key_copies || !rt->d_scratch_key) {
        if (rt->d_scratch_key) hipFree(rt->d_scratch_key);
        rt->d_scratch_key = nullptr;
        rt->key_copies = 0;
        const size_t n = static_cast<size_t>(key_copies) * rt->npix;
        HIP_TRY(hipMalloc(&rt->d_scratch_key, n * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(rt->d_scratch_key, 0, n * sizeof(unsigned long long), rt->stream));
        rt->key_copies = key_copies;
    }
    return SAR_OK;
}

// Bin geometry of the LDS-binned path: bins of 2^shift pixels, at most kMaxBins of them, and the pixel -> (bin, record)
// map (BinMap, sar_internal.hpp). Interleaved bins need a power-of-two bin count: used when that costs at most a third
// more bins (LDS staging is per bin) than consecutive-pixel bins — 2048^2, 1800x2000, 1920x1080, 2560^2, 3840x2160 and
// 4096^2 all qualify; `interleave` 1 = never, 2 = whenever the count fits kMaxBins.
struct BinGeometry {
    uint32_t shift = 0, bins = 0, block = 0, splits = 0;
    BinMap map{};
    bool interleaved = false;
    bool ok = false;
};
BinGeometry bin_geometry(uint32_t npix, uint32_t want_block, uint32_t want_shift, uint32_t want_splits, uint32_t records, bool pool,
                         uint32_t interleave) {
    BinGeometry g;
    uint32_t px = 4096;
    while (px < kMaxHistPx && static_cast<uint64_t>(px) * 256u < npix) px <<= 1;
    if (static_cast<uint64_t>(px) * kMaxBins < npix) px = kMaxBinPx;  // 32..64 Mpx: bins of 65536 pixels
    if (want_shift) px = 1u << want_shift;
    g.bins = (npix + px - 1) / px;
    if (g.bins > kMaxBins) return g;
    while ((1u << g.shift) < px) ++g.shift;
    uint32_t b = 0;
    while ((1u << b) < g.bins) ++b;
    const uint32_t pow2_bins = 1u << b;
    g.interleaved = interleave != 1u && pow2_bins <= kMaxBins && (interleave == 2u || 3ull * pow2_bins <= 4ull * g.bins);
    if (g.interleaved) {
        g.bins = pow2_bins;
        g.map.seg_shift = g.shift < 11u ? g.shift : 11u;  // 2048-pixel segments (== k_fold_resolve's blocks)
        g.map.bin_bits = b;
        g.map.hi_shift = b;
        g.map.low_mask = (1u << g.map.seg_shift) - 1u;
    } else {
        g.map.seg_shift = g.shift;
        g.map.bin_bits = 32u - g.shift;
        g.map.hi_shift = 31u;
        g.map.low_mask = px - 1u;
    }
    const uint32_t waves_fit = (160u * 1024u) / lean_wave_lds_bytes(g.bins, records, pool);
    uint32_t block = want_block;
    if (block > waves_fit * 64u) block = waves_fit * 64u;
    if (block == 0) return g;
    g.block = block;
    g.splits = 0;  // chosen per launch: about one (bin, wave) list per thread of a 1024-thread block
    if (want_splits) g.splits = want_splits;
    g.ok = true;
    return g;
}

int clear_hints(sar_runtime* rt) {
    // hints are lower bounds of depths already accumulated; anything that can lower zbuf voids them
    // Wide hints hold the depth itself as f32 and start at the smallest float above -1.0 (nextafter(-1, +inf) = 0xBF7FFFFF): stage 1's `z >= hint` is then the
    // reference's strict `z > -1.0` (:693, :821) for a pixel nobody has reached. Narrow hints are 16-bit fixed point from 0.
    if (rt->d_zhint && rt->zhint_bytes == 4)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(rt->d_zhint), static_cast<int>(0xBF7FFFFFu), (static_cast<size_t>(rt->npix) + 2u) * 8u, rt->stream));
    else if (rt->d_zhint)
        HIP_TRY(hipMemsetAsync(rt->d_zhint, 0, (static_cast<size_t>(rt->npix) + 2u) * 8u * rt->zhint_bytes, rt->stream));
    rt->hint_range_set = false;  // empty hints: the next launch may measure the view's depth range anew
    return SAR_OK;
}

int do_reset(sar_runtime* rt) {
    SAR_TRY(clear_hints(rt));
    launch_reset(rt->d_count, rt->d_key, rt->d_steps, rt->npix, rt->d_scalars, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

void span_begin(sar_runtime* rt, std::vector<Span>& spans, size_t& used) {
    if (!rt->timing) return;
    if (used == spans.size()) {
        Span s;
        hipEventCreate(&s.a);
        hipEventCreate(&s.b);
        spans.push_back(s);
    }
    hipEventRecord(spans[used].a, rt->stream);
}
void span_end(sar_runtime* rt, std::vector<Span>& spans, size_t& used) {
    if (!rt->timing) return;
    hipEventRecord(spans[used].b, rt->stream);
    ++used;
}
void single_begin(sar_runtime* rt, Span& s) {
    if (!rt->timing) return;
    if (!s.a) { hipEventCreate(&s.a); hipEventCreate(&s.b); }
    hipEventRecord(s.a, rt->stream);
}
void single_end(sar_runtime* rt, Span& s, bool& flag) {
    if (!rt->timing) return;
    hipEventRecord(s.b, rt->stream);
    flag = true;
}

void fill_map_params(const sar_config& cfg, MapParams& p) {
    for (int k = 0; k < 10; ++k) {
        p.cx[k] = cfg.coeff_x[k];
        p.cy[k] = cfg.coeff_y[k];
        p.cz[k] = cfg.coeff_z[k];
    }
    // the reference's sum starts as `0. + 1.*c0` (src/lib.rs:589-597): identical to c0 except that a
    // -0.0 coefficient becomes +0.0
    p.cx[0] = 0. + 1. * cfg.coeff_x[0];
    p.cy[0] = 0. + 1. * cfg.coeff_y[0];
    p.cz[0] = 0. + 1. * cfg.coeff_z[0];
    rotation_matrix(cfg, p.m);                 // :755
    p.sin_v = std::sin(cfg.angle);             // :756
    p.cos_v = std::cos(cfg.angle);             // :757
    p.ccx = cfg.center_camera[0];
    p.ccy = cfg.center_camera[1];
    p.ccz = cfg.center_camera[2];
    p.width = static_cast<double>(cfg.width);   // :760
    p.height = static_cast<double>(cfg.height); // :762
    p.half_height = p.height / 2.;              // `height / 2.` of :786
    p.width_scaled = p.width * cfg.scale;       // :763
    p.scale_adjusted_mid = 0.5 / cfg.scale;     // :764
}

void fill_ct_params(const sar_config& cfg, ColorTransformParams& ct) {
    ct.kind = cfg.color_transform;
    ct._pad = 0;
    ct.offset = cfg.ct_offset;
    ct.factor = cfg.ct_factor;
    ct.ccx = cfg.center_camera[0];
    ct.ccy = cfg.center_camera[1];
}

}  // namespace

int sar::check_cfg_matches(const sar_config* cfg, const sar_runtime* rt) {
    SAR_TRY(validate(cfg));
    if (!rt) { set_error("runtime is NULL"); return SAR_ERR_INVALID; }
    if (cfg->width != rt->W || cfg->height != rt->H) {
        set_error("config is %ux%u but the runtime holds %ux%u", cfg->width, cfg->height, rt->W, rt->H);
        return SAR_ERR_DIM_MISMATCH;
    }
    return SAR_OK;
}

namespace {

// Grows a device buffer (contents are not preserved). cap and need in elements of T.
template <typename T>
int grow_device(T*& ptr, size_t& cap, size_t need) {
    if (need <= cap) return SAR_OK;
    if (ptr) hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ptr), need * sizeof(T)));
    cap = need;
    return SAR_OK;
}

// Everything one render call decides before it launches: which accumulate path, its geometry, and how the job list is
// cut into launch chunks (chunk boundaries fall on whole jobs; a chunk keeps job*iters + t inside 32 bits and its
// scratch inside kCkptBytesCap).
struct LaunchPlan {
    bool binned = false, xcd_local = false, pool = false;
    bool split = false;                 // the iterate kernel as producer / consumer wave pairs (k_iterate_split)
    uint64_t resident_jobs = 0;         // trajectories resident at once under this plan: launch chunks are whole rounds of them
    BinGeometry geo;
    uint32_t R = kDefaultChunkRecords;  // records per chunk
    uint32_t block = 0;                 // trajectories per workgroup of the iterate kernel
    uint32_t pipe = kDefaultDepthPipe;  // depth pipeline length
    uint32_t hint_bytes = 4;            // per-XCD depth hints: 2 (fixed point) or 4 (sortable f32)
    uint32_t C = 0;                     // checkpoint stride
    uint32_t splits = 0;                // accumulate workgroups per bin
    uint64_t n_ckpt = 0, chunks_per_wave = 0, chunk_jobs = 0;
    uint32_t max_waves = 0;             // waves of the largest launch chunk
    uint32_t arena_waves = 0;           // ... of which hold at least one job: only they own a slice of the record arena
};

// Records per chunk: the largest of 28/20/12 whose per-wave LDS staging still fits the waves this launch can use — up to
// 3 per SIMD (more jobs than that run in rounds), at least 2. Measured at 2048^2, 1e9 iterations: 131072 jobs with 28
// records and 196608 jobs with 20 end within 2 % of each other; at 4096^2 12 records (2 waves/SIMD) beat 28 (1 wave/
// SIMD) by 1.4x. The job count is scaled by the share of jobs that survived the previous launch's warm-up.
// Also picks the stager: the pool stager (sar_iterate.hip: full buffers swapped against spares, cooperative copy-out;
// 3-4 % faster where it fits) needs a little more LDS per wave — it is used when it keeps the waves per CU the classic
// stager reaches with the same chunk size.
uint32_t choose_chunk_records(sar_runtime* rt, uint32_t n_jobs, bool& pool, uint32_t& shift, uint32_t& interleave, bool& split,
                              uint64_t& resident_jobs) {
    shift = rt->bin_shift;
    interleave = rt->bin_interleave;
    if (rt->active_pending && hipEventQuery(rt->active_copied) == hipSuccess) {
        rt->active_pending = false;
        if (rt->active_jobs_launched) rt->survivor_fraction = static_cast<double>(*rt->h_active) / rt->active_jobs_launched;
    }
    const uint64_t cus = rt->sm_count ? rt->sm_count : 256u;
    const uint64_t busy = static_cast<uint64_t>(n_jobs * rt->survivor_fraction + 0.5);
    uint64_t want = (busy + 64u * cus - 1) / (64u * cus);  // waves per CU if all surviving jobs were resident
    want = ((want + 3) / 4) * 4;  // workgroups are four waves: residency comes in steps of four waves per CU
    want = want < 8 ? 8 : (want > 12 ? 12 : want);
    // k_iterate_split (producer / consumer wave pairs) keeps 8 staging sets per CU busy with 16 waves: for launches whose
    // jobs are all resident at once (512 per CU). A launch of several rounds of workgroups desynchronises by itself —
    // workgroups of different rounds are in different phases — and the whole kernel is the faster one there (configs[3] on
    // one GPU, 8 rounds: 81.7 against 89.9 ms).
    {
        const uint64_t cap = 512u * cus;
        const bool fills = busy <= cap;
        split = rt->split_waves == 2 || (rt->split_waves == 0 && fills);
        if (rt->measure_mode || rt->stager == 1 || (rt->depth_pipe && rt->depth_pipe != 2)) split = false;
        if (split) want = 8;
    }
    // jobs (dead ones included: they are launched and dropped by the warm-up) whose survivors the chip holds at once
    resident_jobs = static_cast<uint64_t>(64.0 * cus * want / (rt->survivor_fraction > 0.05 ? rt->survivor_fraction : 0.05));
    // Interleaved bins carry equal loads, so few LARGE bins cost the slot requests nothing (with bins of consecutive
    // pixels half the bins idle and the rest collide) and k_bin_accumulate's 128 KiB histograms (one workgroup per CU) get
    // equal work. 128 bins of 32768 pixels leave the pool stager room for 128-byte chunks at two waves per SIMD — half
    // the buffer swaps, whole cache lines for k_bin_accumulate — or for 64-byte chunks at three. (2048^2, 1e9 iterations:
    // 131072 jobs 7.0 -> 6.x ms per frame)
    // Beyond 4 Mpx the same with bins of 65536 pixels (all a 16-bit record addresses; k_bin_accumulate counts such a bin in
    // two halves, reading its lists twice): 4096^2 in 256 bins keeps 64-byte chunks where 512 bins allowed 32-byte ones
    // (1.25e9 iterations there: 13.5 -> 12.3 ms; 2560^2 and 3840x2160 take 128-byte chunks on 128 such bins: -2..3 %).
    if (rt->bin_shift == 0 && rt->chunk_records == 0 && rt->stager != 1 && rt->bin_interleave != 1) {
        for (uint64_t need : {want, static_cast<uint64_t>(8)})  // three waves per SIMD if the launch has the jobs, else two
            for (uint32_t cand : {60u, 28u})                      // the larger chunk first, the smaller bin first
                for (uint32_t sh : {15u, 16u}) {
                    // interleaved whatever the power-of-two bin count costs: the staging is checked to fit right here
                    const BinGeometry big = bin_geometry(rt->npix, rt->block_threads, sh, rt->splits, 12u, true, 2u);
                    if (big.ok && big.interleaved && lean_wave_lds_bytes(big.bins, cand, true) * need <= 160u * 1024u) {
                        pool = true;
                        shift = sh;
                        interleave = 2u;
                        return cand;
                    }
                }
    }
    const BinGeometry probe = bin_geometry(rt->npix, rt->block_threads, rt->bin_shift, rt->splits, 12u, false, rt->bin_interleave);
    uint32_t R = rt->chunk_records, need_waves = 8;
    if (R == 0 && !probe.ok) R = kDefaultChunkRecords;
    if (R == 0) {
        R = 12u;
        bool found = false;
        for (uint32_t need : {static_cast<uint32_t>(want), 8u}) {
            for (uint32_t cand : {28u, 20u, 12u})
                if (lean_wave_lds_bytes(probe.bins, cand, false) * need <= 160u * 1024u) { R = cand; need_waves = need; found = true; break; }
            if (found) break;
        }
    } else if (probe.ok) {
        need_waves = (lean_wave_lds_bytes(probe.bins, R == 60u ? 28u : R, false) * want <= 160u * 1024u) ? static_cast<uint32_t>(want) : 8u;
    }
    pool = rt->stager == 2 || R == 60u ||
           (rt->stager == 0 && probe.ok && lean_wave_lds_bytes(probe.bins, R, true) * need_waves <= 160u * 1024u &&
            lean_wave_lds_bytes(probe.bins, R, false) * need_waves <= 160u * 1024u);
    return R;
}

int plan_launch(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters, LaunchPlan& pl) {
    uint32_t shift = 0, interleave = 0;
    pl.R = choose_chunk_records(rt, n_jobs, pl.pool, shift, interleave, pl.split, pl.resident_jobs);
    pl.geo = bin_geometry(rt->npix, rt->block_threads, shift, rt->splits, pl.R, pl.pool, interleave);
    // which accumulate path: LDS-binned records (default) or one global atomic per visit
    pl.binned = (rt->bins_mode == 0 || rt->bins_mode == 3) && rt->measure_mode != 2 && pl.geo.ok;
    if (rt->bins_mode == 3 && !pl.geo.ok) {
        set_error("the binned path needs width*height <= %u pixels", kMaxBins * kMaxBinPx);
        return SAR_ERR_RANGE;
    }
    pl.xcd_local = (rt->bins_mode == 2);
    pl.block = pl.binned ? pl.geo.block : rt->block_threads;
    pl.pipe = rt->depth_pipe ? rt->depth_pipe : kDefaultDepthPipe;
    // depth hints: the sortable f32 itself (3x fewer stage-2 waits, -7 % at 2048^2) while the hints of the pixels the
    // attractor touches stay near an XCD's 4 MiB L2, 16-bit fixed point beyond. The view maps the attractor onto
    // (width * scale)^2 pixels whatever the height, so that is the measure: 32-bit wins at 2048^2 / 2560^2 / 3072^2,
    // 16-bit at 3840x2160 (-8 %) and 4096^2 (-13 %).
    const double span = static_cast<double>(cfg->width) * cfg->scale;
    pl.hint_bytes = rt->hint_bits ? rt->hint_bits / 8u : ((span * span <= kWideHintMaxSpan2 && rt->npix <= (16u << 20)) ? 4u : 2u);
    // checkpoint stride: a multiple of the depth pipeline's pass length (the iterate kernel runs whole passes)
    pl.C = ((rt->ckpt_stride + pl.pipe - 1u) / pl.pipe) * pl.pipe;
    pl.n_ckpt = (iters + pl.C - 1) / pl.C;
    // Record arena: a wave emits at most one record per lane and iteration, in chunks of R, plus one partly filled chunk
    // per bin at the end. Sized for the lanes that really hold a job — a single-trajectory sar_render (n_jobs = 1) is one
    // lane of one wave, not a full 256-thread block of busy lanes.
    auto lanes_of = [](uint64_t jobs) { return jobs < 64 ? jobs : 64ull; };
    auto chunks_per_wave_of = [&](uint64_t jobs) { return (iters * lanes_of(jobs) + pl.R - 1) / pl.R + pl.geo.bins; };
    pl.chunk_jobs = (rt->max_ordinals ? rt->max_ordinals : kMaxChunkOrdinals) / iters;  // >= 1: iters is one segment
    if (pl.chunk_jobs > n_jobs) pl.chunk_jobs = n_jobs;
    if (pl.binned && chunks_per_wave_of(pl.chunk_jobs) > 0xFFFFFFF0ull) pl.binned = false;
    // scratch per job: checkpoints (24 B each) + its share of its wave's arena (binned path)
    const uint64_t cb = chunk_bytes(pl.R);
    auto scratch_bytes = [&](uint64_t jobs) {
        const uint64_t waves = (jobs + 63) / 64;
        return jobs * pl.n_ckpt * 24ull + (pl.binned ? waves * chunks_per_wave_of(jobs) * cb : 0ull);
    };
    while (pl.chunk_jobs > 1 && scratch_bytes(pl.chunk_jobs) > kCkptBytesCap) {
        // linear in the job count above one wave: one division gets close, the loop finishes the rounding
        const uint64_t per_job = scratch_bytes(pl.chunk_jobs) / pl.chunk_jobs + 1;
        uint64_t fit = kCkptBytesCap / per_job;
        if (fit >= pl.chunk_jobs) fit = pl.chunk_jobs - 1;
        pl.chunk_jobs = fit ? fit : 1;
    }
    if (pl.binned && scratch_bytes(1) > kCkptBytesCap) pl.binned = false;  // one job alone overflows the arena cap: atomics path
    if (rt->debug_chunk_jobs && rt->debug_chunk_jobs < pl.chunk_jobs) pl.chunk_jobs = rt->debug_chunk_jobs;
    if (pl.chunk_jobs > pl.block) pl.chunk_jobs -= pl.chunk_jobs % pl.block;
    // jobs that need several launches anyway (the 2^32 visit ordinals, the scratch cap): launches of whole rounds of resident
    // workgroups, so that no launch ends on a nearly empty round. (Jobs that fit ONE launch stay one launch: its rounds overlap.)
    if (pl.binned && pl.resident_jobs && n_jobs > pl.chunk_jobs && pl.chunk_jobs > pl.resident_jobs && !rt->debug_chunk_jobs)
        pl.chunk_jobs -= pl.chunk_jobs % pl.resident_jobs;
    pl.chunks_per_wave = chunks_per_wave_of(pl.chunk_jobs);
    pl.max_waves = static_cast<uint32_t>(((pl.chunk_jobs + pl.block - 1) / pl.block) * (pl.block / 64u));
    pl.arena_waves = static_cast<uint32_t>((pl.chunk_jobs + 63) / 64);  // waves that hold a job (the others exit at once)
    pl.splits = pl.geo.splits;
    if (pl.binned && pl.splits == 0) {
        // k_bin_accumulate walks one (bin, wave) list per group of lanes (4, or 2 with 32-byte chunks): aim at one
        // list per group, and at enough blocks to cover the chip when only a band of bins is populated
        const uint32_t threads = rt->acc_threads ? rt->acc_threads : 1024u;
        const uint32_t groups = threads / (pl.R == 12u ? 2u : (pl.R == 60u ? 8u : 4u));
        pl.splits = (pl.max_waves + groups - 1u) / groups;
        uint32_t cover = 2048u / pl.geo.bins;
        if (pl.geo.shift >= 15u && pl.geo.interleaved) {
            // 128 KiB histograms: one workgroup per CU is resident, and with interleaved bins all of them carry the same
            // load — two rounds of workgroups over the chip, up to a few lists per lane group (measured, 2048^2: 4
            // workgroups per bin 0.85 ms, 8 or 16 1.2 ms)
            cover = 512u / ((pl.geo.shift == 16u && rt->acc_halves) ? 2u * pl.geo.bins : pl.geo.bins);  // counted in halves: two workgroups per bin and split
            pl.splits = (pl.max_waves + 8u * groups - 1u) / (8u * groups);
        }
        if (pl.splits < cover) pl.splits = cover;
        if (pl.splits < 1) pl.splits = 1;
        if (pl.splits > 16) pl.splits = 16;
    }
    return SAR_OK;
}

// Start points into rt->d_starts, laid out as consecutive per-chunk SoA blocks x[m] y[m] z[m]; `starts` is the caller's
// [n_jobs][3] array in host memory (through one pinned staging buffer) or already in device memory.
int stage_starts(sar_runtime* rt, const LaunchPlan& pl, uint32_t n_jobs, const double* starts, bool on_device) {
    const size_t need = static_cast<size_t>(n_jobs) * 3;
    if (rt->starts_pending) {  // the previous call's upload still reads the staging buffer
        HIP_TRY(hipEventSynchronize(rt->starts_copied));
        rt->starts_pending = false;
    }
    if (need > rt->starts_cap) {
        if (rt->h_starts) hipHostFree(rt->h_starts);
        if (rt->d_starts) hipFree(rt->d_starts);
        rt->h_starts = nullptr;
        rt->d_starts = nullptr;
        rt->starts_cap = 0;
        HIP_TRY(hipHostMalloc(&rt->h_starts, need * sizeof(double), hipHostMallocDefault));
        HIP_TRY(hipMalloc(&rt->d_starts, need * sizeof(double)));
        rt->starts_cap = need;
    }
    if (on_device) {
        for (uint64_t off = 0; off < n_jobs; off += pl.chunk_jobs) {
            const uint32_t m = static_cast<uint32_t>((n_jobs - off < pl.chunk_jobs) ? n_jobs - off : pl.chunk_jobs);
            launch_starts_soa(starts + off * 3, rt->d_starts + off * 3, m, rt->stream);
        }
        HIP_TRY(hipGetLastError());
        return SAR_OK;
    }
    for (uint64_t off = 0; off < n_jobs; off += pl.chunk_jobs) {
        const uint64_t m = (n_jobs - off < pl.chunk_jobs) ? n_jobs - off : pl.chunk_jobs;
        double* blk = rt->h_starts + off * 3;
        for (uint64_t k = 0; k < m; ++k) {
            blk[k] = starts[(off + k) * 3 + 0];
            blk[m + k] = starts[(off + k) * 3 + 1];
            blk[2 * m + k] = starts[(off + k) * 3 + 2];
        }
    }
    HIP_TRY(hipMemcpyAsync(rt->d_starts, rt->h_starts, need * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    HIP_TRY(hipEventRecord(rt->starts_copied, rt->stream));
    rt->starts_pending = true;
    return SAR_OK;
}

// Device buffers of the binned path: record arena, list heads, depth hints, warm-up output, counters.
int ensure_binned_buffers(sar_runtime* rt, const LaunchPlan& pl) {
    {   // hipFuncSetAttribute is per device and function: once for every device a runtime lives on
        static std::mutex attr_mu;
        static bool attr_done[64] = {false};
        std::lock_guard<std::mutex> lock(attr_mu);
        const int dev = rt->device;
        if (dev < 0 || dev >= 64 || !attr_done[dev]) {
            const int attr_status = binned_kernel_attributes();  // on the current device (render_chunked set it)
            if (attr_status != 0) { set_error("hipFuncSetAttribute(max dynamic LDS) failed: %d", attr_status); return SAR_ERR_HIP; }
            if (dev >= 0 && dev < 64) attr_done[dev] = true;
        }
    }
    {
        char* arena = static_cast<char*>(rt->d_arena);
        const int rc = grow_device(arena, rt->arena_cap, static_cast<size_t>(pl.arena_waves) * pl.chunks_per_wave * chunk_bytes(pl.R));
        rt->d_arena = arena;  // also when the allocation failed: the old buffer is gone
        SAR_TRY(rc);
    }
    SAR_TRY(grow_device(rt->d_heads, rt->heads_cap, static_cast<size_t>(pl.max_waves) * pl.geo.bins));
    if (!rt->d_zhint || rt->zhint_bytes != pl.hint_bytes) {
        if (rt->d_zhint) hipFree(rt->d_zhint);
        rt->d_zhint = nullptr;
        HIP_TRY(hipMalloc(&rt->d_zhint, (static_cast<size_t>(rt->npix) + 2u) * 8u * pl.hint_bytes));
        rt->zhint_bytes = pl.hint_bytes;
        SAR_TRY(clear_hints(rt));
    }
    if (pl.chunk_jobs > rt->warm_cap) {
        size_t cap3 = 0, cap1 = 0;  // both buffers are replaced together
        rt->warm_cap = 0;
        SAR_TRY(grow_device(rt->d_warm, cap3, static_cast<size_t>(pl.chunk_jobs) * 3));
        SAR_TRY(grow_device(rt->d_joblist, cap1, static_cast<size_t>(pl.chunk_jobs)));
        rt->warm_cap = pl.chunk_jobs;
    }
    if (!rt->d_active) HIP_TRY(hipMalloc(&rt->d_active, 4 * sizeof(uint32_t)));
    const size_t segs = static_cast<size_t>(rt->npix) / 2048u + 1u;
    if (rt->seg_any_cap < segs) {
        if (rt->d_seg_any) hipFree(rt->d_seg_any);
        rt->d_seg_any = nullptr;
        rt->seg_any_cap = 0;
        HIP_TRY(hipMalloc(&rt->d_seg_any, segs * sizeof(uint32_t)));
        rt->seg_any_cap = segs;
    }
    if (!rt->h_active) {
        HIP_TRY(hipHostMalloc(&rt->h_active, sizeof(uint32_t), hipHostMallocDefault));
        *rt->h_active = 0;
        HIP_TRY(hipEventCreateWithFlags(&rt->active_copied, hipEventDisableTiming));
    }
    if (!rt->d_hint_range) {
        HIP_TRY(hipMalloc(&rt->d_hint_range, 2 * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(rt->d_hint_range, 0, 2 * sizeof(uint32_t), rt->stream));
    }
    if (!rt->d_nan_count) {
        // [0] NaN iterations, [1] depth atomics (stat), [2..5] segment cycles of the SAR_EXPERIMENT_PROF build
        // [6..7] producer wave, [8..12] consumer wave of k_iterate_split in that build
        HIP_TRY(hipMalloc(&rt->d_nan_count, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(rt->d_nan_count, 0, 16 * sizeof(unsigned long long), rt->stream));
    }
    return SAR_OK;
}

// One launch chunk of the binned path: warm-up + packing, iterate, accumulate, fold.
// `first`: the first segment of these jobs (warm-up + packing); `carry`: more segments follow (keep the trajectory state).
int launch_binned_chunk(sar_runtime* rt, const LaunchPlan& pl, const IterArgs& ia, const FoldArgs& fa_in, int mode, bool first, bool carry,
                        bool use_prefetch) {
    FoldArgs fa = fa_in;
    fa.seg_any = rt->d_seg_any;
    const uint32_t m = ia.n_jobs;
    BinIterArgs ba;
    std::memset(&ba, 0, sizeof(ba));
    ba.it = ia;
    ba.map = pl.geo.map;
    ba.n_bins = pl.geo.bins;
    ba.chunks_per_wave = static_cast<uint32_t>(pl.chunks_per_wave);
    ba.n_waves = ((m + pl.block - 1) / pl.block) * (pl.block / 64u);
    ba.arena = rt->d_arena;
    ba.heads = rt->d_heads;
    ba.zhint = rt->d_zhint;
    ba.nan_count = rt->d_nan_count;
    ba.hint_range = pl.hint_bytes == 2 ? rt->d_hint_range : nullptr;
    // One hint array per XCD lets every XCD's L2 serve its own hints coherently; but eight copies of a 4096^2 image's hints
    // (268 MB at 16 bits) no longer fit the 256 MB Infinity Cache behind the L2s, and the misses go to HBM. From 200 MB on
    // the XCDs share ONE array: an XCD then sees another's updates only when its own L2 drops the line — a stale hint lets
    // more visits through stage 1, never a wrong one — and the misses stay on chip (4096^2 share: 9.15 -> 8.85 ms; below
    // that size sharing costs: 2048^2 5.90 -> 6.05 ms).
    const bool share = rt->hint_shared == 2 || (rt->hint_shared == 0 && static_cast<uint64_t>(rt->npix) * pl.hint_bytes * 8u > (200ull << 20));
    ba.hint_copy_mask = share ? 0u : 7u;
    // narrow hints of an image whose width is a power of two: 8 x 8 tiles per 128-byte line (HintTile); the permutation stays
    // inside blocks of eight rows, so the height must be a multiple of eight
    const bool pow2w = (rt->W & (rt->W - 1u)) == 0u && rt->W >= 8u && rt->H % 8u == 0u;
    if (pl.hint_bytes == 2 && pow2w && rt->hint_tile != 1u) {
        uint32_t b = 0;
        while ((1u << b) < rt->W) ++b;
        ba.tile.shift1 = b - 3u;
        ba.tile.mask1 = 0x38u;
        ba.tile.mask2 = ((1u << (b + 3u)) - 1u) & ~7u;
    }
    ba.warm_out = carry ? rt->d_warm : nullptr;
    span_begin(rt, rt->warm_spans, rt->warm_used);
    const sar_runtime::Prefetch& pf = rt->pf;
    // The warm-up is the MAP alone (:750-752): an announcement stands for every call with the same 30 coefficients, start
    // points and job shape — a sweep's next frame has another angle, the same warm-up. (The depth range a warm-up measured
    // for the narrow hints under the announcing view only sets their quantiser: any range gives the same image.)
    const bool same_map = std::memcmp(pf.p.cx, ia.p.cx, sizeof(ia.p.cx)) == 0 && std::memcmp(pf.p.cy, ia.p.cy, sizeof(ia.p.cy)) == 0 &&
                          std::memcmp(pf.p.cz, ia.p.cz, sizeof(ia.p.cz)) == 0;
    const bool ahead = first && !carry && use_prefetch && pf.valid && pf.m == m && pf.iters == ia.iters && pf.width == ia.width && same_map;
    if (ahead) {
        // this chunk's warm-up ran ahead (sar_runtime_prefetch_device): its buffers become the current ones
        HIP_TRY(hipStreamWaitEvent(rt->stream, rt->pf_done, 0));
        std::swap(rt->d_warm, rt->d_warm_alt);
        std::swap(rt->d_joblist, rt->d_joblist_alt);
        std::swap(rt->d_active, rt->d_active_alt);
        std::swap(rt->warm_cap, rt->warm_alt_cap);
        if (pl.hint_bytes == 2 && !rt->hint_range_set) {
            // the quantiser of the narrow hints is fixed here for as long as the hints live: the range the announced warm-up
            // measured — or, if it did not measure one (the options changed in between), the default quantiser (an empty range)
            if (pf.range_measured)
                HIP_TRY(hipMemcpyAsync(rt->d_hint_range, rt->d_hint_range_alt, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, rt->stream));
            else
                HIP_TRY(hipMemsetAsync(rt->d_hint_range, 0, 2 * sizeof(uint32_t), rt->stream));
            rt->hint_range_set = true;
        }
        ++rt->prefetch_used;
    } else if (first) {
        HIP_TRY(hipMemsetAsync(rt->d_active, 0, 4 * sizeof(uint32_t), rt->stream));
        // narrow hints: the first warm-up after the hints were cleared also measures the depth range they quantise
        uint32_t* measure = nullptr;
        if (pl.hint_bytes == 2 && !rt->hint_range_set) {
            HIP_TRY(hipMemsetAsync(rt->d_hint_range, 0, 2 * sizeof(uint32_t), rt->stream));
            measure = rt->d_hint_range;
            rt->hint_range_set = true;
        }
        launch_warmup(ia.p, ia.starts, m, ia.iters, rt->d_warm, rt->d_joblist, rt->d_active,
                      reinterpret_cast<unsigned long long*>(rt->d_active + 2), ia.width, measure, rt->stream);
    } else {
        launch_dead_jobs(rt->d_active, m, ia.iters, rt->d_nan_count, rt->stream);
    }
    if (first) rt->pf.valid = false;  // used, or announced for another call: either way it is spent
    ba.warm = rt->d_warm;
    ba.joblist = rt->d_joblist;
    ba.active = rt->d_active;
    ba.warm_nan = first ? reinterpret_cast<const unsigned long long*>(rt->d_active + 2) : nullptr;
    if (first && !rt->active_pending) {  // statistics for the next call; nobody waits for this copy
        if (hipMemcpyAsync(rt->h_active, rt->d_active, sizeof(uint32_t), hipMemcpyDeviceToHost, rt->stream) == hipSuccess &&
            hipEventRecord(rt->active_copied, rt->stream) == hipSuccess) {
            rt->active_pending = true;
            rt->active_jobs_launched = m;
        }
    }
    span_end(rt, rt->warm_spans, rt->warm_used);
    span_begin(rt, rt->iter_spans, rt->iter_used);
    const bool split = pl.split && pl.pool && (pl.R == 60u || pl.R == 28u) && mode == 2 && pl.pipe == 2 &&
                       (lean_wave_lds_bytes(pl.geo.bins, pl.R, true) + 1024u) * 8u <= 160u * 1024u;
    if (launch_iterate_lean(ba, pl.block, pl.R, pl.pipe, pl.hint_bytes, mode == 2, pl.pool, split, rt->stream) != 0) {
        set_error("bad chunk_records / depth_pipe");
        return SAR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    span_end(rt, rt->iter_spans, rt->iter_used);
    ++rt->last_chunks;
    std::snprintf(rt->last_launch, sizeof(rt->last_launch),
                  "%s R=%u bins=%ux%upx %s hints=%s pipe=%u | k_bin_accumulate splits=%u lists=%u counters=%s",
                  split ? "k_iterate_split" : "k_iterate_lean", pl.R, pl.geo.bins, 1u << pl.geo.shift, pl.geo.interleaved ? "interleaved" : "consecutive",
                  mode != 2 ? "none" : (pl.hint_bytes == 4 ? (share ? "f32/chip" : "f32") : (share ? "q16/chip" : "q16")), pl.pipe, pl.splits,
                  rt->acc_lists ? rt->acc_lists : (pl.geo.shift >= 15u ? 4u : 1u),
                  pl.geo.shift == 16u ? (rt->acc_halves ? "u32-halves" : "u16-packed") : "u32");
    if (rt->iter_done) {  // an announced call's warm-up starts here, under this launch's accumulate and fold
        HIP_TRY(hipEventRecord(rt->iter_done, rt->stream));
        rt->iter_done_recorded = true;
    }
    BinAccArgs ca;
    std::memset(&ca, 0, sizeof(ca));
    ca.bin_shift = pl.geo.shift;
    ca.n_bins = pl.geo.bins;
    ca.chunks_per_wave = ba.chunks_per_wave;
    ca.n_waves = ba.n_waves;
    ca.npix = rt->npix;
    ca.splits = pl.splits;
    ca.arena = rt->d_arena;
    ca.heads = rt->d_heads;
    ca.scratch_count = rt->d_scratch_count;
    ca.map = pl.geo.map;
    ca.seg_any = rt->d_seg_any;
    span_begin(rt, rt->fold_spans, rt->fold_used);
    HIP_TRY(hipMemsetAsync(rt->d_seg_any, 0, (static_cast<size_t>(rt->npix) / 2048u + 1u) * sizeof(uint32_t), rt->stream));
    // lists a lane group walks at the same time: with the 128 KiB histogram one workgroup per CU is resident — four loads
    // in flight per lane make up for the missing second workgroup (2048^2: 0.61 -> 0.46 ms); with two workgroups per CU
    // (64 KiB) more loads in flight change nothing
    launch_bin_accumulate(ca, rt->acc_threads, pl.R, rt->acc_lists ? rt->acc_lists : (pl.geo.shift >= 15u ? 4u : 1u), rt->acc_halves != 0, rt->stream);
    HIP_TRY(hipGetLastError());
    launch_fold_resolve(fa, rt->stream);
    span_end(rt, rt->fold_spans, rt->fold_used);
    return SAR_OK;
}

// Runs n_jobs trajectories of `iters` counted iterations each; starts is AoS [n_jobs][3] on the host (or, with
// starts_on_device, in device memory). Sequential semantics (job-major, iteration-minor): a later launch chunk only
// replaces a depth winner with a strictly greater z, exactly like a later render call.
}  // namespace

// The warm-up of the first `m` jobs of a coming launch, on the side stream, into the second set of warm-up buffers: behind
// the iterate kernel in flight (its accumulate / fold / colorize are what this runs under) or, with nothing in flight, at
// once. `starts` is [m][3] in device memory, or (soa) the kernel's x[m] y[m] z[m] block. Leaves rt->pf describing it.
static int warmup_ahead(sar_runtime* rt, const sar::MapParams& p, const double* starts, bool soa, uint32_t m, uint64_t iters,
                        bool measure_range) {
    rt->pf.valid = false;
    if (!rt->side) {
        HIP_TRY(hipStreamCreateWithFlags(&rt->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&rt->iter_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&rt->pf_done, hipEventDisableTiming));
    }
    if (m > rt->warm_alt_cap) {
        // what wrote the second set last ran on this side stream; what read it last — it was the current set before the last
        // swap — may be an iterate kernel still in flight on the launch stream (a rare path: only while the sets grow)
        HIP_TRY(hipStreamSynchronize(rt->side));
        HIP_TRY(hipStreamSynchronize(rt->stream));
        for (void* q : {static_cast<void*>(rt->d_warm_alt), static_cast<void*>(rt->d_joblist_alt)})
            if (q) hipFree(q);
        rt->d_warm_alt = nullptr; rt->d_joblist_alt = nullptr;
        rt->warm_alt_cap = 0;
        HIP_TRY(hipMalloc(&rt->d_warm_alt, static_cast<size_t>(m) * 3 * sizeof(double)));
        HIP_TRY(hipMalloc(&rt->d_joblist_alt, static_cast<size_t>(m) * sizeof(uint32_t)));
        rt->warm_alt_cap = m;
    }
    // the converted start points of an announced call: NOT one of the two sets that swap (its capacity is its own)
    if (!soa && m > rt->starts_alt_cap) {
        HIP_TRY(hipStreamSynchronize(rt->side));
        if (rt->d_starts_alt) hipFree(rt->d_starts_alt);
        rt->d_starts_alt = nullptr;
        rt->starts_alt_cap = 0;
        HIP_TRY(hipMalloc(&rt->d_starts_alt, static_cast<size_t>(m) * 3 * sizeof(double)));
        rt->starts_alt_cap = m;
    }
    if (!rt->d_active_alt) HIP_TRY(hipMalloc(&rt->d_active_alt, 4 * sizeof(uint32_t)));
    if (!rt->d_hint_range_alt) HIP_TRY(hipMalloc(&rt->d_hint_range_alt, 2 * sizeof(uint32_t)));
    sar_runtime::Prefetch& pf = rt->pf;
    pf.p = p;
    pf.n_jobs = m;
    pf.m = m;
    pf.width = rt->W;
    pf.iters = iters;
    pf.starts = nullptr;
    pf.range_measured = measure_range;
    if (rt->iter_done_recorded) HIP_TRY(hipStreamWaitEvent(rt->side, rt->iter_done, 0));
    if (rt->prefetch_after) HIP_TRY(hipStreamWaitEvent(rt->side, rt->prefetch_after, 0));  // the points are still on their way
    if (!soa) launch_starts_soa(starts, rt->d_starts_alt, m, rt->side);
    HIP_TRY(hipMemsetAsync(rt->d_active_alt, 0, 4 * sizeof(uint32_t), rt->side));
    if (measure_range) HIP_TRY(hipMemsetAsync(rt->d_hint_range_alt, 0, 2 * sizeof(uint32_t), rt->side));
    launch_warmup(pf.p, soa ? starts : rt->d_starts_alt, m, iters, rt->d_warm_alt, rt->d_joblist_alt, rt->d_active_alt,
                  reinterpret_cast<unsigned long long*>(rt->d_active_alt + 2), rt->W, measure_range ? rt->d_hint_range_alt : nullptr, rt->side);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(rt->pf_done, rt->side));
    pf.valid = true;
    return SAR_OK;
}

int sar::render_chunked(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters, const double* starts,
                        bool starts_on_device) {
    if (!rt->timing_accumulate) {
        rt->last_iterations = 0;
        rt->iter_used = 0;
        rt->fold_used = 0;
        rt->warm_used = 0;
    }
    // an announcement is good for the very next render call only, and only if that call is the announced one
    if (!(starts_on_device && rt->pf.valid && rt->pf.starts == starts && rt->pf.n_jobs == n_jobs && rt->pf.iters == iters)) rt->pf.valid = false;
    if (n_jobs == 0 || iters == 0) return SAR_OK;
    HIP_TRY(hipSetDevice(rt->device));
    rt->last_chunks = 0;
    rt->last_launch[0] = 0;

    // A launch orders its visits with a 32-bit ordinal (job * n + t). Config::iterations is a usize (:267): a job with more
    // iterations than that runs as SEGMENTS — successive launches that hand the trajectory state on (no second warm-up),
    // each folded before the next, so that an earlier segment wins depth ties exactly like an earlier iteration.
    const uint64_t max_ord = rt->max_ordinals ? rt->max_ordinals : kMaxChunkOrdinals;
    const uint64_t seg = iters <= max_ord ? iters : max_ord;
    const uint64_t n_seg = (iters + seg - 1) / seg;

    LaunchPlan pl;
    SAR_TRY(plan_launch(cfg, rt, n_jobs, seg, pl));
    SAR_TRY(ensure_scratch(rt, pl.binned ? pl.splits : (pl.xcd_local ? 8u : 1u), (!pl.binned && pl.xcd_local) ? 8u : 1u));
    SAR_TRY(stage_starts(rt, pl, n_jobs, starts, starts_on_device));
    SAR_TRY(grow_device(rt->d_ckpt, rt->ckpt_cap, static_cast<size_t>(pl.n_ckpt) * 3 * pl.chunk_jobs));
    if (pl.binned) SAR_TRY(ensure_binned_buffers(rt, pl));

    IterArgs ia;
    std::memset(&ia, 0, sizeof(ia));
    fill_map_params(*cfg, ia.p);
    ia.width = rt->W;
    ia.npix = rt->npix;
    ia.ckpt_stride = pl.C;
    ia.scratch_count = rt->d_scratch_count;
    ia.scratch_key = rt->d_scratch_key;
    ia.ckpt = rt->d_ckpt;

    FoldArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.p = ia.p;
    fill_ct_params(*cfg, fa.ct);
    fa.npix = rt->npix;
    fa.ckpt_stride = pl.C;
    fa.copies = rt->copies;
    fa.key_copies = rt->key_copies;
    fa.nan_count = pl.binned ? rt->d_nan_count : nullptr;
    fa.count = rt->d_count;
    fa.key = rt->d_key;
    fa.steps = rt->d_steps;
    fa.scratch_count = rt->d_scratch_count;
    fa.scratch_key = rt->d_scratch_key;
    fa.ckpt = rt->d_ckpt;
    fa.scalars = rt->d_scalars;

    const int mode = rt->measure_mode == 0 ? 2 : (rt->measure_mode == 1 ? 1 : 0);
    bool chunk_ahead = false;
    if (pl.binned && n_seg == 1 && n_jobs > pl.chunk_jobs && !rt->side) {  // so that the first chunk's iterate kernel is already marked
        HIP_TRY(hipStreamCreateWithFlags(&rt->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&rt->iter_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&rt->pf_done, hipEventDisableTiming));
    }
    for (uint64_t off = 0; off < n_jobs; off += pl.chunk_jobs) {
        const uint32_t m = static_cast<uint32_t>((n_jobs - off < pl.chunk_jobs) ? n_jobs - off : pl.chunk_jobs);
        ia.n_jobs = m;
        ia.starts = rt->d_starts + off * 3;
        fa.n_jobs = m;
        for (uint64_t s = 0; s < n_seg; ++s) {
            const uint64_t it = (s + 1 == n_seg) ? iters - s * seg : seg;
            const bool first = s == 0, carry = s + 1 < n_seg;
            ia.iters = it;
            fa.iters = it;
            if (pl.binned) {
                // the announced call: same start points, same job count; the first chunk's warm-up may already be done
                const bool announced = off == 0 && starts_on_device && rt->pf.valid && rt->pf.starts == starts && rt->pf.n_jobs == n_jobs;
                SAR_TRY(launch_binned_chunk(rt, pl, ia, fa, mode, first, carry, announced || chunk_ahead));
                chunk_ahead = false;
                // a call of several launch chunks (configs[3] on one GPU: three) announces its own next chunk: that chunk's
                // warm-up runs under this chunk's accumulate and fold (its start points are staged already)
                const uint64_t next = off + pl.chunk_jobs;
                if (n_seg == 1 && next < n_jobs && mode == 2 && rt->chunk_ahead != 2) {
                    const uint32_t m_next = static_cast<uint32_t>((n_jobs - next < pl.chunk_jobs) ? n_jobs - next : pl.chunk_jobs);
                    SAR_TRY(warmup_ahead(rt, ia.p, rt->d_starts + next * 3, true, m_next, it, false));
                    chunk_ahead = true;
                }
            } else {
                ia.resume = first ? 0u : 1u;
                ia.state_out = carry ? rt->d_starts + off * 3 : nullptr;
                span_begin(rt, rt->iter_spans, rt->iter_used);
                launch_iterate(ia, pl.block, pl.xcd_local, mode, rt->stream);
                span_end(rt, rt->iter_spans, rt->iter_used);
                ++rt->last_chunks;
                std::snprintf(rt->last_launch, sizeof(rt->last_launch), "k_iterate (one global atomic per visit%s)", pl.xcd_local ? ", per-XCD copies" : "");
                span_begin(rt, rt->fold_spans, rt->fold_used);
                launch_fold_resolve(fa, rt->stream);
                span_end(rt, rt->fold_spans, rt->fold_used);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    rt->last_iterations = static_cast<uint64_t>(n_jobs) * iters;
    return SAR_OK;
}

namespace {

}  // namespace

int sar::colorize_range(const sar_config* cfg, sar_runtime* rt, uint32_t first, uint32_t n, void* out_dev, bool global_scalars) {
    HIP_TRY(hipSetDevice(rt->device));
    if (first > rt->npix || n > rt->npix - first) { set_error("colorize: pixel range out of bounds"); return SAR_ERR_RANGE; }
    single_begin(rt, rt->colorize_span);
    if (cfg->render_kind == SAR_RENDER_GAS) {
        PaletteParams pal;
        std::memset(&pal, 0, sizeof(pal));
        pal.len = cfg->palette_len;
        for (uint32_t k = 0; k < cfg->palette_len; ++k)
            for (int ch = 0; ch < 3; ++ch) pal.rgb[k][ch] = cfg->palette_rgb[k][ch];
        for (int ch = 0; ch < 3; ++ch)  // Palette::new duplicates the last entry (:416-418)
            pal.rgb[cfg->palette_len][ch] = cfg->palette_rgb[cfg->palette_len - 1][ch];
        if (n)
            launch_colorize_gas(rt->d_count + first, rt->d_steps + first, rt->d_scalars, rt->d_lnlut, kLnLutEntries, pal,
                                cfg->brightness_offset, cfg->brightness_factor, cfg->transparent ? 1 : 0, n, out_dev, rt->stream);
    } else if (global_scalars) {
        if (n) launch_colorize_depth_range(rt->d_key + first, rt->d_scalars, n, out_dev, rt->stream);
    } else {
        launch_colorize_depth(rt->d_key + first, rt->d_scalars, n, out_dev, rt->stream);
    }
    single_end(rt, rt->colorize_span, rt->colorize_timed);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

namespace {

int do_colorize(const sar_config* cfg, sar_runtime* rt, void* out_dev) { return colorize_range(cfg, rt, 0, rt->npix, out_dev, false); }

int ensure_rgba(sar_runtime* rt) {
    if (!rt->d_rgba) HIP_TRY(hipMalloc(&rt->d_rgba, static_cast<size_t>(rt->npix) * 8));
    return SAR_OK;
}

}  // namespace

extern "C" {

int sar_device_count(int* out_count) {
    if (!out_count) return SAR_ERR_INVALID;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return SAR_ERR_NO_DEVICE;
    }
    *out_count = n;
    return SAR_OK;
}

int sar_runtime_new(const sar_config* cfg, int device, sar_runtime** out) {
    if (!out) return SAR_ERR_INVALID;
    *out = nullptr;
    SAR_TRY(validate(cfg));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return SAR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    sar_runtime* rt = new (std::nothrow) sar_runtime();
    if (!rt) return SAR_ERR_OOM;
    rt->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) rt->sm_count = static_cast<uint32_t>(prop.multiProcessorCount);
    int st = SAR_OK;
    auto fail = [&](int code) { sar_runtime_free(rt); return code; };
    if (hipStreamCreateWithFlags(&rt->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(SAR_ERR_HIP); }
    rt->own_stream = true;
    if (hipEventCreateWithFlags(&rt->starts_copied, hipEventDisableTiming) != hipSuccess) return fail(SAR_ERR_HIP);
    if (hipMalloc(&rt->d_scalars, SC_COUNT * sizeof(uint32_t)) != hipSuccess) return fail(SAR_ERR_OOM);
    if (hipMalloc(&rt->d_lnlut, kLnLutEntries * sizeof(double)) != hipSuccess) return fail(SAR_ERR_OOM);
    if (hipMemcpyAsync(rt->d_lnlut, host_ln_lut(), kLnLutEntries * sizeof(double), hipMemcpyHostToDevice, rt->stream) != hipSuccess)
        return fail(SAR_ERR_HIP);
    if ((st = alloc_image_buffers(rt, cfg->width, cfg->height)) != SAR_OK) return fail(st);
    if ((st = do_reset(rt)) != SAR_OK) return fail(st);
    rt->rng.seed(cfg->seed);
    if (hipStreamSynchronize(rt->stream) != hipSuccess) return fail(SAR_ERR_HIP);
    *out = rt;
    return SAR_OK;
}

int sar_runtime_free(sar_runtime* rt) {
    if (!rt) return SAR_OK;
    hipSetDevice(rt->device);
    if (rt->stream) hipStreamSynchronize(rt->stream);
    free_device_buffers(rt);
    if (rt->d_scalars) hipFree(rt->d_scalars);
    if (rt->d_lnlut) hipFree(rt->d_lnlut);
    if (rt->side) { hipStreamSynchronize(rt->side); hipStreamDestroy(rt->side); }
    if (rt->iter_done) hipEventDestroy(rt->iter_done);
    if (rt->pf_done) hipEventDestroy(rt->pf_done);
    for (hipEvent_t e : rt->img_events) if (e) hipEventDestroy(e);
    if (rt->d_warm) hipFree(rt->d_warm);
    if (rt->d_joblist) hipFree(rt->d_joblist);
    if (rt->d_active) hipFree(rt->d_active);
    if (rt->d_warm_alt) hipFree(rt->d_warm_alt);
    if (rt->d_joblist_alt) hipFree(rt->d_joblist_alt);
    if (rt->d_active_alt) hipFree(rt->d_active_alt);
    if (rt->d_hint_range_alt) hipFree(rt->d_hint_range_alt);
    if (rt->d_starts_alt) hipFree(rt->d_starts_alt);
    if (rt->d_seg_any) hipFree(rt->d_seg_any);
    if (rt->h_active) hipHostFree(rt->h_active);
    if (rt->active_copied) hipEventDestroy(rt->active_copied);
    if (rt->d_starts) hipFree(rt->d_starts);
    if (rt->h_starts) hipHostFree(rt->h_starts);
    if (rt->d_ckpt) hipFree(rt->d_ckpt);
    if (rt->d_arena) hipFree(rt->d_arena);
    if (rt->d_heads) hipFree(rt->d_heads);
    if (rt->d_nan_count) hipFree(rt->d_nan_count);
    if (rt->d_hint_range) hipFree(rt->d_hint_range);
    if (rt->starts_copied) hipEventDestroy(rt->starts_copied);
    for (auto& s : rt->iter_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (auto& s : rt->fold_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (auto& s : rt->warm_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    if (rt->colorize_span.a) { hipEventDestroy(rt->colorize_span.a); hipEventDestroy(rt->colorize_span.b); }
    if (rt->merge_span.a) { hipEventDestroy(rt->merge_span.a); hipEventDestroy(rt->merge_span.b); }
    if (rt->own_stream && rt->stream) hipStreamDestroy(rt->stream);
    delete rt;
    return SAR_OK;
}

int sar_runtime_reset(sar_runtime* rt) {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    return do_reset(rt);
}

int sar_runtime_set_width_height(sar_runtime* rt, uint32_t width, uint32_t height) {
    if (!rt) return SAR_ERR_INVALID;
    if (rt->W == width && rt->H == height) return SAR_OK;  // :668
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    SAR_TRY(alloc_image_buffers(rt, width, height));
    return do_reset(rt);
}

int sar_runtime_seed(sar_runtime* rt, uint64_t seed) {
    if (!rt) return SAR_ERR_INVALID;
    rt->rng.seed(seed);
    return SAR_OK;
}

int sar_runtime_merge(sar_runtime* dst, const sar_runtime* src) {
    if (!dst || !src) return SAR_ERR_INVALID;
    if (dst->W != src->W || dst->H != src->H) {  // assert_eq! in the reference (:709-710)
        set_error("merge: %ux%u vs %ux%u", dst->W, dst->H, src->W, src->H);
        return SAR_ERR_DIM_MISMATCH;
    }
    if (dst->device != src->device) { set_error("merge: runtimes live on different devices; use the exchange API"); return SAR_ERR_INVALID; }
    if (dst == src) { set_error("merge: dst and src are the same runtime"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(dst->device));
    if (src->stream != dst->stream) HIP_TRY(hipStreamSynchronize(src->stream));
    single_begin(dst, dst->merge_span);
    launch_merge(dst->d_count, dst->d_key, dst->d_steps, src->d_count, src->d_key, src->d_steps, dst->npix,
                 dst->d_scalars, dst->stream);
    single_end(dst, dst->merge_span, dst->merge_timed);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_runtime_synchronize(sar_runtime* rt) {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_runtime_dims(const sar_runtime* rt, uint32_t* width, uint32_t* height) {
    if (!rt || !width || !height) return SAR_ERR_INVALID;
    *width = rt->W;
    *height = rt->H;
    return SAR_OK;
}

int sar_runtime_set_stream(sar_runtime* rt, void* hip_stream) {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    if (rt->own_stream && rt->stream) hipStreamDestroy(rt->stream);
    rt->stream = static_cast<hipStream_t>(hip_stream);
    rt->own_stream = false;
    return SAR_OK;
}

int sar_runtime_get_stream(const sar_runtime* rt, void** hip_stream_out) {
    if (!rt || !hip_stream_out) return SAR_ERR_INVALID;
    *hip_stream_out = rt->stream;
    return SAR_OK;
}

int sar_render(const sar_config* cfg, sar_runtime* rt) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    double p0[3];
    rt->rng.start_point(p0);  // :748
    return render_chunked(cfg, rt, 1, cfg->iterations, p0);
}

int sar_render_jobs(const sar_config* cfg, sar_runtime* rt, const double* starts_xyz_host) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (cfg->jobs_total == 0) { set_error("jobs_total is 0"); return SAR_ERR_INVALID; }
    const uint64_t per_job = cfg->iterations / cfg->jobs_total;  // :1058
    std::vector<double> drawn;
    if (!starts_xyz_host) {
        drawn.resize(static_cast<size_t>(cfg->jobs_total) * 3);
        for (uint32_t k = 0; k < cfg->jobs_total; ++k) rt->rng.start_point(&drawn[3 * static_cast<size_t>(k)]);
        starts_xyz_host = drawn.data();
    }
    return render_chunked(cfg, rt, cfg->jobs_total, per_job, starts_xyz_host);
}

int sar_render_job_range(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                         const double* starts_xyz_host) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (n_jobs && !starts_xyz_host) { set_error("starts_xyz_host is NULL"); return SAR_ERR_INVALID; }
    return render_chunked(cfg, rt, n_jobs, iters_per_job, starts_xyz_host);
}

int sar_colorize_device(const sar_config* cfg, sar_runtime* rt, void* rgba_out_dev) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_dev) return SAR_ERR_INVALID;
    return do_colorize(cfg, rt, rgba_out_dev);
}

int sar_colorize(const sar_config* cfg, sar_runtime* rt, uint16_t* rgba_out_host) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    SAR_TRY(ensure_rgba(rt));
    SAR_TRY(do_colorize(cfg, rt, rt->d_rgba));
    HIP_TRY(hipMemcpyAsync(rgba_out_host, rt->d_rgba, static_cast<size_t>(rt->npix) * 8, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_render_job_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                                const double* starts_xyz_dev) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (n_jobs && !starts_xyz_dev) { set_error("starts_xyz_dev is NULL"); return SAR_ERR_INVALID; }
    return render_chunked(cfg, rt, n_jobs, iters_per_job, starts_xyz_dev, true);
}

int sar_runtime_describe_last_launch(const sar_runtime* rt, char* out, size_t cap) {
    if (!rt || !out || cap == 0) return SAR_ERR_INVALID;
    std::snprintf(out, cap, "%s | chunks=%u warmup_ahead=%u", rt->last_launch[0] ? rt->last_launch : "nothing launched", rt->last_chunks,
                  rt->prefetch_used);
    return SAR_OK;
}

int sar_runtime_prefetch_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                                const double* starts_xyz_dev) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!starts_xyz_dev) { set_error("starts_xyz_dev is NULL"); return SAR_ERR_INVALID; }
    rt->pf.valid = false;
    if (n_jobs == 0 || iters_per_job == 0) return SAR_OK;
    HIP_TRY(hipSetDevice(rt->device));
    const uint64_t max_ord = rt->max_ordinals ? rt->max_ordinals : kMaxChunkOrdinals;
    if (iters_per_job > max_ord) return SAR_OK;  // a job of several segments: nothing to run ahead
    LaunchPlan pl;
    SAR_TRY(plan_launch(cfg, rt, n_jobs, iters_per_job, pl));
    if (!pl.binned) return SAR_OK;
    const uint32_t m = static_cast<uint32_t>(n_jobs < pl.chunk_jobs ? n_jobs : pl.chunk_jobs);
    MapParams p;
    fill_map_params(*cfg, p);
    SAR_TRY(warmup_ahead(rt, p, starts_xyz_dev, false, m, iters_per_job, pl.hint_bytes == 2));
    rt->pf.n_jobs = n_jobs;
    rt->pf.starts = starts_xyz_dev;
    return SAR_OK;
}

int sar_runtime_extent(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                       const double* starts_xyz_host, double* out12) {
    if (!cfg || !rt || !out12 || n_jobs == 0) return SAR_ERR_INVALID;
    SAR_TRY(sar_config_validate(cfg));
    HIP_TRY(hipSetDevice(rt->device));
    std::vector<double> soa(static_cast<size_t>(n_jobs) * 3);
    for (uint32_t k = 0; k < n_jobs; ++k) {
        double p0[3];
        if (starts_xyz_host) std::memcpy(p0, starts_xyz_host + 3 * static_cast<size_t>(k), sizeof(p0));
        else rt->rng.start_point(p0);  // :748
        soa[k] = p0[0];
        soa[n_jobs + static_cast<size_t>(k)] = p0[1];
        soa[2 * static_cast<size_t>(n_jobs) + k] = p0[2];
    }
    const uint32_t blocks = (n_jobs + 255u) / 256u;
    double *d_starts = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_starts, soa.size() * sizeof(double)));
    if (hipMalloc(&d_out, static_cast<size_t>(blocks) * 12 * sizeof(double)) != hipSuccess) {
        hipFree(d_starts);
        set_error("out of device memory");
        return SAR_ERR_OOM;
    }
    MapParams mp;
    std::memset(&mp, 0, sizeof(mp));
    fill_map_params(*cfg, mp);
    std::vector<double> part(static_cast<size_t>(blocks) * 12);
    hipError_t e = hipMemcpyAsync(d_starts, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, rt->stream);
    if (e == hipSuccess) {
        launch_extent(mp, d_starts, n_jobs, iters_per_job, d_out, rt->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(part.data(), d_out, part.size() * sizeof(double), hipMemcpyDeviceToHost, rt->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(rt->stream);
    hipFree(d_starts);
    hipFree(d_out);
    if (e != hipSuccess) { set_error("sar_runtime_extent: %s", hipGetErrorString(e)); return SAR_ERR_HIP; }
    for (int k = 0; k < 12; ++k) {
        double v = part[k];
        for (uint32_t b = 1; b < blocks; ++b) {
            const double o = part[static_cast<size_t>(b) * 12 + k];
            v = (k & 1) ? (o > v ? o : v) : (o < v ? o : v);
        }
        out12[k] = v;
    }
    return SAR_OK;
}

int sar_image_convert_device(sar_runtime* rt, const void* rgba16_dev, int format, void* out_dev) {
    if (!rt || !rgba16_dev || !out_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (format == SAR_FMT_RGBA16) {
        HIP_TRY(hipMemcpyAsync(out_dev, rgba16_dev, static_cast<size_t>(rt->npix) * 8, hipMemcpyDeviceToDevice, rt->stream));
        return SAR_OK;
    }
    if (launch_convert(rgba16_dev, format, out_dev, rt->npix, rt->stream) != 0) {
        set_error("unknown image format %d", format);
        return SAR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

static int enqueue_colorize_format(const sar_config* cfg, sar_runtime* rt, int format, void* out_host) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    const size_t bytes = sar_image_bytes(format, rt->W, rt->H);
    if (!out_host || bytes == 0) { set_error("sar_colorize_format: bad format or NULL output"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    SAR_TRY(ensure_rgba(rt));
    SAR_TRY(do_colorize(cfg, rt, rt->d_rgba));
    const void* src = rt->d_rgba;
    if (format != SAR_FMT_RGBA16) {
        if (!rt->d_export) HIP_TRY(hipMalloc(&rt->d_export, static_cast<size_t>(rt->npix) * 6));  // largest converted format
        SAR_TRY(sar_image_convert_device(rt, rt->d_rgba, format, rt->d_export));
        src = rt->d_export;
    }
    // d_rgba / d_export are written again by the next frame's colorize, which the stream orders behind this copy
    HIP_TRY(hipMemcpyAsync(out_host, src, bytes, hipMemcpyDeviceToHost, rt->stream));
    return SAR_OK;
}

int sar_colorize_format(const sar_config* cfg, sar_runtime* rt, int format, void* out_host) {
    SAR_TRY(enqueue_colorize_format(cfg, rt, format, out_host));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_colorize_format_async(const sar_config* cfg, sar_runtime* rt, int format, void* out_host, uint64_t* ticket_out) {
    if (!ticket_out) { set_error("sar_colorize_format_async: NULL ticket"); return SAR_ERR_INVALID; }
    SAR_TRY(enqueue_colorize_format(cfg, rt, format, out_host));
    hipEvent_t& ev = rt->img_events[rt->img_next % 8];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, rt->stream));
    *ticket_out = rt->img_next++;
    return SAR_OK;
}

int sar_runtime_wait_image(sar_runtime* rt, uint64_t ticket) {
    if (!rt || ticket >= rt->img_next) { set_error("sar_runtime_wait_image: no such ticket"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipEventSynchronize(rt->img_events[ticket % 8]));
    return SAR_OK;
}

int sar_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) { set_error("sar_host_alloc: NULL output or zero size"); return SAR_ERR_INVALID; }
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SAR_OK;
}

int sar_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p));
    return SAR_OK;
}

int sar_runtime_count(sar_runtime* rt, uint32_t* out_host) {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_count, static_cast<size_t>(rt->npix) * 4, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_runtime_steps(sar_runtime* rt, double* out_host) {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_steps, static_cast<size_t>(rt->npix) * 8, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_runtime_zbuf(sar_runtime* rt, float* out_host) {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (!rt->d_ztmp) HIP_TRY(hipMalloc(&rt->d_ztmp, static_cast<size_t>(rt->npix) * 4));
    launch_zbuf_out(rt->d_key, rt->d_ztmp, rt->npix, rt->stream);
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_ztmp, static_cast<size_t>(rt->npix) * 4, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_runtime_max(sar_runtime* rt, uint32_t* out_max) {
    if (!rt || !out_max) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    uint32_t sc[SC_COUNT];
    HIP_TRY(hipMemcpyAsync(sc, rt->d_scalars, sizeof(sc), hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    *out_max = sc[SC_WRAP] ? 0xFFFFFFFFu : sc[SC_MAX];
    return SAR_OK;
}

int sar_runtime_load(sar_runtime* rt, const uint32_t* count_host, const double* steps_host,
                     const float* zbuf_host, uint32_t max) {
    if (!rt || !count_host || !steps_host || !zbuf_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (!rt->d_ztmp) HIP_TRY(hipMalloc(&rt->d_ztmp, static_cast<size_t>(rt->npix) * 4));
    HIP_TRY(hipMemcpyAsync(rt->d_count, count_host, static_cast<size_t>(rt->npix) * 4, hipMemcpyHostToDevice, rt->stream));
    HIP_TRY(hipMemcpyAsync(rt->d_steps, steps_host, static_cast<size_t>(rt->npix) * 8, hipMemcpyHostToDevice, rt->stream));
    HIP_TRY(hipMemcpyAsync(rt->d_ztmp, zbuf_host, static_cast<size_t>(rt->npix) * 4, hipMemcpyHostToDevice, rt->stream));
    launch_zbuf_in(rt->d_ztmp, rt->d_key, rt->npix, rt->stream);
    SAR_TRY(clear_hints(rt));
    uint32_t sc[SC_COUNT] = {0};
    sc[SC_MAX] = max;
    HIP_TRY(hipMemcpyAsync(rt->d_scalars, sc, sizeof(sc), hipMemcpyHostToDevice, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
}

int sar_runtime_exchange_export(sar_runtime* rt, uint32_t rank, void* key_i64_out_dev) {
    if (!rt || !key_i64_out_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_export(rt->d_key, rank, key_i64_out_dev, rt->npix, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_runtime_exchange_select(sar_runtime* rt, uint32_t rank, const void* key_i64_reduced_dev,
                                void* sum_i32_out_dev) {
    if (!rt || !key_i64_reduced_dev || !sum_i32_out_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_select(rt->d_count, rt->d_key, rt->d_steps, rank, key_i64_reduced_dev, sum_i32_out_dev, rt->npix,
                       rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_runtime_exchange_import(sar_runtime* rt, const void* key_i64_reduced_dev, const void* sum_i32_reduced_dev) {
    if (!rt || !key_i64_reduced_dev || !sum_i32_reduced_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_import(rt->d_count, rt->d_key, rt->d_steps, key_i64_reduced_dev, sum_i32_reduced_dev, rt->npix,
                       rt->d_scalars, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_bin_geometry(uint32_t width, uint32_t height, uint32_t bin_shift, uint32_t bin_interleave, uint32_t out[8]) {
    if (!out || width == 0 || height == 0 || (bin_shift && (bin_shift < 12 || bin_shift > 16)) || bin_interleave > 2) return SAR_ERR_INVALID;
    const uint64_t npix64 = static_cast<uint64_t>(width) * height;
    if (npix64 > 0x7FFFFFFFull) return SAR_ERR_RANGE;  // what a runtime accepts (alloc_image_buffers)
    BinGeometry g;
    if (bin_shift == 0 && bin_interleave == 0) {  // what choose_chunk_records picks for a launch that wants two waves per SIMD
        bool found = false;
        for (uint32_t cand : {60u, 28u}) {
            for (uint32_t sh : {15u, 16u}) {
                g = bin_geometry(static_cast<uint32_t>(npix64), 256u, sh, 0u, 12u, true, 2u);
                if (g.ok && g.interleaved && lean_wave_lds_bytes(g.bins, cand, true) * 8u <= 160u * 1024u) { found = true; break; }
            }
            if (found) break;
        }
        if (!found) g = bin_geometry(static_cast<uint32_t>(npix64), 256u, 0u, 0u, 12u, false, 0u);
    } else {
        g = bin_geometry(static_cast<uint32_t>(npix64), 256u, bin_shift, 0u, 12u, false, bin_interleave);
    }
    out[0] = g.ok ? 1u : 0u;
    out[1] = g.bins;
    out[2] = g.shift;
    out[3] = g.interleaved ? 1u : 0u;
    out[4] = g.map.seg_shift;
    out[5] = g.map.bin_bits;
    out[6] = g.map.hi_shift;
    out[7] = g.map.low_mask;
    return SAR_OK;
}

// ---- sliced exchange (all-to-all of pixel slices; see include/sar.h) ---------------------------------------

int sar_exchange_slice_pixels(uint32_t npix, uint32_t world, uint32_t* out_slice_pixels) {
    if (!out_slice_pixels || world == 0) return SAR_ERR_INVALID;
    const uint64_t s = ((static_cast<uint64_t>(npix) + world - 1) / world + 3u) & ~3ull;
    if (s * world > 0xFFFFFFFFull) { set_error("slice geometry exceeds 2^32 pixels"); return SAR_ERR_RANGE; }
    *out_slice_pixels = static_cast<uint32_t>(s);
    return SAR_OK;
}

int sar_runtime_exchange_pack(sar_runtime* rt, uint32_t world, void* blocks_out_dev) {
    if (!rt || !blocks_out_dev) return SAR_ERR_INVALID;
    uint32_t S = 0;
    SAR_TRY(sar_exchange_slice_pixels(rt->npix, world, &S));
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_pack(rt->d_count, rt->d_key, rt->d_steps, rt->npix, S, world, blocks_out_dev, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_runtime_exchange_merge_slices(sar_runtime* rt, uint32_t world, uint32_t rank, const void* blocks_in_dev) {
    if (!rt || !blocks_in_dev || rank >= world) return SAR_ERR_INVALID;
    uint32_t S = 0;
    SAR_TRY(sar_exchange_slice_pixels(rt->npix, world, &S));
    HIP_TRY(hipSetDevice(rt->device));
    const uint64_t first = static_cast<uint64_t>(rank) * S;
    const uint32_t n = first >= rt->npix ? 0u : static_cast<uint32_t>((rt->npix - first < S) ? rt->npix - first : S);
    launch_exch_merge_slices(rt->d_count, rt->d_key, rt->d_steps, n ? static_cast<uint32_t>(first) : 0u, n, S, world, blocks_in_dev,
                             rt->d_scalars, rank == 0, rt->stream);
    HIP_TRY(hipGetLastError());  // (the depth hints stay valid: a merge only raises zbuf)
    return SAR_OK;
}

int sar_runtime_exchange_scalars_export(sar_runtime* rt, void* i64x4_out_dev) {
    if (!rt || !i64x4_out_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_scalars_export(rt->d_scalars, i64x4_out_dev, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_runtime_exchange_scalars_import(sar_runtime* rt, const void* i64x4_dev) {
    if (!rt || !i64x4_dev) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_scalars_import(rt->d_scalars, i64x4_dev, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

int sar_colorize_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t first_px, uint32_t n_px, void* rgba_out_dev) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_dev) return SAR_ERR_INVALID;
    return colorize_range(cfg, rt, first_px, n_px, rgba_out_dev, true);
}

// ---- measurement --------------------------------------------------------------------------------------

int sar_runtime_enable_timing(sar_runtime* rt, int enabled) {
    if (!rt) return SAR_ERR_INVALID;
    rt->timing = enabled != 0;
    return SAR_OK;
}

int sar_runtime_last_timing(sar_runtime* rt, sar_timing* out) {
    if (!rt || !out) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    std::memset(out, 0, sizeof(*out));
    float ms = 0.f;
    for (size_t k = 0; k < rt->iter_used; ++k)
        if (hipEventElapsedTime(&ms, rt->iter_spans[k].a, rt->iter_spans[k].b) == hipSuccess) out->iterate_ms += ms;
    for (size_t k = 0; k < rt->fold_used; ++k)
        if (hipEventElapsedTime(&ms, rt->fold_spans[k].a, rt->fold_spans[k].b) == hipSuccess) out->resolve_ms += ms;
    for (size_t k = 0; k < rt->warm_used; ++k)
        if (hipEventElapsedTime(&ms, rt->warm_spans[k].a, rt->warm_spans[k].b) == hipSuccess) out->warmup_ms += ms;
    if (rt->colorize_timed && hipEventElapsedTime(&ms, rt->colorize_span.a, rt->colorize_span.b) == hipSuccess) out->colorize_ms = ms;
    if (rt->merge_timed && hipEventElapsedTime(&ms, rt->merge_span.a, rt->merge_span.b) == hipSuccess) out->merge_ms = ms;
    out->iterate_launches = static_cast<uint32_t>(rt->iter_used);
    if (rt->d_nan_count) {  // cumulative statistic of the binned path; cleared by reading
        unsigned long long sent = 0, passed = 0;
        HIP_TRY(hipMemcpy(&sent, rt->d_nan_count + 1, sizeof(sent), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->d_nan_count + 1, 0, sizeof(sent)));
        HIP_TRY(hipMemcpy(&passed, rt->d_nan_count + 14, sizeof(passed), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->d_nan_count + 14, 0, sizeof(passed)));
        out->depth_candidates = passed;
#ifdef SAR_EXPERIMENT_PROF
        unsigned long long seg[11];
        HIP_TRY(hipMemcpy(seg, rt->d_nan_count + 2, sizeof(seg), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->d_nan_count + 2, 0, sizeof(seg)));
        const double tot = static_cast<double>(seg[0] + seg[1] + seg[2] + seg[3]);
        if (tot > 0)
            std::fprintf(stderr, "[prof] wave-cycles: map+projection %.1f%%  place_visit %.1f%%  depth %.1f%%  stores+requests %.1f%%  (total %.3g)\n",
                         100. * seg[0] / tot, 100. * seg[1] / tot, 100. * seg[2] / tot, 100. * seg[3] / tot, tot);
        const double ptot = static_cast<double>(seg[4] + seg[5]);
        double ctot = 0;
        for (int i = 6; i < 11; ++i) ctot += static_cast<double>(seg[i]);
        if (ptot > 0 && ctot > 0)
            std::fprintf(stderr, "[prof-split] producer: map+projection+hand-over %.1f%%  barrier %.1f%% (total %.4g) | consumer: barrier %.1f%%  hand-over read %.1f%%  "
                                 "place_visit %.1f%%  depth settle %.1f%%  slot+hint request %.1f%% (total %.4g)\n",
                         100. * seg[4] / ptot, 100. * seg[5] / ptot, ptot, 100. * seg[6] / ctot, 100. * seg[7] / ctot, 100. * seg[8] / ctot,
                         100. * seg[9] / ctot, 100. * seg[10] / ctot, ctot);
        unsigned long long dp[3];
        HIP_TRY(hipMemcpy(dp, rt->d_nan_count + 13, sizeof(dp), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset(rt->d_nan_count + 13, 0, sizeof(unsigned long long)));
        HIP_TRY(hipMemset(rt->d_nan_count + 15, 0, sizeof(unsigned long long)));
        if (ctot > 0)
            std::fprintf(stderr, "[prof-depth] of the consumer's time (memory drained at each mark): stage 2 (key wait, atomic, hint store, drain) %.1f%%  "
                                 "stage 1 (compare, key load, drain) %.1f%%\n", 100. * dp[0] / ctot, 100. * dp[2] / ctot);
#endif
        out->depth_atomics = sent;
    }
    out->iterations_counted = rt->last_iterations;
    if (rt->timing_accumulate) {
        rt->last_iterations = 0;
        rt->iter_used = 0;
        rt->fold_used = 0;
        rt->warm_used = 0;
    }
    return SAR_OK;
}

int sar_runtime_set_option(sar_runtime* rt, const char* name, uint64_t value) {
    if (!rt || !name) return SAR_ERR_INVALID;
    const uint32_t v = static_cast<uint32_t>(value);
    if (!std::strcmp(name, "block_threads")) {
        if (v == 0) { rt->block_threads = kDefaultBlock; return SAR_OK; }
        if (v % 64 || v > 256) { set_error("block_threads must be 64, 128, 192 or 256"); return SAR_ERR_INVALID; }
        rt->block_threads = v;
    } else if (!std::strcmp(name, "checkpoint_stride")) {
        rt->ckpt_stride = v ? v : kDefaultCkptStride;
    } else if (!std::strcmp(name, "path")) {
        if (v > 3) { set_error("path must be 0..3"); return SAR_ERR_INVALID; }
        rt->bins_mode = v;
    } else if (!std::strcmp(name, "bin_shift")) {
        if (v && (v < 12 || v > 16)) { set_error("bin_shift must be 12..16"); return SAR_ERR_INVALID; }
        rt->bin_shift = v;
    } else if (!std::strcmp(name, "bin_interleave")) {
        if (v > 2) { set_error("bin_interleave must be 0 (automatic), 1 (consecutive-pixel bins) or 2 (interleaved bins)"); return SAR_ERR_INVALID; }
        rt->bin_interleave = v;
    } else if (!std::strcmp(name, "splits")) {
        if (v > 16) { set_error("splits must be 1..16"); return SAR_ERR_INVALID; }
        rt->splits = v;
    } else if (!std::strcmp(name, "chunk_records")) {
        if (v && v != 12 && v != 20 && v != 28 && v != 60) { set_error("chunk_records must be 12, 20, 28 or 60"); return SAR_ERR_INVALID; }
        rt->chunk_records = v;
    } else if (!std::strcmp(name, "stager")) {
        if (v > 2) { set_error("stager must be 0 (automatic), 1 (copy-out by the filling lane) or 2 (buffer pool, cooperative copy-out)"); return SAR_ERR_INVALID; }
        rt->stager = v;
    } else if (!std::strcmp(name, "hint_bits")) {
        if (v && v != 16 && v != 32) { set_error("hint_bits must be 16 or 32"); return SAR_ERR_INVALID; }
        rt->hint_bits = v;
    } else if (!std::strcmp(name, "depth_pipe")) {
        if (v > 2) { set_error("depth_pipe must be 1 or 2"); return SAR_ERR_INVALID; }
        rt->depth_pipe = v;
    } else if (!std::strcmp(name, "split_waves")) {
        if (v > 2) { set_error("split_waves must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->split_waves = v;
    } else if (!std::strcmp(name, "hint_tile")) {
        if (v > 1) { set_error("hint_tile must be 0 (automatic) or 1 (row-major hints)"); return SAR_ERR_INVALID; }
        if (rt->hint_tile != v && rt->d_zhint) { HIP_TRY(hipSetDevice(rt->device)); SAR_TRY(clear_hints(rt)); }  // another layout: the old hints mean nothing
        rt->hint_tile = v;
    } else if (!std::strcmp(name, "acc_lists")) {
        if (v && v != 1 && v != 2 && v != 4 && v != 8) { set_error("acc_lists must be 1, 2, 4 or 8"); return SAR_ERR_INVALID; }
        rt->acc_lists = v;
    } else if (!std::strcmp(name, "hint_shared")) {
        if (v > 2) { set_error("hint_shared must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->hint_shared = v;
    } else if (!std::strcmp(name, "chunk_ahead")) {
        if (v > 2) { set_error("chunk_ahead must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->chunk_ahead = v;
    } else if (!std::strcmp(name, "acc_halves")) {
        rt->acc_halves = v ? 1u : 0u;
    } else if (!std::strcmp(name, "acc_threads")) {
        if (v && v != 256 && v != 512 && v != 1024) { set_error("acc_threads must be 256, 512 or 1024"); return SAR_ERR_INVALID; }
        rt->acc_threads = v;
    } else if (!std::strcmp(name, "measure")) {
        if (v > 2) { set_error("measure must be 0..2"); return SAR_ERR_INVALID; }
        rt->measure_mode = v;
    } else if (!std::strcmp(name, "timing_accumulate")) {
        rt->timing_accumulate = v != 0;
        rt->last_iterations = 0;
        rt->iter_used = 0;
        rt->fold_used = 0;
        rt->warm_used = 0;
    } else if (!std::strcmp(name, "debug_chunk_jobs")) {
        rt->debug_chunk_jobs = v;
    } else if (!std::strcmp(name, "debug_max_ordinals")) {
        rt->max_ordinals = value > kMaxChunkOrdinals ? kMaxChunkOrdinals : value;  // test hook: visits one launch may order
    } else {
        set_error("unknown option '%s'", name);
        return SAR_ERR_INVALID;
    }
    return SAR_OK;
}

}  // extern "C"
