// sar_render.cpp — a render call on one device: staging of the start points, the device buffers of the binned path, one
// launch chunk (warm-up + packing, iterate, accumulate, fold), the announced warm-up that runs ahead, and the ABI entry points
// of `render` (reference src/lib.rs:747-838). Host logic only; the arithmetic is in the kernel files.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "sar_plan.hpp"

using namespace sar;

void sar::fill_map_params(const sar_config& cfg, MapParams& p) {
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

void sar::fill_ct_params(const sar_config& cfg, ColorTransformParams& ct) {
    ct.kind = cfg.color_transform;
    ct._pad = 0;
    ct.offset = cfg.ct_offset;
    ct.factor = cfg.ct_factor;
    ct.ccx = cfg.center_camera[0];
    ct.ccy = cfg.center_camera[1];
}


// the scratch the iterate / accumulate kernels write and k_fold_resolve folds (and clears): `copies` partial histograms
// (one per accumulate workgroup of a bin; one on the atomic path) and one array of depth keys
int sar::ensure_scratch(sar_runtime* rt, uint32_t copies) {
    if (rt->copies != copies || !rt->d_scratch_count) {
        if (rt->d_scratch_count) dev_free(rt, rt->d_scratch_count);
        rt->d_scratch_count = nullptr;
        rt->copies = 0;
        const size_t n = static_cast<size_t>(copies) * rt->npix;
        HIP_TRY(dev_alloc(rt, &rt->d_scratch_count, n * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(rt->d_scratch_count, 0, n * sizeof(uint32_t), rt->stream));
        rt->copies = copies;
    }
    if (!rt->d_scratch_key) {
        HIP_TRY(dev_alloc(rt, &rt->d_scratch_key, static_cast<size_t>(rt->npix) * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(rt->d_scratch_key, 0, static_cast<size_t>(rt->npix) * sizeof(unsigned long long), rt->stream));
    }
    return SAR_OK;
}

// Start points into rt->d_starts, laid out as consecutive per-chunk SoA blocks x[m] y[m] z[m]; `starts` is the caller's
// [n_jobs][3] array in host memory (through one pinned staging buffer) or already in device memory.
// With `upload` (a batched launch: the leader's upload stream) the copy runs there, behind the last kernel known to have read
// d_starts, and the launch stream only waits for its event: the upload of the next batch runs under the current one.
// With `in_place` nothing is copied: the caller's kernel reads the page-locked staging buffer itself (device-visible host
// memory; 1.5 MB over PCIe under a kernel of a thousand iterations per point) and records starts_copied behind that kernel.
int sar::stage_starts(sar_runtime* rt, const LaunchPlan& pl, uint32_t n_jobs, const double* starts, bool on_device, hipStream_t upload,
                      bool in_place) {
    const size_t need = static_cast<size_t>(n_jobs) * 3;
    if (rt->starts_pending) {  // the previous call's upload still reads the staging buffer
        HIP_TRY(hipEventSynchronize(rt->starts_copied));
        rt->starts_pending = false;
    }
    if (need > rt->starts_cap) {
        if (rt->h_starts) host_free(rt, rt->h_starts);
        if (rt->d_starts) dev_free(rt, rt->d_starts);
        rt->h_starts = nullptr;
        rt->d_starts = nullptr;
        rt->starts_cap = 0;
        // (two doubles more: k_batch_fetch moves 16-byte pieces)
        HIP_TRY(host_alloc(rt, &rt->h_starts, (need + 2) * sizeof(double)));
        HIP_TRY(dev_alloc(rt, &rt->d_starts, (need + 2) * sizeof(double)));
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
    if (in_place) return SAR_OK;
    if (upload) {
        if (rt->starts_consumed_recorded) HIP_TRY(hipStreamWaitEvent(upload, rt->starts_consumed, 0));
        HIP_TRY(hipMemcpyAsync(rt->d_starts, rt->h_starts, need * sizeof(double), hipMemcpyHostToDevice, upload));
        HIP_TRY(hipEventRecord(rt->starts_copied, upload));
        HIP_TRY(hipStreamWaitEvent(rt->stream, rt->starts_copied, 0));
        rt->starts_pending = true;
        return SAR_OK;
    }
    HIP_TRY(hipMemcpyAsync(rt->d_starts, rt->h_starts, need * sizeof(double), hipMemcpyHostToDevice, rt->stream));
    HIP_TRY(hipEventRecord(rt->starts_copied, rt->stream));
    rt->starts_pending = true;
    return SAR_OK;
}

// One hint array per XCD lets every XCD's L2 serve its own hints coherently; but eight copies of a 4096^2 image's hints
// (268 MB at 16 bits) no longer fit the 256 MB Infinity Cache behind the L2s, and the misses go to HBM. From 200 MB on
// the XCDs share ONE array: an XCD then sees another's updates only when its own L2 drops the line — a stale hint lets
// more visits through stage 1, never a wrong one — and the misses stay on chip (4096^2 share: 9.15 -> 8.85 ms; below
// that size sharing costs: 2048^2 5.90 -> 6.05 ms).
// one_hint_array: a batched frame that runs on one or two XCDs of its own — what its XCDs share is all there is; a runtime of a
// frame group holds one array for that reason and shares it in its other launches as well (a sweep's last frame or two).
bool sar::hints_shared(const sar_runtime* rt, const sar_runtime* opt, const LaunchPlan& pl, bool one_hint_array) {
    if (opt->hint_shared == 2) return true;
    if (opt->hint_shared == 1) return false;
    return one_hint_array || rt->single_hint_array || static_cast<uint64_t>(rt->npix) * pl.hint_bytes * 8u > (200ull << 20);
}

// Device buffers of the binned path: record arena, list heads, depth hints (hint_copies arrays: one per XCD, or one), warm-up
// output, counters.
int sar::ensure_binned_buffers(sar_runtime* rt, const LaunchPlan& pl, uint32_t hint_copies) {
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
        const int rc = grow_device(rt, arena, rt->arena_cap, static_cast<size_t>(pl.arena_waves) * pl.chunks_per_wave * chunk_bytes(pl.R));
        rt->d_arena = arena;  // also when the allocation failed: the old buffer is gone
        SAR_TRY(rc);
    }
    SAR_TRY(grow_device(rt, rt->d_heads, rt->heads_cap, static_cast<size_t>(pl.max_waves) * pl.geo.bins));
    if (!rt->d_zhint || rt->zhint_bytes != pl.hint_bytes || rt->hint_copies_alloc < hint_copies) {
        // (hints only ever reject visits that cannot win: starting over with empty ones is always right. What may still read the
        // old arrays is on this runtime's streams; hipFree waits for the device, memory of a frame group is simply left behind)
        if (rt->d_zhint) dev_free(rt, rt->d_zhint);
        rt->d_zhint = nullptr;
        rt->hint_copies_alloc = 0;
        HIP_TRY(dev_alloc(rt, &rt->d_zhint, (static_cast<size_t>(rt->npix) + 2u) * hint_copies * pl.hint_bytes));
        rt->zhint_bytes = pl.hint_bytes;
        rt->hint_copies_alloc = hint_copies;
        rt->hint_copies_used = hint_copies;  // fresh memory: all of it
        SAR_TRY(clear_hints(rt));
    }
    if (pl.chunk_jobs > rt->warm_cap) {
        size_t cap3 = 0, cap1 = 0;  // both buffers are replaced together
        rt->warm_cap = 0;
        SAR_TRY(grow_device(rt, rt->d_warm, cap3, static_cast<size_t>(pl.chunk_jobs) * 3));
        SAR_TRY(grow_device(rt, rt->d_joblist, cap1, static_cast<size_t>(pl.chunk_jobs)));
        rt->warm_cap = pl.chunk_jobs;
    }
    if (!rt->d_active) HIP_TRY(dev_alloc(rt, &rt->d_active, 4 * sizeof(uint32_t)));
    const size_t segs = static_cast<size_t>(rt->npix) / 2048u + 1u;
    if (rt->seg_any_cap < segs) {
        if (rt->d_seg_any) dev_free(rt, rt->d_seg_any);
        rt->d_seg_any = nullptr;
        rt->seg_any_cap = 0;
        HIP_TRY(dev_alloc(rt, &rt->d_seg_any, segs * sizeof(uint32_t)));
        rt->seg_any_cap = segs;
    }
    if (!rt->h_active) {
        HIP_TRY(host_alloc(rt, &rt->h_active, sizeof(uint32_t)));
        *rt->h_active = 0;
        HIP_TRY(hipEventCreateWithFlags(&rt->active_copied, hipEventDisableTiming));
    }
    if (!rt->d_hint_range) {
        HIP_TRY(dev_alloc(rt, &rt->d_hint_range, 2 * sizeof(uint32_t)));
        HIP_TRY(hipMemsetAsync(rt->d_hint_range, 0, 2 * sizeof(uint32_t), rt->stream));
    }
    if (!rt->d_nan_count) {
        // [0] NaN iterations, [1] depth atomics (stat), [2..5] segment cycles of the SAR_EXPERIMENT_PROF build
        // [6..7] producer wave, [8..12] consumer wave of k_iterate_split in that build
        HIP_TRY(dev_alloc(rt, &rt->d_nan_count, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(rt->d_nan_count, 0, 16 * sizeof(unsigned long long), rt->stream));
    }
    return SAR_OK;
}

// The argument blocks of one launch of the binned path on `rt` (the launch options — hint sharing, hint tiles — are `opt`'s:
// rt itself, or the leader of a batch).
void sar::fill_bin_iter_args(sar_runtime* rt, const sar_runtime* opt, const LaunchPlan& pl, const IterArgs& ia, BinIterArgs& ba, bool* shared_out,
                             bool one_hint_array) {
    const uint32_t m = ia.n_jobs;
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
    const bool share = hints_shared(rt, opt, pl, one_hint_array);
    ba.hint_copy_mask = share ? 0u : 7u;
    const uint32_t written = share ? 1u : 8u;
    if (rt->hint_copies_used < written) rt->hint_copies_used = written;
    if (shared_out) *shared_out = share;
    // narrow hints of an image whose width is a power of two: 8 x 8 tiles per 128-byte line (HintTile); the permutation stays
    // inside blocks of eight rows, so the height must be a multiple of eight
    const bool pow2w = (rt->W & (rt->W - 1u)) == 0u && rt->W >= 8u && rt->H % 8u == 0u;
    if (pl.hint_bytes == 2 && pow2w && opt->hint_tile != 1u) {
        uint32_t b = 0;
        while ((1u << b) < rt->W) ++b;
        ba.tile.shift1 = b - 3u;
        ba.tile.mask1 = 0x38u;
        ba.tile.mask2 = ((1u << (b + 3u)) - 1u) & ~7u;
    }
}

void sar::fill_bin_acc_args(sar_runtime* rt, const LaunchPlan& pl, const BinIterArgs& ba, BinAccArgs& ca) {
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
}

void sar::describe_launch(sar_runtime* rt, const LaunchPlan& pl, bool share, uint32_t batch_frames, uint32_t xcd_map) {
    char batch[64] = "";
    if (batch_frames) std::snprintf(batch, sizeof(batch), " | batch of %u frames (xcd map %u)", batch_frames, xcd_map);
    std::snprintf(rt->last_launch, sizeof(rt->last_launch),
                  "%s R=%u bins=%ux%upx %s hints=%s pipe=%u | k_bin_accumulate splits=%u lists=%u counters=%s%s",
                  pl.split ? "k_iterate_split" : "k_iterate_lean", pl.R, pl.geo.bins, 1u << pl.geo.shift, pl.geo.interleaved ? "interleaved" : "consecutive",
                  pl.hint_bytes == 4 ? (share ? "f32/chip" : "f32") : (share ? "q16/chip" : "q16"), kDefaultDepthPipe, pl.splits, pl.acc_lists,
                  pl.geo.shift == 16u ? "u16-packed" : "u32", batch);
}

// what every launch chunk of a render call on `rt` shares: the map, the image, the scratch and the persistent buffers
void sar::fill_iter_fold_args(const sar_config* cfg, sar_runtime* rt, const LaunchPlan& pl, IterArgs& ia, FoldArgs& fa) {
    std::memset(&ia, 0, sizeof(ia));
    fill_map_params(*cfg, ia.p);
    ia.width = rt->W;
    ia.npix = rt->npix;
    ia.ckpt_stride = pl.C;
    ia.scratch_count = rt->d_scratch_count;
    ia.scratch_key = rt->d_scratch_key;
    ia.ckpt = rt->d_ckpt;

    std::memset(&fa, 0, sizeof(fa));
    fa.p = ia.p;
    fill_ct_params(*cfg, fa.ct);
    fa.npix = rt->npix;
    fa.ckpt_stride = pl.C;
    fa.copies = rt->copies;
    fa.key_copies = 1;
    fa.nan_count = pl.binned ? rt->d_nan_count : nullptr;
    fa.count = rt->d_count;
    fa.key = rt->d_key;
    fa.steps = rt->d_steps;
    fa.scratch_count = rt->d_scratch_count;
    fa.scratch_key = rt->d_scratch_key;
    fa.ckpt = rt->d_ckpt;
    fa.scalars = rt->d_scalars;
}

sar::WarmArgs sar::warm_args(const sar::MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* warm, uint32_t* joblist,
                             uint32_t* active, uint32_t width, uint32_t* hint_range) {
    sar::WarmArgs w;
    std::memset(&w, 0, sizeof(w));
    w.p = p;
    w.starts = starts;
    w.n_jobs = n_jobs;
    w.width = width;
    w.iters = iters;
    w.warm = warm;
    w.joblist = joblist;
    w.active = active;
    w.nan_count = reinterpret_cast<unsigned long long*>(active + 2);
    w.hint_range = hint_range;
    w.n_iter = 1000u;  // "skip first 1000 to get good values in the attractor" (:750-752)
    return w;
}

namespace {

// One launch chunk of the binned path: warm-up + packing, iterate, accumulate, fold.
// `first`: the first segment of these jobs (warm-up + packing); `carry`: more segments follow (keep the trajectory state).
int launch_binned_chunk(sar_runtime* rt, const LaunchPlan& pl, const IterArgs& ia, const FoldArgs& fa_in, bool first, bool carry,
                        bool use_prefetch) {
    FoldArgs fa = fa_in;
    fa.seg_any = rt->d_seg_any;
    const uint32_t m = ia.n_jobs;
    BinIterArgs ba;
    bool share = false;
    fill_bin_iter_args(rt, rt, pl, ia, ba, &share);
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
        launch_warmup(warm_args(ia.p, ia.starts, m, ia.iters, rt->d_warm, rt->d_joblist, rt->d_active, ia.width, measure), rt->stream);
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
    if (launch_iterate_lean(ba, pl.block, pl.R, pl.hint_bytes, pl.split, rt->stream) != 0) {
        set_error("no iterate kernel for chunk_records %u / %u-byte hints", pl.R, pl.hint_bytes);
        return SAR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    span_end(rt, rt->iter_spans, rt->iter_used);
    ++rt->last_chunks;
    describe_launch(rt, pl, share, 0);
    if (rt->iter_done) {  // an announced call's warm-up starts here, under this launch's accumulate and fold
        HIP_TRY(hipEventRecord(rt->iter_done, rt->stream));
        rt->iter_done_recorded = true;
    }
    BinAccArgs ca;
    fill_bin_acc_args(rt, pl, ba, ca);
    span_begin(rt, rt->fold_spans, rt->fold_used);
    HIP_TRY(hipMemsetAsync(rt->d_seg_any, 0, (static_cast<size_t>(rt->npix) / 2048u + 1u) * sizeof(uint32_t), rt->stream));
    if (launch_bin_accumulate(ca, rt->acc_threads, pl.R, pl.acc_lists, rt->stream) != 0) {
        set_error("no accumulate kernel for chunk_records %u / %u lists per lane group", pl.R, pl.acc_lists);
        return SAR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    launch_fold_resolve(fa, rt->stream);
    span_end(rt, rt->fold_spans, rt->fold_used);
    return SAR_OK;
}

}  // namespace

// The warm-up of the first `m` jobs of a coming launch, on the side stream, into the second set of warm-up buffers: behind
// the iterate kernel in flight (its accumulate / fold / colorize are what this runs under) or, with nothing in flight, at
// once. `starts` is [m][3] in device memory, or (soa) the kernel's x[m] y[m] z[m] block. Leaves rt->pf describing it.
namespace {
int warmup_ahead(sar_runtime* rt, const sar::MapParams& p, const double* starts, bool soa, uint32_t m, uint64_t iters,
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
            if (q) dev_free(rt, q);
        rt->d_warm_alt = nullptr; rt->d_joblist_alt = nullptr;
        rt->warm_alt_cap = 0;
        HIP_TRY(dev_alloc(rt, &rt->d_warm_alt, static_cast<size_t>(m) * 3 * sizeof(double)));
        HIP_TRY(dev_alloc(rt, &rt->d_joblist_alt, static_cast<size_t>(m) * sizeof(uint32_t)));
        rt->warm_alt_cap = m;
    }
    // the converted start points of an announced call: NOT one of the two sets that swap (its capacity is its own)
    if (!soa && m > rt->starts_alt_cap) {
        HIP_TRY(hipStreamSynchronize(rt->side));
        if (rt->d_starts_alt) dev_free(rt, rt->d_starts_alt);
        rt->d_starts_alt = nullptr;
        rt->starts_alt_cap = 0;
        HIP_TRY(dev_alloc(rt, &rt->d_starts_alt, static_cast<size_t>(m) * 3 * sizeof(double)));
        rt->starts_alt_cap = m;
    }
    if (!rt->d_active_alt) HIP_TRY(dev_alloc(rt, &rt->d_active_alt, 4 * sizeof(uint32_t)));
    if (!rt->d_hint_range_alt) HIP_TRY(dev_alloc(rt, &rt->d_hint_range_alt, 2 * sizeof(uint32_t)));
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
    launch_warmup(warm_args(pf.p, soa ? starts : rt->d_starts_alt, m, iters, rt->d_warm_alt, rt->d_joblist_alt, rt->d_active_alt, rt->W,
                            measure_range ? rt->d_hint_range_alt : nullptr), rt->side);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(rt->pf_done, rt->side));
    pf.valid = true;
    return SAR_OK;
}

}  // namespace

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
    SAR_TRY(ensure_scratch(rt, pl.binned ? pl.splits : 1u));
    SAR_TRY(stage_starts(rt, pl, n_jobs, starts, starts_on_device));
    SAR_TRY(grow_device(rt, rt->d_ckpt, rt->ckpt_cap, static_cast<size_t>(pl.n_ckpt) * 3 * pl.chunk_jobs));
    if (pl.binned) SAR_TRY(ensure_binned_buffers(rt, pl, hints_shared(rt, rt, pl, false) ? 1u : 8u));

    IterArgs ia;
    FoldArgs fa;
    fill_iter_fold_args(cfg, rt, pl, ia, fa);

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
                SAR_TRY(launch_binned_chunk(rt, pl, ia, fa, first, carry, announced || chunk_ahead));
                chunk_ahead = false;
                // a call of several launch chunks (configs[3] on one GPU: three) announces its own next chunk: that chunk's
                // warm-up runs under this chunk's accumulate and fold (its start points are staged already)
                const uint64_t next = off + pl.chunk_jobs;
                if (n_seg == 1 && next < n_jobs && rt->chunk_ahead != 2) {
                    const uint32_t m_next = static_cast<uint32_t>((n_jobs - next < pl.chunk_jobs) ? n_jobs - next : pl.chunk_jobs);
                    SAR_TRY(warmup_ahead(rt, ia.p, rt->d_starts + next * 3, true, m_next, it, false));
                    chunk_ahead = true;
                }
            } else {
                ia.resume = first ? 0u : 1u;
                ia.state_out = carry ? rt->d_starts + off * 3 : nullptr;
                span_begin(rt, rt->iter_spans, rt->iter_used);
                launch_iterate(ia, pl.block, rt->stream);
                span_end(rt, rt->iter_spans, rt->iter_used);
                ++rt->last_chunks;
                std::snprintf(rt->last_launch, sizeof(rt->last_launch), "k_iterate (one global atomic per visit)");
                span_begin(rt, rt->fold_spans, rt->fold_used);
                launch_fold_resolve(fa, rt->stream);
                span_end(rt, rt->fold_spans, rt->fold_used);
            }
        }
    }
    HIP_TRY(hipGetLastError());
    rt->last_iterations = static_cast<uint64_t>(n_jobs) * iters;
    rt->starts_consumed_recorded = false;  // (what read d_starts here is ordered by this stream, not by the batch path's event)
    return SAR_OK;
}

extern "C" {

int sar_render(const sar_config* cfg, sar_runtime* rt) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    double p0[3];
    rt->rng.start_point(p0);  // :748
    return render_chunked(cfg, rt, 1, cfg->iterations, p0);
} catch (...) { return sar::abi_caught(); }

int sar_render_jobs(const sar_config* cfg, sar_runtime* rt, const double* starts_xyz_host) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_render_job_range(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                         const double* starts_xyz_host) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (n_jobs && !starts_xyz_host) { set_error("starts_xyz_host is NULL"); return SAR_ERR_INVALID; }
    return render_chunked(cfg, rt, n_jobs, iters_per_job, starts_xyz_host);
} catch (...) { return sar::abi_caught(); }

int sar_render_job_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                                const double* starts_xyz_dev) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (n_jobs && !starts_xyz_dev) { set_error("starts_xyz_dev is NULL"); return SAR_ERR_INVALID; }
    return render_chunked(cfg, rt, n_jobs, iters_per_job, starts_xyz_dev, true);
} catch (...) { return sar::abi_caught(); }

int sar_runtime_describe_last_launch(const sar_runtime* rt, char* out, size_t cap) try {
    if (!rt || !out || cap == 0) return SAR_ERR_INVALID;
    std::snprintf(out, cap, "%s | chunks=%u warmup_ahead=%u", rt->last_launch[0] ? rt->last_launch : "nothing launched", rt->last_chunks,
                  rt->prefetch_used);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_prefetch_device(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                                const double* starts_xyz_dev) try {
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
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
