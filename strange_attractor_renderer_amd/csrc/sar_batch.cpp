// sar_batch.cpp — F frames of one shape through ONE launch of every kernel of the binned path (sar_render_jobs_batch).
//
// The reference's `sequence` loop (src/bin/main.rs:493-517) renders frame after frame, each a reset (src/lib.rs:950-951) and a
// render_parallel of fresh jobs. A frame of BASELINE configs[4] (65 536 jobs, 38 % of them lost in the warm-up) fills a third
// of an MI355X; frames are independent, so F of them — each with its own Runtime, view angle and start points — share the
// chip: workgroup -> frame (blockIdx.z) -> that frame's argument block (BatchFrame) in a table in device memory. Every
// frame's buffers are the ones its Runtime already owns, every kernel body is the single-frame body, so the result is what F
// sar_render_jobs calls leave, bit for bit. Host logic only.
#include <cstring>
#include <vector>

#include "sar_plan.hpp"

using namespace sar;

namespace {

int sequential(uint32_t n_frames, const sar_config* const* cfgs, sar_runtime* const* rts, const double* const* starts) {
    for (uint32_t i = 0; i < n_frames; ++i) SAR_TRY(sar_render_jobs(cfgs[i], rts[i], starts ? starts[i] : nullptr));
    return SAR_OK;
}

// Frames [0, F) as one batched launch on rts[0]'s stream; `batched` says whether that form applied (otherwise nothing was done).
int launch_batch(uint32_t F, const sar_config* const* cfgs, sar_runtime* const* rts, const double* const* starts, bool& batched) {
    batched = false;
    sar_runtime* lead = rts[0];
    const uint32_t n_jobs = cfgs[0]->jobs_total;
    const uint64_t iters = n_jobs ? cfgs[0]->iterations / n_jobs : 0;  // :1058
    if (n_jobs == 0 || iters == 0) return SAR_OK;
    for (uint32_t i = 0; i < F; ++i) {
        const sar_runtime* rt = rts[i];
        if (rt->device != lead->device || rt->W != lead->W || rt->H != lead->H) return SAR_OK;
        if (cfgs[i]->jobs_total != n_jobs || cfgs[i]->iterations / n_jobs != iters || cfgs[i]->scale != cfgs[0]->scale) return SAR_OK;
        for (uint32_t j = 0; j < i; ++j)
            if (rts[j] == rt) return SAR_OK;
    }
    if (iters > (lead->max_ordinals ? lead->max_ordinals : kMaxChunkOrdinals)) return SAR_OK;  // jobs of several segments
    HIP_TRY(hipSetDevice(lead->device));
    LaunchPlan pl;
    SAR_TRY(plan_launch(cfgs[0], lead, n_jobs, iters, pl, F));
    // the batched kernels: wave pairs, bins with 32-bit counters, every frame ONE launch chunk
    if (!pl.binned || !pl.split || pl.geo.shift > 15u || pl.chunk_jobs < n_jobs) return SAR_OK;

    // the members' own streams meet the leader's: what they hold (a reset, a read-back) comes first, what follows waits for the batch
    hipStream_t own[kMaxBatchFrames];
    for (uint32_t i = 0; i < F; ++i) {
        sar_runtime* rt = rts[i];
        own[i] = rt->stream;
        if (rt->stream != lead->stream) {
            if (!rt->batch_join) HIP_TRY(hipEventCreateWithFlags(&rt->batch_join, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(rt->batch_join, rt->stream));
            HIP_TRY(hipStreamWaitEvent(lead->stream, rt->batch_join, 0));
        }
    }
    struct Restore {  // the helpers below enqueue on rt->stream: the leader's, for the length of this call
        uint32_t F; sar_runtime* const* rts; hipStream_t* own;
        ~Restore() { for (uint32_t i = 0; i < F; ++i) rts[i]->stream = own[i]; }
    } restore{F, rts, own};
    for (uint32_t i = 0; i < F; ++i) rts[i]->stream = lead->stream;

    if (!lead->d_batch) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&lead->d_batch), sizeof(BatchFrame) * kMaxBatchFrames));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&lead->h_batch), sizeof(BatchFrame) * kMaxBatchFrames * kBatchRing, hipHostMallocDefault));
        for (hipEvent_t& e : lead->batch_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const uint32_t ring = static_cast<uint32_t>(lead->batch_next % kBatchRing);
    if (lead->batch_next >= kBatchRing) HIP_TRY(hipEventSynchronize(lead->batch_copied[ring]));  // (eight batches back: long done)
    BatchFrame* table = lead->h_batch + static_cast<size_t>(ring) * kMaxBatchFrames;

    std::vector<double> drawn;
    bool share = false;
    for (uint32_t i = 0; i < F; ++i) {
        sar_runtime* rt = rts[i];
        if (!rt->timing_accumulate) {
            rt->last_iterations = 0;
            rt->iter_used = rt->fold_used = rt->warm_used = 0;
        }
        rt->pf.valid = false;  // an announcement was for another call
        rt->last_chunks = 0;
        const double* st = starts ? starts[i] : nullptr;
        if (!st) {  // as sar_render_jobs: from the runtime's own stream (:748)
            drawn.resize(static_cast<size_t>(n_jobs) * 3);
            for (uint32_t k = 0; k < n_jobs; ++k) rt->rng.start_point(&drawn[3 * static_cast<size_t>(k)]);
            st = drawn.data();
        }
        SAR_TRY(ensure_scratch(rt, pl.splits));
        SAR_TRY(stage_starts(rt, pl, n_jobs, st, false));
        SAR_TRY(grow_device(rt->d_ckpt, rt->ckpt_cap, static_cast<size_t>(pl.n_ckpt) * 3 * pl.chunk_jobs));
        SAR_TRY(ensure_binned_buffers(rt, pl));

        BatchFrame& f = table[i];
        std::memset(&f, 0, sizeof(f));
        IterArgs ia;
        fill_iter_fold_args(cfgs[i], rt, pl, ia, f.fold);
        ia.n_jobs = n_jobs;
        ia.iters = iters;
        ia.starts = rt->d_starts;
        f.fold.n_jobs = n_jobs;
        f.fold.iters = iters;
        f.fold.seg_any = rt->d_seg_any;
        fill_bin_iter_args(rt, lead, pl, ia, f.it, &share);
        // narrow hints: the first warm-up after the hints were cleared also measures the depth range they quantise
        uint32_t* measure = nullptr;
        if (pl.hint_bytes == 2 && !rt->hint_range_set) {
            measure = rt->d_hint_range;
            rt->hint_range_set = true;
            f.clear_hint_range = 1u;
        }
        f.warm = warm_args(ia.p, ia.starts, n_jobs, iters, rt->d_warm, rt->d_joblist, rt->d_active, ia.width, measure);
        f.it.warm = rt->d_warm;
        f.it.joblist = rt->d_joblist;
        f.it.active = rt->d_active;
        f.it.warm_nan = reinterpret_cast<const unsigned long long*>(rt->d_active + 2);
        fill_bin_acc_args(rt, pl, f.it, f.acc);
        f.seg_any = rt->d_seg_any;
        f.seg_words = rt->npix / 2048u + 1u;
    }

    const BatchFrame* dtab = lead->d_batch;
    HIP_TRY(hipMemcpyAsync(lead->d_batch, table, sizeof(BatchFrame) * F, hipMemcpyHostToDevice, lead->stream));
    HIP_TRY(hipEventRecord(lead->batch_copied[ring], lead->stream));
    ++lead->batch_next;

    span_begin(lead, lead->warm_spans, lead->warm_used);
    launch_batch_clear(dtab, F, lead->npix / 2048u + 1u, lead->stream);
    launch_warmup_batch(dtab, F, n_jobs, lead->stream);
    span_end(lead, lead->warm_spans, lead->warm_used);
    span_begin(lead, lead->iter_spans, lead->iter_used);
    if (launch_iterate_split_batch(dtab, F, table[0].it.n_waves, pl.geo.bins, pl.R, pl.hint_bytes, lead->batch_xcd != 1u, lead->stream) != 0) {
        set_error("no batched iterate kernel for chunk_records %u / %u-byte hints", pl.R, pl.hint_bytes);
        return SAR_ERR_INVALID;
    }
    HIP_TRY(hipGetLastError());
    span_end(lead, lead->iter_spans, lead->iter_used);
    span_begin(lead, lead->fold_spans, lead->fold_used);
    if (launch_bin_accumulate_batch(dtab, F, pl.geo.bins, pl.splits, pl.geo.shift, lead->acc_threads, pl.R, pl.acc_lists, lead->stream) != 0) {
        set_error("no batched accumulate kernel for chunk_records %u / %u lists per lane group", pl.R, pl.acc_lists);
        return SAR_ERR_INVALID;
    }
    launch_fold_resolve_batch(dtab, F, lead->npix, lead->stream);
    HIP_TRY(hipGetLastError());
    span_end(lead, lead->fold_spans, lead->fold_used);

    // survivor statistics for the next plan (never waited for), from the leader's frame
    if (!lead->active_pending &&
        hipMemcpyAsync(lead->h_active, lead->d_active, sizeof(uint32_t), hipMemcpyDeviceToHost, lead->stream) == hipSuccess &&
        hipEventRecord(lead->active_copied, lead->stream) == hipSuccess) {
        lead->active_pending = true;
        lead->active_jobs_launched = n_jobs;
    }
    ++lead->batches_launched;
    for (uint32_t i = 0; i < F; ++i) {
        sar_runtime* rt = rts[i];
        rt->last_chunks = 1;
        rt->last_iterations += static_cast<uint64_t>(n_jobs) * iters;
        describe_launch(rt, pl, share, F);
        if (i) rt->survivor_fraction = lead->survivor_fraction;
    }
    bool joined = false;
    for (uint32_t i = 0; i < F; ++i) {
        if (own[i] == lead->stream) continue;
        if (!joined) {
            if (!lead->batch_join) HIP_TRY(hipEventCreateWithFlags(&lead->batch_join, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(lead->batch_join, lead->stream));
            joined = true;
        }
        HIP_TRY(hipStreamWaitEvent(own[i], lead->batch_join, 0));
    }
    batched = true;
    return SAR_OK;
}

}  // namespace

extern "C" {

int sar_render_jobs_batch(uint32_t n_frames, const sar_config* const* cfgs, sar_runtime* const* rts, const double* const* starts_xyz_host) {
    if (n_frames == 0) return SAR_OK;
    if (!cfgs || !rts) { set_error("sar_render_jobs_batch: NULL argument"); return SAR_ERR_INVALID; }
    for (uint32_t i = 0; i < n_frames; ++i) {
        if (!cfgs[i] || !rts[i]) { set_error("sar_render_jobs_batch: frame %u is NULL", i); return SAR_ERR_INVALID; }
        SAR_TRY(check_cfg_matches(cfgs[i], rts[i]));
        if (cfgs[i]->jobs_total == 0) { set_error("jobs_total is 0"); return SAR_ERR_INVALID; }
    }
    for (uint32_t first = 0; first < n_frames; first += kMaxBatchFrames) {
        const uint32_t F = n_frames - first < kMaxBatchFrames ? n_frames - first : kMaxBatchFrames;
        const double* const* st = starts_xyz_host ? starts_xyz_host + first : nullptr;
        bool batched = false;
        if (F > 1) SAR_TRY(launch_batch(F, cfgs + first, rts + first, st, batched));
        if (!batched) SAR_TRY(sequential(F, cfgs + first, rts + first, st));
    }
    return SAR_OK;
}

int sar_runtime_batch_frames(const sar_config* cfg, sar_runtime* rt, uint32_t* out_frames) {
    if (!out_frames) return SAR_ERR_INVALID;
    *out_frames = 1;
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (cfg->jobs_total == 0) return SAR_OK;
    if (rt->active_pending && hipEventQuery(rt->active_copied) == hipSuccess) {
        rt->active_pending = false;
        if (rt->active_jobs_launched) rt->survivor_fraction = static_cast<double>(*rt->h_active) / rt->active_jobs_launched;
    }
    // the chip holds eight wave pairs per CU; a frame occupies one per 64 surviving jobs
    const uint64_t cus = rt->sm_count ? rt->sm_count : 256u;
    const double live = cfg->jobs_total * (rt->survivor_fraction > 0.05 ? rt->survivor_fraction : 0.05);
    const uint64_t pairs = static_cast<uint64_t>(live / 64.0 + 0.999);
    uint64_t f = pairs ? (8u * cus) / pairs : 1u;
    if (f < 1) f = 1;
    if (f > kMaxBatchFrames) f = kMaxBatchFrames;
    *out_frames = static_cast<uint32_t>(f);
    return SAR_OK;
}

}  // extern "C"
