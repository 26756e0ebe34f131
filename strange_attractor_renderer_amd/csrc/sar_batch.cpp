// sar_batch.cpp — F frames of one shape through ONE launch of every kernel of the binned path (sar_render_jobs_batch).
//
// The reference's `sequence` loop (src/bin/main.rs:493-517) renders frame after frame, each a reset (src/lib.rs:950-951) and a
// render_parallel of fresh jobs. A frame of BASELINE configs[4] (65 536 jobs, 38 % of them lost in the warm-up) fills a third
// of an MI355X; frames are independent, so F of them — each with its own Runtime, view angle and start points — share the
// chip: workgroup -> frame (blockIdx.z) -> that frame's argument block (BatchFrame) in a table in device memory. Every
// frame's buffers are the ones its Runtime already owns, every kernel body is the single-frame body, so the result is what F
// sar_render_jobs calls leave, bit for bit. Host logic only.
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "sar_plan.hpp"

using namespace sar;

namespace {

// The iterate kernels of the batches of one device run ONE AFTER THE OTHER, whatever streams they are on: a batched launch
// fills every CU's LDS, so two of them never share a CU anyway — but left to themselves two lanes of batches fall into step
// (both iterate kernels interleaved workgroup by workgroup, then both tails at once: 0.75 ms per frame of configs[4]) as
// often as out of step (one lane's accumulate / fold / colorize / reset / warm-up under the other's iterate kernel: 0.6).
// One event per device, waited for before a batch's iterate kernel and recorded behind it, keeps them out of step.
struct IterateChain {
    std::mutex mu;
    hipEvent_t done = nullptr;
    bool recorded = false;
} g_chain[64];

int sequential(uint32_t n_frames, const sar_config* const* cfgs, sar_runtime* const* rts, const double* const* starts) {
    for (uint32_t i = 0; i < n_frames; ++i) SAR_TRY(sar_render_jobs(cfgs[i], rts[i], starts ? starts[i] : nullptr));
    return SAR_OK;
}

// Whether F frames of (cfg's job split) on runtimes like `lead` can share ONE set of launches — and the plan they would share: the
// batched kernels are wave pairs over bins with 32-bit counters, every frame one launch chunk of one segment. What
// sar_render_jobs_batch launches and what sar_runtime_batch_frames answers go through this one test.
int batch_plan(const sar_config* cfg, sar_runtime* lead, uint32_t F, LaunchPlan& pl, bool& applies) {
    applies = false;
    const uint32_t n_jobs = cfg->jobs_total;
    const uint64_t iters = n_jobs ? cfg->iterations / n_jobs : 0;  // :1058
    if (n_jobs == 0 || iters == 0 || F < 2) return SAR_OK;
    if (iters > (lead->max_ordinals ? lead->max_ordinals : kMaxChunkOrdinals)) return SAR_OK;  // jobs of several segments
    SAR_TRY(plan_launch(cfg, lead, n_jobs, iters, pl, F));
    applies = pl.binned && pl.split && pl.geo.shift <= 15u && pl.chunk_jobs >= n_jobs;
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
        // the frames' hints are laid out by the LEADER's options: a member that may hold hints written in another layout (its own
        // hint_tile option) renders on its own — a hint read in the wrong layout belongs to another pixel and could reject a winner
        if (rt->hint_tile != lead->hint_tile) return SAR_OK;
        for (uint32_t j = 0; j < i; ++j)
            if (rts[j] == rt) return SAR_OK;
    }
    HIP_TRY(hipSetDevice(lead->device));
    LaunchPlan pl;
    bool applies = false;
    SAR_TRY(batch_plan(cfgs[0], lead, F, pl, applies));
    if (!applies) return SAR_OK;

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
        uint32_t F; sar_runtime* const* rts; hipStream_t* own; hipStream_t lead_stream; bool joined;
        ~Restore() {
            for (uint32_t i = 0; i < F; ++i) rts[i]->stream = own[i];
            // an error return after work was enqueued: the members' own streams never waited for the leader's — what is in flight
            // there (kernels on the members' buffers) ends before anybody resets, colorizes or frees them
            if (!joined) hipStreamSynchronize(lead_stream);
        }
    } restore{F, rts, own, lead->stream, false};
    for (uint32_t i = 0; i < F; ++i) rts[i]->stream = lead->stream;
    // A preset that loses jobs in the warm-up (solar-sail: 38 %, all within the first ~100 iterations) warms up in two phases:
    // kFirstPhase iterations, the survivors packed, the rest on full waves. (A launch of one frame runs one wave per SIMD and gains
    // nothing from fewer waves; a batch keeps the vector units busy, and 38 % fewer waves are 33 % less warm-up time.)
    constexpr uint32_t kFirstPhase = 160;
    const bool two_phase = lead->batch_warm == 2u || (lead->batch_warm == 0u && lead->survivor_fraction < 0.9);
    // start points: 0 = read in place by the (one-phase) warm-up kernel — a thousand iterations hide 1.5 MB per frame over PCIe —
    // or, before a first phase too short for that, fetched by a kernel of a few waves; 1 / 2 = copied on the upload / launch
    // stream; 3 = in place whatever the warm-up
    const uint32_t starts_mode = lead->batch_starts;
    const bool fetch = starts_mode == 0u && two_phase;
    const bool in_place = (starts_mode == 0u && !two_phase) || starts_mode == 3u;
    if (starts_mode == 1u) {
        if (!lead->upload_stream) HIP_TRY(hipStreamCreateWithFlags(&lead->upload_stream, hipStreamNonBlocking));
        for (uint32_t i = 0; i < F; ++i) {
            sar_runtime* rt = rts[i];
            if (!rt->starts_consumed) HIP_TRY(hipEventCreateWithFlags(&rt->starts_consumed, hipEventDisableTiming));
            if (!rt->starts_consumed_recorded) {  // its last render was not a batch: whatever its stream holds comes first
                HIP_TRY(hipEventRecord(rt->starts_consumed, own[i]));
                rt->starts_consumed_recorded = true;
            }
        }
    }

    if (!lead->d_batch) {
        HIP_TRY(dev_alloc(lead, &lead->d_batch, sizeof(BatchFrame) * kMaxBatchFrames));
        HIP_TRY(host_alloc(lead, &lead->h_batch, sizeof(BatchFrame) * kMaxBatchFrames * kBatchRing));
        for (hipEvent_t& e : lead->batch_copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const uint32_t ring = static_cast<uint32_t>(lead->batch_next % kBatchRing);
    if (lead->batch_next >= kBatchRing) HIP_TRY(hipEventSynchronize(lead->batch_copied[ring]));  // (eight batches back: long done)
    BatchFrame* table = lead->h_batch + static_cast<size_t>(ring) * kMaxBatchFrames;

    // how the iterate kernel deals the frames to the XCDs; a frame on one or two XCDs of its own keeps ONE array of depth hints
    // (a frame on every XCD keeps one per XCD, as a single-frame launch does)
    const uint32_t n_waves = static_cast<uint32_t>(((n_jobs + pl.block - 1) / pl.block) * (pl.block / 64u));
    const uint32_t xcd_map = lead->batch_xcd == 1u ? 0u : batch_xcd_map(F, n_waves);
    const bool one_hint_array = xcd_map == 2u || xcd_map == 3u || (xcd_map == 1u && F >= 4u);
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
        SAR_TRY(stage_starts(rt, pl, n_jobs, st, false, starts_mode == 1u ? lead->upload_stream : nullptr, in_place || fetch));
        SAR_TRY(grow_device(rt, rt->d_ckpt, rt->ckpt_cap, static_cast<size_t>(pl.n_ckpt) * 3 * pl.chunk_jobs));
        SAR_TRY(ensure_binned_buffers(rt, pl, hints_shared(rt, lead, pl, one_hint_array) ? 1u : 8u));

        BatchFrame& f = table[i];
        std::memset(&f, 0, sizeof(f));
        IterArgs ia;
        fill_iter_fold_args(cfgs[i], rt, pl, ia, f.fold);
        ia.n_jobs = n_jobs;
        ia.iters = iters;
        ia.starts = in_place ? rt->h_starts : rt->d_starts;  // (page-locked host memory is device-visible at its own address)
        f.fold.n_jobs = n_jobs;
        f.fold.iters = iters;
        f.fold.seg_any = rt->d_seg_any;
        fill_bin_iter_args(rt, lead, pl, ia, f.it, &share, one_hint_array);
        // narrow hints: the first warm-up after the hints were cleared also measures the depth range they quantise
        uint32_t* measure = nullptr;
        if (pl.hint_bytes == 2 && !rt->hint_range_set) {
            measure = rt->d_hint_range;
            rt->hint_range_set = true;
            f.clear_hint_range = 1u;
        }
        f.warm = warm_args(ia.p, ia.starts, n_jobs, iters, rt->d_warm, rt->d_joblist, rt->d_active, ia.width, measure);
        if (two_phase) {
            // (an announced warm-up nobody consumed may still write that second set on the runtime's side stream: behind it)
            if (rt->pf_done) HIP_TRY(hipStreamWaitEvent(lead->stream, rt->pf_done, 0));
            if (n_jobs > rt->warm_alt_cap) {  // the second set of warm-up buffers (an announced call's otherwise): the first phase's output
                if (rt->side) HIP_TRY(hipStreamSynchronize(rt->side));
                HIP_TRY(hipStreamSynchronize(lead->stream));
                if (rt->d_warm_alt) dev_free(rt, rt->d_warm_alt);
                if (rt->d_joblist_alt) dev_free(rt, rt->d_joblist_alt);
                rt->d_warm_alt = nullptr; rt->d_joblist_alt = nullptr;
                rt->warm_alt_cap = 0;
                HIP_TRY(dev_alloc(rt, &rt->d_warm_alt, static_cast<size_t>(n_jobs) * 3 * sizeof(double)));
                HIP_TRY(dev_alloc(rt, &rt->d_joblist_alt, static_cast<size_t>(n_jobs) * sizeof(uint32_t)));
                rt->warm_alt_cap = n_jobs;
            }
            if (!rt->d_active_alt) HIP_TRY(dev_alloc(rt, &rt->d_active_alt, 4 * sizeof(uint32_t)));
            f.warm_first = warm_args(ia.p, ia.starts, n_jobs, iters, rt->d_warm_alt, rt->d_joblist_alt, rt->d_active_alt, ia.width, nullptr);
            f.warm_first.n_iter = kFirstPhase;
            f.warm_first.nan_count = f.warm.nan_count;  // the jobs it drops count where the iterate kernel looks
            f.warm.starts = rt->d_warm_alt;
            f.warm.n_iter = 1000u - kFirstPhase;
            f.warm.in_active = rt->d_active_alt;
            f.warm.in_joblist = rt->d_joblist_alt;
        }
        f.it.warm = rt->d_warm;
        f.it.joblist = rt->d_joblist;
        f.it.active = rt->d_active;
        f.it.warm_nan = reinterpret_cast<const unsigned long long*>(rt->d_active + 2);
        fill_bin_acc_args(rt, pl, f.it, f.acc);
        f.seg_any = rt->d_seg_any;
        f.seg_words = rt->npix / 2048u + 1u;
        f.starts_host = rt->h_starts;
        f.starts_dev = rt->d_starts;
        f.n_start_quads = static_cast<uint32_t>((static_cast<size_t>(n_jobs) * 3 * sizeof(double) + 15u) / 16u);
    }

    const BatchFrame* dtab = lead->d_batch;
    HIP_TRY(hipMemcpyAsync(lead->d_batch, table, sizeof(BatchFrame) * F, hipMemcpyHostToDevice, lead->stream));
    HIP_TRY(hipEventRecord(lead->batch_copied[ring], lead->stream));
    ++lead->batch_next;

    if (fetch) {
        launch_batch_fetch(dtab, F, lead->stream);
        for (uint32_t i = 0; i < F; ++i) {  // the page-locked buffers may be written again
            HIP_TRY(hipEventRecord(rts[i]->starts_copied, lead->stream));
            rts[i]->starts_pending = true;
        }
    }
    span_begin(lead, lead->warm_spans, lead->warm_used);
    launch_batch_clear(dtab, F, lead->npix / 2048u + 1u, lead->stream);
    if (two_phase) launch_warmup_batch(dtab, F, n_jobs, true, lead->stream);
    if (two_phase && in_place)  // the start points have been read: the page-locked buffers may be written again
        for (uint32_t i = 0; i < F; ++i) {
            HIP_TRY(hipEventRecord(rts[i]->starts_copied, lead->stream));
            rts[i]->starts_pending = true;
        }
    launch_warmup_batch(dtab, F, n_jobs, false, lead->stream);
    span_end(lead, lead->warm_spans, lead->warm_used);
    for (uint32_t i = 0; i < F; ++i) {  // the start points have been read: their buffer may be written again
        if (starts_mode == 1u) HIP_TRY(hipEventRecord(rts[i]->starts_consumed, lead->stream));
        if (in_place && !two_phase) {
            HIP_TRY(hipEventRecord(rts[i]->starts_copied, lead->stream));
            rts[i]->starts_pending = true;
        }
    }
    span_begin(lead, lead->iter_spans, lead->iter_used);
    IterateChain* chain = (lead->batch_chain != 1u && lead->device >= 0 && lead->device < 64) ? &g_chain[lead->device] : nullptr;
    {
        std::unique_lock<std::mutex> lock;
        if (chain) {
            lock = std::unique_lock<std::mutex>(chain->mu);
            if (!chain->done) HIP_TRY(hipEventCreateWithFlags(&chain->done, hipEventDisableTiming));
            if (chain->recorded) HIP_TRY(hipStreamWaitEvent(lead->stream, chain->done, 0));
        }
        if (launch_iterate_split_batch(dtab, F, n_waves, pl.geo.bins, pl.R, pl.hint_bytes, xcd_map, lead->stream) != 0) {
            set_error("no batched iterate kernel for chunk_records %u / %u-byte hints", pl.R, pl.hint_bytes);
            return SAR_ERR_INVALID;
        }
        HIP_TRY(hipGetLastError());
        if (chain) {
            HIP_TRY(hipEventRecord(chain->done, lead->stream));
            chain->recorded = true;
        }
    }
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
        describe_launch(rt, pl, share, F, xcd_map);
        if (i) { rt->survivor_fraction = lead->survivor_fraction; rt->survivors_known = lead->survivors_known; }
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
    restore.joined = true;
    batched = true;
    return SAR_OK;
}

}  // namespace

extern "C" {

int sar_render_jobs_batch(uint32_t n_frames, const sar_config* const* cfgs, sar_runtime* const* rts, const double* const* starts_xyz_host) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_runtime_batch_frames(const sar_config* cfg, sar_runtime* rt, uint32_t* out_frames) try {
    if (!out_frames) return SAR_ERR_INVALID;
    *out_frames = 1;
    sar_runtime probe;  // rt == NULL: the answer for a runtime yet to be made (default options, no launch behind it)
    if (!rt) {
        SAR_TRY(sar_config_validate(cfg));
        probe.W = cfg->width; probe.H = cfg->height;
        probe.npix = static_cast<uint32_t>(static_cast<uint64_t>(cfg->width) * cfg->height);
        rt = &probe;
    }
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (cfg->jobs_total == 0) return SAR_OK;
    if (rt->active_pending && hipEventQuery(rt->active_copied) == hipSuccess) {
        rt->active_pending = false;
        if (rt->active_jobs_launched) {
            rt->survivor_fraction = static_cast<double>(*rt->h_active) / rt->active_jobs_launched;
            rt->survivors_known = true;
        }
    }
    // A batch of 8k frames gives every XCD k frames, one after the other (k_iterate_split_batch); an XCD holds eight wave pairs per
    // CU, a frame occupies one per 64 jobs that survive the warm-up, and all pairs run equally long: the XCD works in ROUNDS. The
    // smallest batch whose last round is at least 95 % full — or the fullest (configs[4]: 635 pairs per frame on 256 slots are
    // 2.48 rounds for one frame per XCD, 4.96 for two).
    const uint64_t cus = rt->sm_count ? rt->sm_count : 256u;
    const double slots = static_cast<double>(cus);  // eight pairs per CU on an eighth of the CUs
    const double live = cfg->jobs_total * (rt->survivor_fraction > 0.05 ? rt->survivor_fraction : 0.05);
    const double pairs = std::ceil(live / 64.0);
    uint32_t best = 8;
    double best_fill = 0.0;
    // (no launch of this runtime has reported its survivors yet: two frames per XCD — a full last round matters less the more
    // rounds there are, and a preset that loses no job fills its rounds at 8 and at 16 alike)
    for (uint32_t k = rt->survivors_known ? 1u : 2u; 8u * k <= kMaxBatchFrames; ++k) {
        const double rounds = k * pairs / slots;
        const double fill = rounds / std::ceil(rounds);
        if (fill > best_fill + 1e-9) { best_fill = fill; best = 8u * k; }
        if (fill >= 0.95) break;
    }
    // ... if frames of this shape share launches at all (the same test sar_render_jobs_batch makes: images beyond 4 Mpx, jobs of
    // several launch chunks or segments, the one-atomic-per-visit path render frame after frame — a caller that builds `best`
    // runtimes per lane for those would hold 8..32 times the memory for nothing)
    LaunchPlan pl;
    bool applies = false;
    SAR_TRY(batch_plan(cfg, rt, best, pl, applies));
    *out_frames = applies ? best : 1u;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
