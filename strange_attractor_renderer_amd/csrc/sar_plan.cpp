// sar_plan.cpp — planning of a render call (host logic only): bin geometry, chunk size, iterate-kernel form, launch chunks.
#include "sar_plan.hpp"

using namespace sar;

BinGeometry sar::bin_geometry(uint32_t npix, uint32_t want_block, uint32_t want_shift, uint32_t want_splits, uint32_t records,
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
    const uint32_t waves_fit = (160u * 1024u) / lean_wave_lds_bytes(g.bins, records);
    uint32_t block = want_block;
    if (block > waves_fit * 64u) block = waves_fit * 64u;
    if (block == 0) return g;
    g.block = block;
    g.splits = 0;  // chosen per launch: about one (bin, wave) list per thread of a 1024-thread block
    if (want_splits) g.splits = want_splits;
    g.ok = true;
    return g;
}


namespace {

// Records per chunk, bin geometry and the form of the iterate kernel. The job count is scaled by the share of jobs that
// survived the previous launch's warm-up (solar-sail loses 38 % of its start points there).
uint32_t choose_chunk_records(sar_runtime* rt, uint64_t n_jobs, bool batched, uint32_t& shift, uint32_t& interleave, bool& split, uint64_t& resident_jobs) {
    shift = rt->bin_shift;
    interleave = rt->bin_interleave;
    if (rt->active_pending && hipEventQuery(rt->active_copied) == hipSuccess) {
        rt->active_pending = false;
        if (rt->active_jobs_launched) {
            rt->survivor_fraction = static_cast<double>(*rt->h_active) / rt->active_jobs_launched;
            rt->survivors_known = true;
        }
    }
    const uint64_t cus = rt->sm_count ? rt->sm_count : 256u;
    const uint64_t busy = static_cast<uint64_t>(n_jobs * rt->survivor_fraction + 0.5);
    uint64_t want = (busy + 64u * cus - 1) / (64u * cus);  // waves per CU if all surviving jobs were resident
    want = ((want + 3) / 4) * 4;  // workgroups are four waves: residency comes in steps of four waves per CU
    want = want < 8 ? 8 : (want > 12 ? 12 : want);
    // k_iterate_split (producer / consumer wave pairs) keeps 8 staging sets per CU busy with 16 waves: for launches whose
    // jobs are all resident at once (512 per CU). A launch of several rounds of workgroups desynchronises by itself —
    // workgroups of different rounds are in different phases — and the whole kernel is the faster one there (configs[3] on
    // one GPU, 8 rounds: 81.7 against 89.9 ms). Three pairs per SIMD are slower everywhere (profiles/dead_ends.md).
    // (a batched launch exists as wave pairs only; its caller keeps the frames' survivors near what the chip holds)
    split = batched || rt->split_waves == 2 || (rt->split_waves == 0 && busy <= 512u * cus);
    if (split) want = 8;
    // jobs (dead ones included: they are launched and dropped by the warm-up) whose survivors the chip holds at once
    resident_jobs = static_cast<uint64_t>(64.0 * cus * want / (rt->survivor_fraction > 0.05 ? rt->survivor_fraction : 0.05));
    // Interleaved bins carry equal loads, so few LARGE bins cost the slot requests nothing (with bins of consecutive
    // pixels half the bins idle and the rest collide) and k_bin_accumulate's 128 KiB histograms (one workgroup per CU) get
    // equal work. 128 bins of 32768 pixels leave room for 128-byte chunks at two waves per SIMD — half the buffer swaps,
    // whole cache lines for k_bin_accumulate — or for 64-byte chunks at three. (2048^2, 1e9 iterations: 131072 jobs
    // 7.0 -> 6.x ms per frame)
    // Beyond 4 Mpx the same with bins of 65536 pixels (all a 16-bit record addresses; k_bin_accumulate counts such a bin
    // with packed 16-bit counters): 4096^2 in 256 bins keeps 64-byte chunks where 512 bins allowed 32-byte ones
    // (1.25e9 iterations there: 13.5 -> 12.3 ms; 2560^2 and 3840x2160 take 128-byte chunks on 128 such bins: -2..3 %).
    if (rt->bin_shift == 0 && rt->chunk_records == 0 && rt->bin_interleave != 1) {
        for (uint64_t need : {want, static_cast<uint64_t>(8)})  // three waves per SIMD if the launch has the jobs, else two
            for (uint32_t cand : {60u, 28u})                      // the larger chunk first, the smaller bin first
                for (uint32_t sh : {15u, 16u}) {
                    // interleaved whatever the power-of-two bin count costs: the staging is checked to fit right here
                    const BinGeometry big = bin_geometry(rt->npix, rt->block_threads, sh, rt->splits, 12u, 2u);
                    if (big.ok && big.interleaved && lean_wave_lds_bytes(big.bins, cand) * need <= 160u * 1024u) {
                        shift = sh;
                        interleave = 2u;
                        return cand;
                    }
                }
    }
    // every other shape (bin size or map fixed by an option, more than 4 Mpx of consecutive-pixel bins, 8192^2): the largest
    // of 28 / 20 / 12 records whose staging keeps the waves this launch can use — up to 3 per SIMD, at least 2
    if (rt->chunk_records) return rt->chunk_records;
    const BinGeometry probe = bin_geometry(rt->npix, rt->block_threads, rt->bin_shift, rt->splits, 12u, rt->bin_interleave);
    if (!probe.ok) return kDefaultChunkRecords;
    for (uint32_t need : {static_cast<uint32_t>(want), 8u})
        for (uint32_t cand : {28u, 20u, 12u})
            if (lean_wave_lds_bytes(probe.bins, cand) * need <= 160u * 1024u) return cand;
    return 12u;
}

}  // namespace

int sar::plan_launch(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters, LaunchPlan& pl, uint32_t batch_frames) {
    uint32_t shift = 0, interleave = 0;
    if (batch_frames == 0) batch_frames = 1;
    // (a batched launch holds batch_frames x n_jobs jobs at once: that decides between the wave-pair and the whole kernel)
    pl.R = choose_chunk_records(rt, static_cast<uint64_t>(n_jobs) * batch_frames, batch_frames > 1, shift, interleave, pl.split, pl.resident_jobs);
    pl.geo = bin_geometry(rt->npix, rt->block_threads, shift, rt->splits, pl.R, interleave);
    // which accumulate path: LDS-binned records (default) or one global atomic per visit
    pl.binned = (rt->bins_mode == 0 || rt->bins_mode == 3) && pl.geo.ok;
    if (rt->bins_mode == 3 && !pl.geo.ok) {
        set_error("the binned path needs width*height <= %u pixels", kMaxBins * kMaxBinPx);
        return SAR_ERR_RANGE;
    }
    pl.block = pl.binned ? pl.geo.block : rt->block_threads;
    // the wave-pair kernel exists for 64- and 128-byte chunks whose staging + hand-over fit eight pairs per CU
    pl.split = pl.split && pl.binned && (pl.R == 60u || pl.R == 28u) && (lean_wave_lds_bytes(pl.geo.bins, pl.R) + 1024u) * 8u <= 160u * 1024u;
    // depth hints: the sortable f32 itself (3x fewer stage-2 waits, -7 % at 2048^2) while the hints of the pixels the
    // attractor touches stay near an XCD's 4 MiB L2, 16-bit fixed point beyond. The view maps the attractor onto
    // (width * scale)^2 pixels whatever the height, so that is the measure: 32-bit wins at 2048^2 / 2560^2 / 3072^2,
    // 16-bit at 3840x2160 (-8 %) and 4096^2 (-13 %).
    const double span = static_cast<double>(cfg->width) * cfg->scale;
    pl.hint_bytes = rt->hint_bits ? rt->hint_bits / 8u : ((span * span <= kWideHintMaxSpan2 && rt->npix <= (16u << 20)) ? 4u : 2u);
    // checkpoint stride: a multiple of the depth pipeline's pass length (the iterate kernel runs whole passes)
    pl.C = ((rt->ckpt_stride + kDefaultDepthPipe - 1u) / kDefaultDepthPipe) * kDefaultDepthPipe;
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
    // several launch chunks: their boundaries fall on whole workgroups (a job list that fits ONE launch is launched as it is — round
    // 5: 1447 jobs used to run as 1280 + 167)
    if (pl.chunk_jobs > pl.block && pl.chunk_jobs < n_jobs) pl.chunk_jobs -= pl.chunk_jobs % pl.block;
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
            cover = (512u + pl.geo.bins * batch_frames - 1u) / (pl.geo.bins * batch_frames);  // (a batch: over all its frames)
            pl.splits = (pl.max_waves + 8u * groups - 1u) / (8u * groups);
        }
        if (pl.splits < cover) pl.splits = cover;
        if (pl.splits < 1) pl.splits = 1;
        if (pl.splits > 16) pl.splits = 16;
    }
    // lists a lane group walks at the same time: with the 128 KiB histogram one workgroup per CU is resident — four loads
    // in flight per lane make up for the missing second workgroup (2048^2: 0.61 -> 0.46 ms); with two workgroups per CU
    // (64 KiB) more loads in flight change nothing
    pl.acc_lists = rt->acc_lists ? rt->acc_lists : (pl.geo.shift >= 15u ? 4u : 1u);
    return SAR_OK;
}


extern "C" {

int sar_bin_geometry(uint32_t width, uint32_t height, uint32_t bin_shift, uint32_t bin_interleave, uint32_t out[8]) try {
    if (!out || width == 0 || height == 0 || (bin_shift && (bin_shift < 12 || bin_shift > 16)) || bin_interleave > 2) return SAR_ERR_INVALID;
    const uint64_t npix64 = static_cast<uint64_t>(width) * height;
    if (npix64 > 0x7FFFFFFFull) return SAR_ERR_RANGE;  // what a runtime accepts (alloc_image_buffers)
    BinGeometry g;
    if (bin_shift == 0 && bin_interleave == 0) {  // what choose_chunk_records picks for a launch that wants two waves per SIMD
        bool found = false;
        for (uint32_t cand : {60u, 28u}) {
            for (uint32_t sh : {15u, 16u}) {
                g = bin_geometry(static_cast<uint32_t>(npix64), 256u, sh, 0u, 12u, 2u);
                if (g.ok && g.interleaved && lean_wave_lds_bytes(g.bins, cand) * 8u <= 160u * 1024u) { found = true; break; }
            }
            if (found) break;
        }
        if (!found) g = bin_geometry(static_cast<uint32_t>(npix64), 256u, 0u, 0u, 12u, 0u);
    } else {
        g = bin_geometry(static_cast<uint32_t>(npix64), 256u, bin_shift, 0u, 12u, bin_interleave);
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
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
