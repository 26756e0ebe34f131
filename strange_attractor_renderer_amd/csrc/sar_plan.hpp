// sar_plan.hpp — what one render call decides before it launches (sar_plan.cpp): the bin geometry of the LDS-binned path,
// chunk size, hint type, which form of the iterate kernel, and how the job list is cut into launch chunks. Not part of the ABI.
#pragma once

#include "sar_runtime_impl.hpp"

namespace sar {

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
BinGeometry bin_geometry(uint32_t npix, uint32_t want_block, uint32_t want_shift, uint32_t want_splits, uint32_t records, uint32_t interleave);

// Everything one render call decides before it launches: which accumulate path, its geometry, and how the job list is
// cut into launch chunks (chunk boundaries fall on whole jobs; a chunk keeps job*iters + t inside 32 bits and its
// scratch inside kCkptBytesCap).
struct LaunchPlan {
    bool binned = false;                // LDS-binned records (default) or one global atomic per visit (beyond 64 Mpx)
    bool split = false;                 // the iterate kernel as producer / consumer wave pairs (k_iterate_split)
    uint64_t resident_jobs = 0;         // trajectories resident at once under this plan: launch chunks are whole rounds of them
    BinGeometry geo;
    uint32_t R = kDefaultChunkRecords;  // records per chunk
    uint32_t block = 0;                 // trajectories per workgroup of the iterate kernel
    uint32_t hint_bytes = 4;            // depth hints: 2 (fixed point) or 4 (the depth itself as f32)
    uint32_t C = 0;                     // checkpoint stride
    uint32_t splits = 0;                // accumulate workgroups per bin
    uint32_t acc_lists = 1;             // (bin, wave) lists a lane group of k_bin_accumulate walks at the same time
    uint64_t n_ckpt = 0, chunks_per_wave = 0, chunk_jobs = 0;
    uint32_t max_waves = 0;             // waves of the largest launch chunk
    uint32_t arena_waves = 0;           // ... of which hold at least one job: only they own a slice of the record arena
};
// batch_frames > 1: the plan of ONE frame of a batched launch (sar_batch.cpp) — the form of the iterate kernel and the accumulate
// grid are chosen for batch_frames x n_jobs jobs on the chip at once
int plan_launch(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters, LaunchPlan& pl, uint32_t batch_frames = 1);

// sar_render.cpp: the pieces of a render call that the batched launch (sar_batch.cpp) shares
int ensure_scratch(sar_runtime* rt, uint32_t copies);
int stage_starts(sar_runtime* rt, const LaunchPlan& pl, uint32_t n_jobs, const double* starts, bool on_device, hipStream_t upload = nullptr,
                 bool in_place = false);
bool hints_shared(const sar_runtime* rt, const sar_runtime* opt, const LaunchPlan& pl, bool one_hint_array);  // a launch's XCDs use ONE hint array
int ensure_binned_buffers(sar_runtime* rt, const LaunchPlan& pl, uint32_t hint_copies);
void fill_iter_fold_args(const sar_config* cfg, sar_runtime* rt, const LaunchPlan& pl, IterArgs& ia, FoldArgs& fa);
void fill_bin_iter_args(sar_runtime* rt, const sar_runtime* opt, const LaunchPlan& pl, const IterArgs& ia, BinIterArgs& ba, bool* shared_out,
                        bool one_hint_array = false);
void fill_bin_acc_args(sar_runtime* rt, const LaunchPlan& pl, const BinIterArgs& ba, BinAccArgs& ca);
WarmArgs warm_args(const MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* warm, uint32_t* joblist,
                   uint32_t* active, uint32_t width, uint32_t* hint_range);
void describe_launch(sar_runtime* rt, const LaunchPlan& pl, bool share, uint32_t batch_frames, uint32_t xcd_map = 0);

}  // namespace sar
