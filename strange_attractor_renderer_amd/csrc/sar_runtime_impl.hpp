// sar_runtime_impl.hpp — the private layout of sar_runtime and the few internals of sar_runtime.cpp that the
// multi-device ParallelRenderer (sar_multi.cpp) builds on. Not part of the ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <vector>

#include "sar_launch.hpp"

namespace sar {

struct Span {
    hipEvent_t a = nullptr, b = nullptr;
};

constexpr uint32_t kDefaultBlock = 256;
// iterations between trajectory checkpoints: k_fold_resolve replays on average half a stride per new depth winner (32: the 4096^2
// share -1.3 %, a sequence frame -2 %, 2048^2 -0.5 % against 64; 16 costs the iterate kernel more than the fold saves)
constexpr uint32_t kDefaultCkptStride = 32;
constexpr uint32_t kBatchRing = 8;               // page-locked copies of a batch's argument table in flight
constexpr uint64_t kCkptBytesCap = 24ull << 30;  // checkpoint + record-arena scratch per launch chunk (HBM is 288 GB)

// What the runtimes of a frame group (sar_runtime_new_group: the frames of one batched launch) share — the launch stream, the
// read-back stream and one page-locked allocation for their start points.
struct RuntimeGroup {
    int refs = 0;  // runtimes alive
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    char* hslab = nullptr;
    size_t hslab_bytes = 0;
};

}  // namespace sar

struct sar_runtime {
    int device = 0;
    // frame group (nullptr: a runtime of its own). `sub` is ONE device allocation of this runtime its buffers are carved from
    // (sized from the plan of the frames the group was made for: some twenty hipMalloc / hipFree calls less per runtime — a hipFree
    // costs 0.16 ms; one allocation for the WHOLE group, 10 GB, took between 0.3 ms and two seconds on the same box), `hsub` its
    // share of the group's page-locked allocation; both handed out front to back by dev_alloc / host_alloc and never reused (what
    // does not fit comes from hipMalloc / hipHostMalloc)
    sar::RuntimeGroup* group = nullptr;
    char* sub = nullptr;
    size_t sub_bytes = 0, sub_used = 0;
    char* hsub = nullptr;
    size_t hsub_bytes = 0, hsub_used = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint32_t W = 0, H = 0, npix = 0;
    uint32_t sm_count = 0;

    // persistent state (Runtime, reference src/lib.rs:631-646)
    uint32_t* d_count = nullptr;            // count
    unsigned long long* d_key = nullptr;    // hi: sortable(zbuf), lo: 0xFFFFFFFF between launches
    double* d_steps = nullptr;              // steps
    uint32_t* d_scalars = nullptr;          // max + flags + depth range
    sar::Rng rng;

    // scratch bins the iterate kernel accumulates into (zero between launches)
    uint32_t copies = 0;      // scratch_count copies
    uint32_t* d_scratch_count = nullptr;
    unsigned long long* d_scratch_key = nullptr;

    // binned path: per-wave record arenas, list heads, per-XCD depth hints, NaN iteration counter
    void* d_arena = nullptr;
    size_t arena_cap = 0;  // bytes
    uint32_t* d_heads = nullptr;
    size_t heads_cap = 0;  // entries
    void* d_zhint = nullptr;
    uint32_t zhint_bytes = 0;        // bytes per hint of the current allocation (2 or 4)
    uint32_t hint_copies_used = 8;   // of the eight per-XCD arrays, how many [0, n) a launch has written since they were last
                                     // cleared (a launch whose XCDs share ONE array writes array 0 only): what clear_hints clears
    uint32_t hint_copies_alloc = 0;  // arrays the allocation holds: 8, or 1 where every launch so far shared one array (a runtime of
                                     // a frame group — its frames are dealt to XCDs of their own —, an image beyond 200 MB of hints)
    bool single_hint_array = false;  // a runtime of a frame group: its launches share ONE hint array whatever their form
    uint32_t hint_bits = 0;          // option: 0 = by image size, 16, 32
    uint32_t hint_tile = 0;          // option: 0 = narrow hints of power-of-two-wide images in 8 x 8 tiles, 1 = always row-major
    uint32_t hint_shared = 0;        // option: 0 = automatic, 1 = one hint array per XCD, 2 = one array for the whole chip
    unsigned long long* d_nan_count = nullptr;
    uint32_t* d_hint_range = nullptr;   // {~sortable(min z), sortable(max z)}: what the narrow depth hints quantise (HintQuant)
    bool hint_range_set = false;        // measured since the hints were last cleared (the quantiser must not change under them)

    // staging
    double* h_starts = nullptr;  // pinned
    double* d_starts = nullptr;
    double* d_warm = nullptr;        // binned path: packed post-warm-up points, job list, survivor count
    uint32_t* d_joblist = nullptr;
    uint32_t* d_active = nullptr;    // [4]: survivors, pad, iterations of the jobs that died in the warm-up (u64)
    // The warm-up of an ANNOUNCED render call (sar_runtime_prefetch_device) runs ahead on a side stream, under the current
    // frame's accumulate / fold / colorize, into a second set of these buffers; the announced call swaps the sets.
    double* d_warm_alt = nullptr;
    uint32_t* d_joblist_alt = nullptr;
    uint32_t* d_active_alt = nullptr;
    uint32_t* d_hint_range_alt = nullptr;
    double* d_starts_alt = nullptr;
    size_t warm_alt_cap = 0;         // jobs (of d_warm_alt / d_joblist_alt: the two sets swap, capacities included)
    size_t starts_alt_cap = 0;       // jobs (of d_starts_alt, which does not swap)
    hipStream_t side = nullptr;
    hipEvent_t iter_done = nullptr, pf_done = nullptr;
    hipEvent_t prefetch_after = nullptr;  // set around sar_runtime_prefetch_device by the multi-device renderer: the side stream
                                          // waits for it (the upload of the announced points) instead of the host
    bool iter_done_recorded = false;
    struct Prefetch {
        bool valid = false;
        sar::MapParams p;
        uint32_t n_jobs = 0, m = 0, width = 0;
        uint64_t iters = 0;
        const double* starts = nullptr;
        bool range_measured = false;
    } pf;
    uint32_t prefetch_used = 0;      // statistic: launches that found their warm-up done
    uint32_t chunk_ahead = 0;        // option: 2 = a call of several launch chunks does not run its next chunk's warm-up ahead (A/B)
    hipEvent_t img_events[8] = {};   // sar_colorize_format_async tickets (ticket t is event t % 8: a later recording on the
    uint64_t img_next = 0;           // same stream completes no earlier, so waiting for it is always sufficient)
    // The read-back of an async frame runs on its own stream (the copy engine), behind `img_ready`: the launch stream goes on
    // with the next frame at once, and waits for the last read-back only before it writes d_rgba / d_export again.
    hipStream_t copy_stream = nullptr;
    bool own_copy_stream = false;
    hipEvent_t img_ready = nullptr;
    bool copy_in_flight = false;
    uint32_t readback_inline = 0;    // option: 1 = async read-backs stay on the launch stream (A/B)
    uint32_t batch_starts = 0;       // option: how a batched launch gets its start points: 0 = the warm-up kernel reads the page-locked
                                     // staging buffer itself (no copy), 1 = copied on the upload stream, 2 = copied on the launch stream
    // Start points of a batched launch are uploaded on the leader's upload stream, behind the last kernel that read d_starts
    // (`starts_consumed`, recorded by whatever launched it): the upload of batch k+1 runs under batch k.
    hipStream_t upload_stream = nullptr;
    hipEvent_t starts_consumed = nullptr;
    bool starts_consumed_recorded = false;
    char last_launch[256] = {0};     // sar_runtime_describe_last_launch
    uint32_t last_chunks = 0;
    uint32_t* d_seg_any = nullptr;   // [npix / 2048 + 1] 2048-pixel segments with a count in the current launch (k_fold_resolve skips the rest)
    size_t seg_any_cap = 0;
    size_t warm_cap = 0;             // jobs
    // survivor statistics of the last launch, copied back lazily (never waited for): the next render call sizes its
    // staging for the lanes that will really be busy (solar-sail loses 38 % of its jobs in the warm-up)
    uint32_t* h_active = nullptr;    // pinned
    hipEvent_t active_copied = nullptr;
    bool active_pending = false;
    uint32_t active_jobs_launched = 0;
    double survivor_fraction = 1.0;
    bool survivors_known = false;    // a launch has reported its survivors (until then the fraction is the optimistic default)
    size_t starts_cap = 0;       // doubles
    hipEvent_t starts_copied = nullptr;
    bool starts_pending = false;
    double* d_ckpt = nullptr;
    size_t ckpt_cap = 0;         // doubles
    double* d_lnlut = nullptr;
    void* d_rgba = nullptr;
    void* d_export = nullptr;  // converted image of sar_colorize_format (<= 6 bytes per pixel)
    const void* export_src = nullptr;  // where the last sar_colorize_format* left its image (d_rgba or d_export) and how long it is:
    size_t export_bytes = 0;           // what a read-back copies
    float* d_ztmp = nullptr;

    // batched launches (sar_batch.cpp). As the LEADER of a batch: the table of per-frame argument blocks in device memory and
    // the page-locked ring it is uploaded from (entry k % kBatchRing is free again once batch_copied[k % kBatchRing] has fired).
    // As any member: the event its own stream and the leader's stream meet through when they differ.
    sar::BatchFrame* d_batch = nullptr;
    sar::BatchFrame* h_batch = nullptr;
    hipEvent_t batch_copied[sar::kBatchRing] = {};
    uint64_t batch_next = 0;
    hipEvent_t batch_join = nullptr;
    uint32_t batches_launched = 0;   // statistic: batched launches this runtime led
    uint32_t batch_warm = 0;         // option: the warm-up of a batched launch: 0 = two phases when the last launch lost a tenth of its jobs, 1 = one phase, 2 = two
    uint32_t batch_chain = 0;        // option: 1 = the iterate kernels of this device's batches are NOT chained one behind the other (A/B)
    uint32_t batch_xcd = 0;          // option: 1 = the frames of a batch are NOT dealt to the XCDs (every frame runs on all eight: A/B)

    // tuning
    uint32_t block_threads = sar::kDefaultBlock;
    uint32_t ckpt_stride = sar::kDefaultCkptStride;
    uint32_t bins_mode = 0;     // 0 default (binned when eligible), 1 one global atomic per visit, 3 LDS-binned records or an error
    bool timing_accumulate = false;  // spans of successive render calls add up until sar_runtime_last_timing reads them
    uint32_t debug_chunk_jobs = 0;  // test hook: cap on jobs per launch chunk (0 = none)
    uint64_t max_ordinals = 0;      // test hook: visits one launch may order (0 = 2^32-2); longer jobs run as segments
    uint32_t bin_shift = 0;         // 0 = automatic
    uint32_t bin_interleave = 0;    // 0 = automatic, 1 = bins of consecutive pixels, 2 = interleaved bins (BinMap)
    uint32_t splits = 0;            // 0 = automatic
    uint32_t acc_threads = 0;       // threads per k_bin_accumulate block (0 = automatic)
    uint32_t split_waves = 0;       // 2: the iterate kernel as producer / consumer wave pairs (k_iterate_split) where it applies
    uint32_t acc_lists = 0;         // (bin, wave) lists a lane group of k_bin_accumulate walks at the same time: 1, 4 (0 = automatic)
    uint32_t chunk_records = 0;     // records per chunk: 12 / 20 / 28 / 60 (0 = automatic)

    // timing
    bool timing = false;
    std::vector<sar::Span> iter_spans, fold_spans, warm_spans;
    size_t iter_used = 0, fold_used = 0, warm_used = 0;
    sar::Span colorize_span, merge_span;
    bool colorize_timed = false, merge_timed = false;
    uint64_t last_iterations = 0;
};


#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            sar::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? SAR_ERR_OOM : SAR_ERR_HIP;           \
        }                                                                             \
    } while (0)

#define SAR_TRY(expr)                    \
    do {                                 \
        int s_ = (expr);                 \
        if (s_ != SAR_OK) return s_;     \
    } while (0)

namespace sar {

// Device / page-locked memory of a runtime: from its own slab / its share of its frame group's page-locked allocation while that
// lasts (256-byte granules, never reused), from hipMalloc / hipHostMalloc otherwise. dev_free / host_free leave slab memory alone
// (it goes with the runtime / with the group's last runtime).
hipError_t dev_alloc_bytes(sar_runtime* rt, void** out, size_t bytes);
hipError_t host_alloc_bytes(sar_runtime* rt, void** out, size_t bytes);
void dev_free(sar_runtime* rt, void* p);
void host_free(sar_runtime* rt, void* p);
template <typename T>
hipError_t dev_alloc(sar_runtime* rt, T** out, size_t bytes) { return dev_alloc_bytes(rt, reinterpret_cast<void**>(out), bytes); }
template <typename T>
hipError_t host_alloc(sar_runtime* rt, T** out, size_t bytes) { return host_alloc_bytes(rt, reinterpret_cast<void**>(out), bytes); }

// Grows a device buffer of `rt` (contents are not preserved). cap and need in elements of T.
template <typename T>
int grow_device(sar_runtime* rt, T*& ptr, size_t& cap, size_t need) {
    if (need <= cap) return SAR_OK;
    if (ptr) dev_free(rt, ptr);
    ptr = nullptr;
    cap = 0;
    HIP_TRY(dev_alloc(rt, &ptr, need * sizeof(T)));
    cap = need;
    return SAR_OK;
}

// sar_runtime.cpp
int clear_hints(sar_runtime* rt);  // hints are lower bounds of depths already accumulated; anything that can lower zbuf voids them
void span_begin(sar_runtime* rt, std::vector<Span>& spans, size_t& used);
void span_end(sar_runtime* rt, std::vector<Span>& spans, size_t& used);
void single_begin(sar_runtime* rt, Span& s);
void single_end(sar_runtime* rt, Span& s, bool& flag);
// sar_render.cpp
void fill_map_params(const sar_config& cfg, MapParams& p);      // render's hoisted constants (:755-764)
void fill_ct_params(const sar_config& cfg, ColorTransformParams& ct);

// Runs n_jobs trajectories of `iters` counted iterations each into rt (sequential job-major semantics); `starts` is
// [n_jobs][3] in host memory, or in device memory with starts_on_device. Enqueues only.
int render_chunked(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters, const double* starts,
                   bool starts_on_device = false);
// colorize of the pixel range [first, first + n) into out_dev (n * 8 bytes). global_scalars: the scalars (max, depth
// range) already hold the values of the WHOLE image (sliced multi-GPU colorize); otherwise the depth range is folded
// over the range first, as colorize does (:877-882).
int colorize_range(const sar_config* cfg, sar_runtime* rt, uint32_t first, uint32_t n, void* out_dev, bool global_scalars);
int check_cfg_matches(const sar_config* cfg, const sar_runtime* rt);

}  // namespace sar
