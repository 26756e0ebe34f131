// sar_internal.hpp — declarations shared by the host and device halves of libsar_hip.so.
#pragma once

#include <cstdarg>
#include <cstddef>
#include <cstdint>

#include "../../include/sar.h"

namespace sar {

extern thread_local char g_last_error[512];
void set_error(const char* fmt, ...);
// What an exception becomes at the C ABI (every entry returns a status, nothing unwinds across the boundary): every
// int-returning entry point is a function-try-block whose handler returns this — std::bad_alloc -> SAR_ERR_OOM, anything else ->
// SAR_ERR_INVALID with what() in sar_last_error().
int abi_caught() noexcept;

// Start-point stream (include/sar.h: sar_start_points): xoshiro256++ seeded through SplitMix64, in BLOCKS of
// kStartBlockJobs jobs — block b draws from the generator after b applications of xoshiro256's published jump()
// (2^128 steps each), so that any job's point is found without drawing its predecessors' (a multi-GPU frame draws
// its job slices on one host thread per device).
constexpr uint32_t kStartBlockJobs = 4096;
struct Rng {
    uint64_t s[4];        // the generator the current block draws from
    uint64_t base[4];     // its state at the start of the current block
    uint32_t in_block;    // jobs drawn from the current block
    void seed(uint64_t seed);
    uint64_t next_u64();                  // one raw xoshiro256++ output (no block logic)
    void start_point(double out[3]);      // the next job's point
    void skip_points(uint64_t n_jobs);    // as if n_jobs points had been drawn: whole blocks by jump(), the rest by drawing
    static void jump(uint64_t st[4]);     // xoshiro256's jump(): 2^128 steps
};

void rotation_matrix(const sar_config& cfg, double m[9]);
int validate(const sar_config* cfg);

// ---- device-side argument blocks (passed by value as kernel arguments -> SGPRs) -------------------

// Everything the per-iteration arithmetic needs; hoisted exactly like render's setup
// (reference src/lib.rs:755-764).
struct MapParams {
    double cx[10], cy[10], cz[10];  // PolynomialSprott2Degree coefficients (c0 canonicalised: 0. + 1.*c0)
    double m[9];                    // rotation matrix, row-major
    double sin_v, cos_v;            // config.angle
    double ccx, ccy, ccz;           // center_camera
    double width, height;           // as f64
    double half_height;             // height / 2.
    double width_scaled;            // width * scale
    double scale_adjusted_mid;      // 0.5 / scale
};

struct ColorTransformParams {
    int32_t kind;
    int32_t _pad;
    double offset, factor;  // AdjustedVelocity
    double ccx, ccy;        // center_camera.x / .y used by poisson_saturne's part()
};

struct IterArgs {
    MapParams p;
    uint64_t iters;            // counted iterations per job (<= 0xFFFFFFFE)
    uint32_t n_jobs;           // jobs in this launch chunk; n_jobs*iters <= 0xFFFFFFFE
    uint32_t width;            // image width (index = j*width + i)
    uint32_t npix;             // width*height (stride between scratch copies)
    uint32_t ckpt_stride;      // iterations between trajectory checkpoints
    uint32_t resume;           // 1: `starts` holds the trajectory state a previous segment of this job left (no warm-up)
    uint32_t _pad_resume;
    const double* starts;      // [3][n_jobs] SoA, pre-warm-up start points (or, with `resume`, the carried state)
    double* state_out;         // nullable: [3][n_jobs] SoA, the state after the last iteration (for the job's next segment)
    uint32_t* scratch_count;   // [copies][npix]
    unsigned long long* scratch_key;  // [copies][npix]
    double* ckpt;              // [n_ckpt][3][n_jobs]
};

// Pixel -> (bin, 16-bit record) map of the LDS-binned path. Linear: bin = idx >> k, record = idx & (2^k - 1) — a bin is 2^k
// consecutive pixels. Interleaved (B = 2^b bins): the image is dealt to the bins in segments of 2^s pixels, round-robin —
// idx = [hi | bin (b bits) | lo (s bits)], record = [hi | lo] — so that every bin receives the same share of the visits
// whatever part of the image the attractor covers: the slot requests of a wave spread over all B counters, and the
// accumulate workgroups (one bin each) get equal work. One formula serves both (linear: s = k, hi_shift = 31):
//   bin = bfe(idx, seg_shift, bin_bits);  record = (idx & low_mask) | ((idx >> hi_shift) & ~low_mask)
//   idx = (record & low_mask) | (bin << seg_shift) | ((record & ~low_mask) << hi_shift)
struct BinMap {
    uint32_t seg_shift, bin_bits, hi_shift, low_mask;
};

// Layout of the NARROW depth hints for images whose width is a power of two: tiles of 8 x 8 pixels = one 128-byte line of
// 16-bit hints, instead of 64 x 1 row pieces. The attractor covers solid regions, and a square footprint per line needs
// fewer lines for the same pixels: at 4096^2 the hints an XCD touches (6.4 MB against 4 MB of L2) miss 10 % instead of 15 %
// (LRU model on the oracle's visit stream; measured: DESIGN.md section 3.2). The permutation moves j0..j2 under i3..:
//   t = bfi(mask1, idx >> shift1, idx << 3);  hint index = bfi(mask2, t, idx)     (mask2 = 0: the identity)
struct HintTile {
    uint32_t shift1, mask1, mask2, _pad;
};

// LDS-binned iterate kernel (see sar_iterate.hip: k_iterate_lean).
struct BinIterArgs {
    IterArgs it;                 // scratch_count unused here (counts travel as records)
    BinMap map;                  // pixel -> (bin, record)
    uint32_t n_bins;             // B = ceil(npix / bin_px) <= kMaxBins
    uint32_t chunks_per_wave;    // arena capacity of one wave, in 64-byte chunks
    uint32_t n_waves;            // launched waves (= heads stride)
    void* arena;                 // [n_waves][chunks_per_wave] chunks {prev, n, R x u16}, R = 12 / 20 / 28
    uint32_t* heads;             // [n_bins][n_waves] last chunk of each (bin, wave) list, or kNoChunk
    void* zhint;                 // [8][npix(+1)] per-XCD depth hints: u16 fixed point (depth_q16) or the depth itself as f32 bits
    // k_warmup's output: the trajectories that are still finite after the warm-up, packed densely
    const double* warm;          // [3][n_jobs] SoA by packed slot: the point after 1000 warm-up iterations
    const uint32_t* joblist;     // [n_jobs] job index (within this launch chunk) of every packed slot
    const uint32_t* active;      // number of packed slots
    const unsigned long long* warm_nan;  // nullable: iterations of the jobs that died in the warm-up (from k_warmup; the first
                                 // workgroup adds them to nan_count — the warm-up may have run ahead, under the previous frame)
    unsigned long long* nan_count;  // iterations of diverged (NaN) trajectories: all land on pixel (0,0)
    uint32_t hint_copy_mask;     // 7: one array of hints per XCD (index = XCC id); 0: one array for the whole chip
    uint32_t _pad_hint;
    const uint32_t* hint_range;  // nullable: {~sortable(min z), sortable(max z)} of the view, from k_warmup: the range the narrow
                                 // depth hints quantise (fixed from the first launch after the hints were cleared)
    double* warm_out;            // nullable: the state after the last iteration, by packed slot (== warm: the next segment of
                                 // jobs with more than 2^32-2 iterations starts from it, without a warm-up)
    HintTile tile;               // narrow hints: where pixel idx keeps its hint (identity unless the width is a power of two)
};

struct BinAccArgs {
    uint32_t bin_shift, n_bins, chunks_per_wave, n_waves;   // bin_shift: log2(pixels per bin); 16: two workgroups per (bin, split),
                                                            // one per half of the bin's pixels (the histogram holds 32768)
    uint32_t npix, splits, _pad0, _pad1;
    BinMap map;
    const void* arena;
    const uint32_t* heads;
    uint32_t* scratch_count;     // [splits][npix]: all-zero before the launch, the non-zero counts are stored
    uint32_t* seg_any;           // [ceil(npix / 2048)] set to 1 for every 2048-pixel segment of the image that received a
                                 // count (zero before the launch)
};

struct FoldArgs {
    MapParams p;
    ColorTransformParams ct;
    uint64_t iters;
    uint32_t n_jobs;
    uint32_t npix;
    uint32_t ckpt_stride;
    uint32_t copies;             // scratch_count copies
    uint32_t key_copies;         // scratch_key copies
    uint32_t _pad_fold;
    const uint32_t* seg_any;     // binned path: [ceil(npix / 2048)] "some visit landed in this 2048-pixel segment" (nullptr:
                                 // fold everything)
    unsigned long long* nan_count;  // nullable; added to pixel 0 and cleared
    uint32_t* count;                 // persistent [npix]
    unsigned long long* key;         // persistent [npix]: hi = sortable(zbuf), lo = 0xFFFFFFFF
    double* steps;                   // persistent [npix]
    uint32_t* scratch_count;         // [copies][npix], zeroed again by the fold
    unsigned long long* scratch_key; // [copies][npix]
    const double* ckpt;
    uint32_t* scalars;               // [0] max, [1] wrap flag
};

// k_warmup (sar_iterate.hip): the 1000 uncounted iterations of `n_jobs` jobs and the packing of the survivors
struct WarmArgs {
    MapParams p;
    const double* starts;        // [3][n_jobs] SoA: the start points — or, for the second phase, the points the first phase packed
    uint32_t n_jobs;
    uint32_t width;
    uint64_t iters;              // counted iterations per job: what a job that dies in the warm-up adds to nan_count
    double* warm;                // [3][n_jobs] SoA by packed slot
    uint32_t* joblist;           // [n_jobs]
    uint32_t* active;            // survivors (zero before the launch)
    unsigned long long* nan_count;
    uint32_t* hint_range;        // nullable: the depth range of the view (narrow hints)
    // The warm-up in TWO phases (batched launches of a preset that loses jobs): the first phase runs the iterations within
    // which trajectories diverge and packs the survivors, the second runs the rest on full waves of survivors. Same
    // iterations per job, in the same order.
    uint32_t n_iter;             // warm-up iterations this launch runs (1000 in one phase)
    uint32_t _pad_iter;
    const uint32_t* in_active;   // nullable: how many input slots hold a job (the first phase's survivor count); else n_jobs
    const uint32_t* in_joblist;  // nullable: the job index of every input slot; else the slot itself
};

// One frame of a BATCHED launch (sar_batch.cpp): F frames of the same shape — a `sequence` sweep's consecutive frames, each with
// its own Runtime, view angle and start points (src/bin/main.rs:493-517) — go through ONE launch of every kernel of the
// binned path; a workgroup finds its frame in blockIdx.z and its argument block in a table of these in device memory.
struct BatchFrame {
    WarmArgs warm;               // the warm-up, or its second phase
    WarmArgs warm_first;         // its first phase (n_iter == 0: one phase)
    BinIterArgs it;
    BinAccArgs acc;
    FoldArgs fold;
    uint32_t* seg_any;           // cleared by k_batch_clear before the launch, like `active` (both phases') and (if measured) `hint_range`
    uint32_t seg_words;
    uint32_t clear_hint_range;
    // k_batch_fetch: the start points from the page-locked staging buffer into device memory (n_start_quads 16-byte pieces)
    const void* starts_host;
    void* starts_dev;
    uint32_t n_start_quads;
    uint32_t _pad_fetch;
};
constexpr uint32_t kMaxBatchFrames = 32;

// Sparse exchange (sar_image.hip): a RECORD is one granule — 64 consecutive pixels — of one rank's partial buffers, 1 KiB as
// [count u32 x 64 | sortable(zbuf) u32 x 64 | steps f64 x 64]; only granules that differ from the reset state travel (a frame
// touches a fifth of its pixels, and 21 % of its 64-pixel granules: measured on BASELINE configs[1]; whole 2048-pixel rows: 51 %).
constexpr uint32_t kExchSeg = 64;
constexpr size_t kExchRecordBytes = (size_t)kExchSeg * 16u;
constexpr uint32_t kMaxExchRanks = 256;      // ranks one sar_exchange serves (the per-owner counters of k_exch_plan live in LDS)
constexpr uint32_t kExchSliceAlign = 2048;   // slices are whole 2048-pixel blocks (k_fold_resolve's), hence whole granules
constexpr uint32_t kMaxExchDevices = 64;
// k_exch_push (the multi-device renderer): device `src` writes the records of its touched granules straight into every owner's
// record buffer (peer memory: over xGMI) and tells the owner where they are (slot table) — nothing else crosses the links
struct ExchPushArgs {
    const uint32_t* count;
    const unsigned long long* key;
    const double* steps;
    uint32_t npix, nseg, sps, src, G, _pad;     // nseg: granules of the image, sps: granules per slice
    unsigned long long* bytes;                  // statistic: bytes written into OTHER shards' buffers (on a node: over xGMI)
    unsigned char* recv[kMaxExchDevices];       // owner o's record buffer [G * sps records]
    int32_t* slot[kMaxExchDevices];             // owner o's slot table [G][sps]: record index of (source, segment) or -1
};

struct PaletteParams {
    uint32_t len;  // user entries; entry len == entry len-1 (Palette::new, src/lib.rs:416-418)
    uint32_t _pad;
    double rgb[SAR_PALETTE_MAX + 1][3];
};

// the tables of the batched reset / colorize launches (kernel arguments by value: 32 frames fit the 4 KB of a launch)
struct ResetBatch {
    struct Frame {
        uint32_t* count;
        unsigned long long* key;
        double* steps;
        uint32_t* scalars;
        uint32_t* hints;
        uint32_t hint_words, hint_fill;
    } f[kMaxBatchFrames];
};
struct ColorizeBatch {
    struct Frame {
        const uint32_t* count;
        const double* steps;
        const uint32_t* scalars;
        void* out;
    } f[kMaxBatchFrames];
};

enum ScalarSlot : uint32_t {
    SC_MAX = 0,        // Runtime::max
    SC_WRAP = 1,       // a count wrapped u32 (running max would have hit u32::MAX)
    SC_ZMAX = 2,       // depth colorize: sortable(max z)
    SC_ZMIN = 3,       // depth colorize: sortable(min z)
    SC_COUNT = 8
};

// sortable u32 image of an f32 (monotone for all non-NaN values)
static inline uint32_t f32_sortable_host(float f) {
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// per-XCD hint arrays: an even number of entries each, so that the dword holding a 16-bit hint is aligned (constexpr: host and device)
constexpr size_t kHintStride(uint32_t npix) { return ((size_t)npix + 1u) & ~(size_t)1u; }
constexpr uint32_t kLnLutEntries = 1u << 20;  // ln(k+1), k < 2^20, host libm (exact parity with the oracle)
constexpr uint64_t kMaxChunkOrdinals = 0xFFFFFFFEull;
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;
constexpr uint32_t kDefaultChunkRecords = 28;  // u16 records per chunk (8-byte header): 12 / 20 / 28 / 60 -> 32 / 48-on-64 / 64 / 128-byte chunks
constexpr double kWideHintMaxSpan2 = 11.0e6;  // (width * scale)^2 up to which 32-bit depth hints are used
constexpr uint32_t kDefaultDepthPipe = 2;   // visits between a depth-hint load and its use in the iterate kernel
constexpr uint32_t kMaxBins = 1024;      // LDS staging is 64 B per bin per wave
constexpr uint32_t kMaxBinPx = 65536;    // a record is 16 bits; k_bin_accumulate counts a bin of 65536 pixels with packed 16-bit counters
constexpr uint32_t kMaxHistPx = 32768;   // its LDS histogram: 4 B per pixel, 128 KiB

}  // namespace sar
