/*
 * sar_test_hooks.h — the A/B and test options of a runtime. NOT part of the product's ABI (include/sar.h): the function below is
 * defined in csrc/sar_test_hooks.cpp, which is linked only into the hooks build (tests/hooks/libsar_hip_hooks.so = the object
 * files of libsar_hip.so + that one file; strange_attractor_renderer_amd/build.py makes both). The test-suite, tools/ and the
 * A/B switches of bench.py load the hooks build; __graft_entry__.smoke() and bench.py's measurements load the product.
 */
#ifndef SAR_TEST_HOOKS_H
#define SAR_TEST_HOOKS_H

#include "sar.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Options by name (value 0 restores the default unless noted):
 *   "path"               accumulate path: 0 default (= 3 when the image fits, up to 64 Mpx), 1 one global atomic per
 *                        visit, 3 LDS-binned records (an error where they do not fit)
 *   "bin_shift"          log2(pixels per bin) of the binned path (12..16; 16: the accumulate kernel packs two 16-bit counters
 *                        with a guard bit into an LDS word)
 *   "bin_interleave"     which pixels form a bin: 1 consecutive pixels, 2 every B-th 2048-pixel segment of the image
 *                        (B bins, a power of two: every bin carries the same share of the visits whatever the attractor
 *                        covers); 0 = 2 when the power-of-two bin count costs at most a third more bins, else 1
 *   "splits"             workgroups per bin in the record-accumulate kernel (1..16)
 *   "chunk_records"      u16 records per chunk: 12, 20, 28 or 60 (32/48/64/128-byte chunks; fewer = less LDS per wave)
 *   "hint_shared"        1: one array of depth hints per XCD, 2: one array for the whole chip (each XCD's L2 then sees the
 *                        others' updates late — more visits pass the filter, none wrongly); 0 = per XCD unless the eight
 *                        copies exceed 200 MB (then they would not fit the Infinity Cache)
 *   "hint_tile"          1: 16-bit hints always in row-major order; 0 = in 8 x 8 tiles (one 128-byte line each) where the
 *                        image width is a power of two and the height a multiple of eight
 *   "chunk_ahead"        a render call of several launch chunks runs its next chunk's warm-up ahead, under the current chunk's
 *                        accumulate and fold (0 / 1, the default); 2 = not (A/B)
 *   "acc_threads"        threads per block of the record-accumulate kernel (256, 512, 1024)
 *   "acc_lists"          (bin, wave) record lists a lane group of that kernel walks at the same time: 1 or 4
 *   "debug_chunk_jobs"   test hook: cap on jobs per launch chunk
 *   "debug_throw"        test hook (rt may be NULL): 1 / 2 / 3 throw std::bad_alloc / std::runtime_error / an int inside the entry point —
 *                        the call returns SAR_ERR_OOM / SAR_ERR_INVALID with the text in sar_last_error(): nothing unwinds across the ABI
 *   "debug_max_ordinals" test hook: visits one launch may order (default 2^32-2); jobs with more iterations run as segments
 *
 *   "readback_inline"    1: the read-back of sar_colorize_format_async stays on the launch stream (A/B of the copy stream)
 *   "batch_starts"       how a batched launch gets its start points: 0 = read in place by the one-phase warm-up / fetched by a
 *                        kernel before a two-phase one, 1 = copied on an upload stream, 2 = on the launch stream, 3 = always in place
 *   "batch_warm"         the warm-up of a batched launch: 0 = two phases when the last launch lost a tenth of its jobs, 1 = one, 2 = two
 *   "batch_chain"        1: the iterate kernels of a device's batches are not chained one behind the other
 *   "batch_xcd"          1: the frames of a batch are not dealt to the XCDs (every frame runs on all eight)
 */
int sar_runtime_set_test_option(sar_runtime* rt, const char* name, uint64_t value);

/* Measurement hook: the individual HIP-event spans the runtime holds ("timing_accumulate": one per launch of every render call
 * since sar_runtime_last_timing last read them), in launch order. which = 0 the iterate kernel, 1 accumulate + fold, 2 warm-up.
 * Writes min(*out_n, cap) values; synchronises the runtime's stream. (tools/dispatch_times.py: one kernel's dispatch-by-dispatch
 * durations with and without a tracer attached.) */
int sar_runtime_debug_spans(sar_runtime* rt, uint32_t which, float* out_ms, uint32_t cap, uint32_t* out_n);

#ifdef __cplusplus
}
#endif
#endif /* SAR_TEST_HOOKS_H */
