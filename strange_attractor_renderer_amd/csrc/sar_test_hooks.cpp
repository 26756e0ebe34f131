// sar_test_hooks.cpp — the A/B and test options of a runtime (include/sar_test_hooks.h). NOT part of libsar_hip.so: this file is
// linked only into the hooks build the test-suite and the A/B tools load (tests/hooks/libsar_hip_hooks.so: the product's own object
// files plus this one), so that the shipped library's ABI is include/sar.h and nothing else.
#include <cstring>
#include <new>
#include <stdexcept>

#include "../../include/sar_test_hooks.h"
#include "sar_plan.hpp"

using namespace sar;

extern "C" {

int sar_runtime_set_test_option(sar_runtime* rt, const char* name, uint64_t value) try {
    if (name && !std::strcmp(name, "debug_throw")) {  // (no runtime needed) what an exception inside an entry point becomes at the ABI
        if (value == 1) throw std::bad_alloc();
        if (value == 2) throw std::runtime_error("thrown on request");
        if (value == 3) throw 42;
        return SAR_OK;
    }
    if (!rt || !name) return SAR_ERR_INVALID;
    const uint32_t v = static_cast<uint32_t>(value);
    if (!std::strcmp(name, "path")) {
        if (v != 0 && v != 1 && v != 3) { set_error("path must be 0 (automatic), 1 (one global atomic per visit) or 3 (LDS-binned records)"); return SAR_ERR_INVALID; }
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
    } else if (!std::strcmp(name, "hint_tile")) {
        if (v > 1) { set_error("hint_tile must be 0 (automatic) or 1 (row-major hints)"); return SAR_ERR_INVALID; }
        if (rt->hint_tile != v && rt->d_zhint) { HIP_TRY(hipSetDevice(rt->device)); SAR_TRY(clear_hints(rt)); }  // another layout: the old hints mean nothing
        rt->hint_tile = v;
    } else if (!std::strcmp(name, "acc_lists")) {
        if (v && v != 1 && v != 4) { set_error("acc_lists must be 0 (automatic), 1 or 4"); return SAR_ERR_INVALID; }
        rt->acc_lists = v;
    } else if (!std::strcmp(name, "hint_shared")) {
        if (v > 2) { set_error("hint_shared must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->hint_shared = v;
    } else if (!std::strcmp(name, "readback_inline")) {
        rt->readback_inline = v ? 1u : 0u;
    } else if (!std::strcmp(name, "batch_starts")) {
        if (v > 3) { set_error("batch_starts must be 0 (automatic), 1 (upload stream), 2 (launch stream) or 3 (read in place)"); return SAR_ERR_INVALID; }
        rt->batch_starts = v;
    } else if (!std::strcmp(name, "batch_warm")) {
        if (v > 2) { set_error("batch_warm must be 0 (automatic), 1 (one phase) or 2 (two phases)"); return SAR_ERR_INVALID; }
        rt->batch_warm = v;
    } else if (!std::strcmp(name, "batch_chain")) {
        rt->batch_chain = v ? 1u : 0u;
    } else if (!std::strcmp(name, "batch_xcd")) {
        if (v > 1) { set_error("batch_xcd must be 0 (frames dealt to the XCDs) or 1 (every frame on all XCDs)"); return SAR_ERR_INVALID; }
        rt->batch_xcd = v;
    } else if (!std::strcmp(name, "chunk_ahead")) {
        if (v > 2) { set_error("chunk_ahead must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->chunk_ahead = v;
    } else if (!std::strcmp(name, "acc_threads")) {
        if (v && v != 256 && v != 512 && v != 1024) { set_error("acc_threads must be 256, 512 or 1024"); return SAR_ERR_INVALID; }
        rt->acc_threads = v;
    } else if (!std::strcmp(name, "debug_chunk_jobs")) {
        rt->debug_chunk_jobs = v;
    } else if (!std::strcmp(name, "debug_max_ordinals")) {
        rt->max_ordinals = value > kMaxChunkOrdinals ? kMaxChunkOrdinals : value;  // test hook: visits one launch may order
    } else {
        set_error("unknown test option '%s'", name);
        return SAR_ERR_INVALID;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

// The individual spans a runtime holds (timing_accumulate: one per launch of every render call since they were last read), in
// launch order: which = 0 iterate, 1 accumulate + fold, 2 warm-up. Synchronises the runtime's stream; clears nothing.
int sar_runtime_debug_spans(sar_runtime* rt, uint32_t which, float* out_ms, uint32_t cap, uint32_t* out_n) try {
    if (!rt || !out_n || which > 2u) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    const std::vector<Span>& spans = which == 0u ? rt->iter_spans : which == 1u ? rt->fold_spans : rt->warm_spans;
    const size_t used = which == 0u ? rt->iter_used : which == 1u ? rt->fold_used : rt->warm_used;
    *out_n = static_cast<uint32_t>(used);
    for (size_t k = 0; k < used && k < cap && out_ms; ++k) {
        float ms = 0.f;
        out_ms[k] = hipEventElapsedTime(&ms, spans[k].a, spans[k].b) == hipSuccess ? ms : -1.f;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
