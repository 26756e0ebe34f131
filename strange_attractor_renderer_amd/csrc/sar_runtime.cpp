// sar_runtime.cpp — the C ABI of the Runtime (include/sar.h): life cycle, reset, merge, colorize and image export, read-back
// accessors, timing and the tuning options. The render call itself is sar_render.cpp, its planning sar_plan.cpp, the multi-device
// ParallelRenderer sar_multi.cpp, the exchange of the one-process-per-GPU path sar_exchange.cpp.
//
// Host logic only (allocation, argument blocks, stream ordering); all arithmetic on image data happens in the kernel files
// (sar_iterate.hip, sar_accumulate.hip, sar_image.hip). There is no CPU fallback: without a HIP device every entry point
// that touches a runtime returns SAR_ERR_NO_DEVICE.
#include <sys/mman.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "sar_plan.hpp"

using namespace sar;

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

// The table in device memory: ONE per device, shared by every runtime there (read-only; it used to be 8 MiB of allocation and
// upload per runtime — 0.35 ms of each of a sweep's sixteen), released with the device's last runtime.
struct DeviceLnLut {
    std::mutex mu;
    double* table = nullptr;
    int refs = 0;
} g_lnlut[64];

int acquire_ln_lut(sar_runtime* rt) {
    if (rt->device < 0 || rt->device >= 64) { set_error("device %d: this library addresses devices 0..63", rt->device); return SAR_ERR_INVALID; }
    DeviceLnLut& d = g_lnlut[rt->device];
    std::lock_guard<std::mutex> lock(d.mu);
    if (!d.table) {
        double* t = nullptr;
        if (hipMalloc(&t, kLnLutEntries * sizeof(double)) != hipSuccess) return SAR_ERR_OOM;
        // (a blocking copy: the table is complete before any stream of any runtime can read it)
        if (hipMemcpy(t, host_ln_lut(), kLnLutEntries * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { hipFree(t); return SAR_ERR_HIP; }
        d.table = t;
    }
    ++d.refs;
    rt->d_lnlut = d.table;
    return SAR_OK;
}

void release_ln_lut(sar_runtime* rt) {
    if (!rt->d_lnlut || rt->device < 0 || rt->device >= 64) return;
    DeviceLnLut& d = g_lnlut[rt->device];
    std::lock_guard<std::mutex> lock(d.mu);
    if (--d.refs == 0) {
        hipFree(d.table);
        d.table = nullptr;
    }
    rt->d_lnlut = nullptr;
}

std::mutex g_group_mu;  // the reference counts of the frame groups

constexpr size_t kSlabAlign = 256;
size_t slab_round(size_t bytes) { return (bytes + kSlabAlign - 1) & ~(kSlabAlign - 1); }

}  // namespace

hipError_t sar::dev_alloc_bytes(sar_runtime* rt, void** out, size_t bytes) {
    const size_t need = slab_round(bytes ? bytes : 1);
    if (rt->sub && need <= rt->sub_bytes - rt->sub_used) {
        *out = rt->sub + rt->sub_used;
        rt->sub_used += need;
        return hipSuccess;
    }
    return hipMalloc(out, bytes);
}

hipError_t sar::host_alloc_bytes(sar_runtime* rt, void** out, size_t bytes) {
    const size_t need = slab_round(bytes ? bytes : 1);
    if (rt->hsub && need <= rt->hsub_bytes - rt->hsub_used) {
        *out = rt->hsub + rt->hsub_used;
        rt->hsub_used += need;
        return hipSuccess;
    }
    return hipHostMalloc(out, bytes, hipHostMallocDefault);
}

void sar::dev_free(sar_runtime* rt, void* p) {
    if (!p) return;
    const char* q = static_cast<const char*>(p);
    if (rt->sub && q >= rt->sub && q < rt->sub + rt->sub_bytes) return;  // goes with the runtime
    hipFree(p);
}

void sar::host_free(sar_runtime* rt, void* p) {
    if (!p) return;
    const char* q = static_cast<const char*>(p);
    if (rt->group && q >= rt->group->hslab && q < rt->group->hslab + rt->group->hslab_bytes) return;
    hipHostFree(p);
}

namespace {

int free_device_buffers(sar_runtime* rt) {
    if (rt->d_count) dev_free(rt, rt->d_count);
    if (rt->d_key) dev_free(rt, rt->d_key);
    if (rt->d_steps) dev_free(rt, rt->d_steps);
    if (rt->d_scratch_count) dev_free(rt, rt->d_scratch_count);
    if (rt->d_scratch_key) dev_free(rt, rt->d_scratch_key);
    if (rt->d_rgba) dev_free(rt, rt->d_rgba);
    if (rt->d_export) dev_free(rt, rt->d_export);
    rt->d_export = nullptr;
    rt->export_src = nullptr;
    if (rt->d_ztmp) dev_free(rt, rt->d_ztmp);
    if (rt->d_zhint) dev_free(rt, rt->d_zhint);
    rt->d_zhint = nullptr;
    rt->d_count = nullptr;
    rt->d_key = nullptr;
    rt->d_steps = nullptr;
    rt->d_scratch_count = nullptr;
    rt->d_scratch_key = nullptr;
    rt->d_rgba = nullptr;
    rt->d_ztmp = nullptr;
    rt->copies = 0;
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
    hipError_t e = dev_alloc(rt, &count, npix64 * sizeof(uint32_t));
    if (e == hipSuccess) e = dev_alloc(rt, &key, npix64 * sizeof(unsigned long long));
    if (e == hipSuccess) e = dev_alloc(rt, &steps, npix64 * sizeof(double));
    if (e != hipSuccess) {
        if (count) dev_free(rt, count);
        if (key) dev_free(rt, key);
        if (steps) dev_free(rt, steps);
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

}  // namespace

int sar::clear_hints(sar_runtime* rt) {
    // hints are lower bounds of depths already accumulated; anything that can lower zbuf voids them
    // Wide hints hold the depth itself as f32 and start at the smallest float above -1.0 (nextafter(-1, +inf) = 0xBF7FFFFF): stage 1's `z >= hint` is then the
    // reference's strict `z > -1.0` (:693, :821) for a pixel nobody has reached. Narrow hints are 16-bit fixed point from 0.
    // (only the arrays a launch has written since the last clear: a batched frame whose XCDs share one array leaves seven untouched)
    const size_t entries = kHintStride(rt->npix) * rt->hint_copies_used;
    if (rt->d_zhint && rt->zhint_bytes == 4 && entries)
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(rt->d_zhint), static_cast<int>(0xBF7FFFFFu), entries, rt->stream));
    else if (rt->d_zhint && entries)
        HIP_TRY(hipMemsetAsync(rt->d_zhint, 0, entries * rt->zhint_bytes, rt->stream));
    rt->hint_copies_used = 0;
    rt->hint_range_set = false;  // empty hints: the next launch may measure the view's depth range anew
    return SAR_OK;
}

namespace {

int do_reset(sar_runtime* rt) {
    // the hints go with the buffers, in the same launch (clear_hints: what they are cleared to, and which of them)
    const size_t entries = rt->d_zhint ? kHintStride(rt->npix) * rt->hint_copies_used : 0;
    const uint32_t words = static_cast<uint32_t>(rt->zhint_bytes == 4 ? entries : entries / 2u);  // (kHintStride is even)
    launch_reset(rt->d_count, rt->d_key, rt->d_steps, rt->npix, rt->d_scalars, rt->d_zhint, words, rt->zhint_bytes == 4 ? 0xBF7FFFFFu : 0u, rt->stream);
    rt->hint_copies_used = 0;
    rt->hint_range_set = false;
    HIP_TRY(hipGetLastError());
    return SAR_OK;
}

}  // namespace

void sar::span_begin(sar_runtime* rt, std::vector<Span>& spans, size_t& used) {
    if (!rt->timing) return;
    if (used == spans.size()) {
        Span s;
        hipEventCreate(&s.a);
        hipEventCreate(&s.b);
        spans.push_back(s);
    }
    hipEventRecord(spans[used].a, rt->stream);
}
void sar::span_end(sar_runtime* rt, std::vector<Span>& spans, size_t& used) {
    if (!rt->timing) return;
    hipEventRecord(spans[used].b, rt->stream);
    ++used;
}
void sar::single_begin(sar_runtime* rt, Span& s) {
    if (!rt->timing) return;
    if (!s.a) { hipEventCreate(&s.a); hipEventCreate(&s.b); }
    hipEventRecord(s.a, rt->stream);
}
void sar::single_end(sar_runtime* rt, Span& s, bool& flag) {
    if (!rt->timing) return;
    hipEventRecord(s.b, rt->stream);
    flag = true;
}

int sar::check_cfg_matches(const sar_config* cfg, const sar_runtime* rt) {
    SAR_TRY(validate(cfg));
    if (!rt) { set_error("runtime is NULL"); return SAR_ERR_INVALID; }
    if (cfg->width != rt->W || cfg->height != rt->H) {
        set_error("config is %ux%u but the runtime holds %ux%u", cfg->width, cfg->height, rt->W, rt->H);
        return SAR_ERR_DIM_MISMATCH;
    }
    return SAR_OK;
}

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

int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return SAR_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(device));
    return SAR_OK;
}

// Streams, events, the persistent buffers and the reset state of a new runtime (enqueued on its stream; the caller waits). A
// runtime of a frame group runs on the group's streams and draws its memory from the group's allocations.
int init_runtime(sar_runtime* rt, const sar_config* cfg, int device) {
    rt->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) rt->sm_count = static_cast<uint32_t>(prop.multiProcessorCount);
    if (rt->group) {
        rt->stream = rt->group->stream;
        rt->copy_stream = rt->group->copy_stream;
    } else {
        if (hipStreamCreateWithFlags(&rt->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return SAR_ERR_HIP; }
        rt->own_stream = true;
    }
    if (hipEventCreateWithFlags(&rt->starts_copied, hipEventDisableTiming) != hipSuccess) return SAR_ERR_HIP;
    if (dev_alloc(rt, &rt->d_scalars, SC_COUNT * sizeof(uint32_t)) != hipSuccess) return SAR_ERR_OOM;
    SAR_TRY(acquire_ln_lut(rt));
    SAR_TRY(alloc_image_buffers(rt, cfg->width, cfg->height));
    SAR_TRY(do_reset(rt));
    rt->rng.seed(cfg->seed);
    return SAR_OK;
}

// the last runtime of a group takes the group's streams and allocations with it (force: a group no runtime ever joined)
void release_group(RuntimeGroup* g, bool force) {
    {
        std::lock_guard<std::mutex> lock(g_group_mu);
        if (!force && --g->refs > 0) return;
    }
    hipSetDevice(g->device);
    if (g->stream) { hipStreamSynchronize(g->stream); hipStreamDestroy(g->stream); }
    if (g->copy_stream) { hipStreamSynchronize(g->copy_stream); hipStreamDestroy(g->copy_stream); }
    if (g->hslab) hipHostFree(g->hslab);
    delete g;
}

// Device and page-locked bytes ONE runtime of a frame group holds once it renders frames like cfg in batches of n: the persistent
// buffers, the scratch, one array of depth hints, the record arena, checkpoints, warm-up sets, start points, the colorized and the
// converted image. An estimate with slack — what it misses is allocated separately.
void group_bytes_per_runtime(const sar_config* cfg, sar_runtime* probe, uint32_t n, size_t& dev_bytes, size_t& host_bytes) {
    const size_t npix = probe->npix;
    size_t dev = 0, host = 0, items = 0;
    auto add = [&](size_t b) { dev += slab_round(b); ++items; };
    add(SC_COUNT * sizeof(uint32_t));
    add(npix * 4); add(npix * 8); add(npix * 8);  // count, key, steps
    add(npix * 8);                                // colorized image
    add(npix * 6);                                // converted image
    const uint32_t n_jobs = cfg->jobs_total;
    const uint64_t iters = n_jobs ? cfg->iterations / n_jobs : 0;
    LaunchPlan pl;
    if (n_jobs && iters && iters <= kMaxChunkOrdinals && plan_launch(cfg, probe, n_jobs, iters, pl, n) == SAR_OK && pl.binned) {
        add(npix * 8);                                                                            // depth keys of a launch
        add(npix * 4 * pl.splits);                                                                // partial histograms
        add((npix + 2) * pl.hint_bytes * (n >= 3u ? 1u : 8u));                                    // depth hints
        add(static_cast<size_t>(pl.arena_waves) * pl.chunks_per_wave * chunk_bytes(pl.R));        // record arena
        add(static_cast<size_t>(pl.max_waves) * pl.geo.bins * 4);                                 // list heads
        add(static_cast<size_t>(pl.n_ckpt) * 3 * pl.chunk_jobs * 8);                              // checkpoints
        for (int set = 0; set < 2; ++set) { add(static_cast<size_t>(pl.chunk_jobs) * 24); add(static_cast<size_t>(pl.chunk_jobs) * 4); }  // warm-up sets
        add((static_cast<size_t>(n_jobs) * 3 + 2) * 8);                                           // start points
        add((npix / 2048 + 1) * 4);                                                               // segment flags
        host += slab_round((static_cast<size_t>(n_jobs) * 3 + 2) * 8);
    }
    dev += slab_round(sizeof(BatchFrame) * kMaxBatchFrames);                                      // (a leader's argument table)
    host += slab_round(sizeof(BatchFrame) * kMaxBatchFrames * kBatchRing);
    dev_bytes = dev + 16 * kSlabAlign;   // the handful of counters
    host_bytes = host + 4 * kSlabAlign;
    (void)items;
}

int ensure_rgba(sar_runtime* rt) {
    if (!rt->d_rgba) HIP_TRY(dev_alloc(rt, &rt->d_rgba, static_cast<size_t>(rt->npix) * 8));
    return SAR_OK;
}

}  // namespace

extern "C" {

int sar_device_count(int* out_count) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_device_pci_bus_id(int device, char* out, size_t cap) try {
    if (!out || cap < 16) return SAR_ERR_INVALID;
    out[0] = 0;
    if (hipDeviceGetPCIBusId(out, static_cast<int>(cap), device) != hipSuccess) {
        set_error("hipDeviceGetPCIBusId(%d) failed", device);
        return SAR_ERR_NO_DEVICE;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_new(const sar_config* cfg, int device, sar_runtime** out) try {
    if (!out) return SAR_ERR_INVALID;
    *out = nullptr;
    SAR_TRY(validate(cfg));
    SAR_TRY(check_device(device));
    sar_runtime* rt = new (std::nothrow) sar_runtime();
    if (!rt) return SAR_ERR_OOM;
    const int st = init_runtime(rt, cfg, device);
    if (st != SAR_OK) { sar_runtime_free(rt); return st; }
    if (hipStreamSynchronize(rt->stream) != hipSuccess) { sar_runtime_free(rt); return SAR_ERR_HIP; }
    *out = rt;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_new_group(const sar_config* cfg, int device, uint32_t n, sar_runtime** out) try {
    if (!out || n == 0 || n > kMaxBatchFrames) { set_error("sar_runtime_new_group: 1..%u runtimes", kMaxBatchFrames); return SAR_ERR_INVALID; }
    for (uint32_t i = 0; i < n; ++i) out[i] = nullptr;
    SAR_TRY(validate(cfg));
    SAR_TRY(check_device(device));
    RuntimeGroup* g = new (std::nothrow) RuntimeGroup();
    if (!g) return SAR_ERR_OOM;
    g->device = device;
    auto fail = [&](int code) {
        bool any = false;
        for (uint32_t i = 0; i < n; ++i)
            if (out[i]) { any = true; sar_runtime_free(out[i]); out[i] = nullptr; }  // (the last one releases the group)
        if (!any) release_group(g, true);
        return code;
    };
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(SAR_ERR_HIP); }
    // what one runtime of the group will hold for frames like cfg in batches of n (the plan its first batched launch will make;
    // a need the estimate misses is served by hipMalloc as before)
    size_t dev_bytes = 0, host_bytes = 0;
    {
        sar_runtime probe;
        probe.device = device;
        probe.W = cfg->width; probe.H = cfg->height;
        probe.npix = static_cast<uint32_t>(static_cast<uint64_t>(cfg->width) * cfg->height);
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) probe.sm_count = static_cast<uint32_t>(prop.multiProcessorCount);
        group_bytes_per_runtime(cfg, &probe, n, dev_bytes, host_bytes);
    }
    g->hslab_bytes = host_bytes * n;
    if (hipHostMalloc(reinterpret_cast<void**>(&g->hslab), g->hslab_bytes, hipHostMallocDefault) != hipSuccess) { g->hslab = nullptr; g->hslab_bytes = 0; (void)hipGetLastError(); }
    for (uint32_t i = 0; i < n; ++i) {
        sar_runtime* rt = new (std::nothrow) sar_runtime();
        if (!rt) return fail(SAR_ERR_OOM);
        rt->group = g;
        { std::lock_guard<std::mutex> lock(g_group_mu); ++g->refs; }
        out[i] = rt;
        // (one allocation per runtime, not one for the group: see sar_runtime_impl.hpp; beyond 4 GiB the buffers stay separate)
        if (dev_bytes <= (4ull << 30) && hipMalloc(reinterpret_cast<void**>(&rt->sub), dev_bytes) == hipSuccess) rt->sub_bytes = dev_bytes;
        else { rt->sub = nullptr; (void)hipGetLastError(); }
        if (g->hslab) { rt->hsub = g->hslab + host_bytes * i; rt->hsub_bytes = host_bytes; }
        rt->single_hint_array = n >= 3u;  // (batches of three frames or more deal their frames to the XCDs)
        const int st = init_runtime(rt, cfg, device);
        if (st != SAR_OK) return fail(st);
    }
    if (hipStreamSynchronize(g->stream) != hipSuccess) return fail(SAR_ERR_HIP);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_free(sar_runtime* rt) try {
    if (!rt) return SAR_OK;
    hipSetDevice(rt->device);
    if (rt->stream) hipStreamSynchronize(rt->stream);
    free_device_buffers(rt);
    if (rt->d_scalars) dev_free(rt, rt->d_scalars);
    release_ln_lut(rt);
    if (rt->side) { hipStreamSynchronize(rt->side); hipStreamDestroy(rt->side); }
    if (rt->iter_done) hipEventDestroy(rt->iter_done);
    if (rt->pf_done) hipEventDestroy(rt->pf_done);
    for (hipEvent_t e : rt->img_events) if (e) hipEventDestroy(e);
    if (rt->copy_stream) hipStreamSynchronize(rt->copy_stream);  // (a borrowed one as well: a read-back may still read d_export)
    if (rt->copy_stream && rt->own_copy_stream) hipStreamDestroy(rt->copy_stream);
    if (rt->upload_stream) { hipStreamSynchronize(rt->upload_stream); hipStreamDestroy(rt->upload_stream); }
    if (rt->img_ready) hipEventDestroy(rt->img_ready);
    if (rt->starts_consumed) hipEventDestroy(rt->starts_consumed);
    if (rt->d_warm) dev_free(rt, rt->d_warm);
    if (rt->d_joblist) dev_free(rt, rt->d_joblist);
    if (rt->d_active) dev_free(rt, rt->d_active);
    if (rt->d_warm_alt) dev_free(rt, rt->d_warm_alt);
    if (rt->d_joblist_alt) dev_free(rt, rt->d_joblist_alt);
    if (rt->d_active_alt) dev_free(rt, rt->d_active_alt);
    if (rt->d_hint_range_alt) dev_free(rt, rt->d_hint_range_alt);
    if (rt->d_starts_alt) dev_free(rt, rt->d_starts_alt);
    if (rt->d_batch) dev_free(rt, rt->d_batch);
    if (rt->h_batch) host_free(rt, rt->h_batch);
    for (hipEvent_t e : rt->batch_copied) if (e) hipEventDestroy(e);
    if (rt->batch_join) hipEventDestroy(rt->batch_join);
    if (rt->d_seg_any) dev_free(rt, rt->d_seg_any);
    if (rt->h_active) host_free(rt, rt->h_active);
    if (rt->active_copied) hipEventDestroy(rt->active_copied);
    if (rt->d_starts) dev_free(rt, rt->d_starts);
    if (rt->h_starts) host_free(rt, rt->h_starts);
    if (rt->d_ckpt) dev_free(rt, rt->d_ckpt);
    if (rt->d_arena) dev_free(rt, rt->d_arena);
    if (rt->d_heads) dev_free(rt, rt->d_heads);
    if (rt->d_nan_count) dev_free(rt, rt->d_nan_count);
    if (rt->d_hint_range) dev_free(rt, rt->d_hint_range);
    if (rt->starts_copied) hipEventDestroy(rt->starts_copied);
    for (auto& s : rt->iter_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (auto& s : rt->fold_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    for (auto& s : rt->warm_spans) { hipEventDestroy(s.a); hipEventDestroy(s.b); }
    if (rt->colorize_span.a) { hipEventDestroy(rt->colorize_span.a); hipEventDestroy(rt->colorize_span.b); }
    if (rt->merge_span.a) { hipEventDestroy(rt->merge_span.a); hipEventDestroy(rt->merge_span.b); }
    if (rt->own_stream && rt->stream) hipStreamDestroy(rt->stream);
    RuntimeGroup* g = rt->group;
    if (rt->sub) hipFree(rt->sub);
    delete rt;
    if (g) release_group(g, false);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_reset(sar_runtime* rt) try {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    return do_reset(rt);
} catch (...) { return sar::abi_caught(); }

int sar_runtime_reset_batch(uint32_t n, sar_runtime* const* rts) try {
    if (n && !rts) return SAR_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i)
        if (!rts[i]) return SAR_ERR_INVALID;
    for (uint32_t first = 0; first < n;) {
        // runs of runtimes that share a device, a stream and an image size go through ONE launch
        sar_runtime* lead = rts[first];
        uint32_t m = 1;
        while (first + m < n && m < kMaxBatchFrames && rts[first + m]->device == lead->device && rts[first + m]->stream == lead->stream &&
               rts[first + m]->npix == lead->npix) ++m;
        HIP_TRY(hipSetDevice(lead->device));
        if (m == 1) {
            SAR_TRY(do_reset(lead));
        } else {
            ResetBatch t;
            std::memset(&t, 0, sizeof(t));
            for (uint32_t i = 0; i < m; ++i) {
                sar_runtime* rt = rts[first + i];
                const size_t entries = rt->d_zhint ? kHintStride(rt->npix) * rt->hint_copies_used : 0;
                t.f[i].count = rt->d_count;
                t.f[i].key = rt->d_key;
                t.f[i].steps = rt->d_steps;
                t.f[i].scalars = rt->d_scalars;
                t.f[i].hints = static_cast<uint32_t*>(rt->d_zhint);
                t.f[i].hint_words = static_cast<uint32_t>(rt->zhint_bytes == 4 ? entries : entries / 2u);
                t.f[i].hint_fill = rt->zhint_bytes == 4 ? 0xBF7FFFFFu : 0u;
                rt->hint_copies_used = 0;
                rt->hint_range_set = false;
            }
            launch_reset_batch(t, m, lead->npix, lead->stream);
            HIP_TRY(hipGetLastError());
        }
        first += m;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_set_width_height(sar_runtime* rt, uint32_t width, uint32_t height) try {
    if (!rt) return SAR_ERR_INVALID;
    if (rt->W == width && rt->H == height) return SAR_OK;  // :668
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    SAR_TRY(alloc_image_buffers(rt, width, height));
    return do_reset(rt);
} catch (...) { return sar::abi_caught(); }

int sar_runtime_seed(sar_runtime* rt, uint64_t seed) try {
    if (!rt) return SAR_ERR_INVALID;
    rt->rng.seed(seed);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_merge(sar_runtime* dst, const sar_runtime* src) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_runtime_synchronize(sar_runtime* rt) try {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    if (rt->copy_stream) HIP_TRY(hipStreamSynchronize(rt->copy_stream));  // the read-backs of async frames
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_dims(const sar_runtime* rt, uint32_t* width, uint32_t* height) try {
    if (!rt || !width || !height) return SAR_ERR_INVALID;
    *width = rt->W;
    *height = rt->H;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_set_stream(sar_runtime* rt, void* hip_stream) try {
    if (!rt) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    if (rt->copy_stream) HIP_TRY(hipStreamSynchronize(rt->copy_stream));
    if (rt->upload_stream) HIP_TRY(hipStreamSynchronize(rt->upload_stream));
    if (rt->own_stream && rt->stream) hipStreamDestroy(rt->stream);
    rt->stream = static_cast<hipStream_t>(hip_stream);
    rt->own_stream = false;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_get_copy_stream(sar_runtime* rt, void** hip_stream_out) try {
    if (!rt || !hip_stream_out) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (!rt->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&rt->copy_stream, hipStreamNonBlocking));
        rt->own_copy_stream = true;
    }
    *hip_stream_out = rt->copy_stream;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_set_copy_stream(sar_runtime* rt, void* hip_stream) try {
    if (!rt || !hip_stream) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (rt->copy_stream) HIP_TRY(hipStreamSynchronize(rt->copy_stream));
    if (rt->copy_stream && rt->own_copy_stream) hipStreamDestroy(rt->copy_stream);
    rt->copy_stream = static_cast<hipStream_t>(hip_stream);
    rt->own_copy_stream = false;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_get_stream(const sar_runtime* rt, void** hip_stream_out) try {
    if (!rt || !hip_stream_out) return SAR_ERR_INVALID;
    *hip_stream_out = rt->stream;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_colorize_device(const sar_config* cfg, sar_runtime* rt, void* rgba_out_dev) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_dev) return SAR_ERR_INVALID;
    return do_colorize(cfg, rt, rgba_out_dev);
} catch (...) { return sar::abi_caught(); }

int sar_colorize_device_batch(uint32_t n, const sar_config* const* cfgs, sar_runtime* const* rts, void* const* rgba_out_dev) try {
    if (n && (!cfgs || !rts || !rgba_out_dev)) return SAR_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        if (!cfgs[i] || !rts[i] || !rgba_out_dev[i]) return SAR_ERR_INVALID;
        SAR_TRY(check_cfg_matches(cfgs[i], rts[i]));
    }
    for (uint32_t first = 0; first < n;) {
        // runs of Gas frames of one palette, brightness and alpha rule on one device, stream and image size: ONE launch
        const sar_config* c0 = cfgs[first];
        sar_runtime* lead = rts[first];
        auto same_colours = [&](const sar_config* c) {
            return c->render_kind == SAR_RENDER_GAS && c->palette_len == c0->palette_len && c->transparent == c0->transparent &&
                   std::memcmp(&c->brightness_offset, &c0->brightness_offset, sizeof(double)) == 0 &&
                   std::memcmp(&c->brightness_factor, &c0->brightness_factor, sizeof(double)) == 0 &&
                   std::memcmp(c->palette_rgb, c0->palette_rgb, sizeof(double) * 3 * c0->palette_len) == 0;
        };
        uint32_t m = 1;
        if (c0->render_kind == SAR_RENDER_GAS && !lead->timing)
            while (first + m < n && m < kMaxBatchFrames && rts[first + m]->device == lead->device && rts[first + m]->stream == lead->stream &&
                   rts[first + m]->npix == lead->npix && !rts[first + m]->timing && same_colours(cfgs[first + m])) ++m;  // (a timed runtime records its own span)
        if (m == 1) {
            SAR_TRY(do_colorize(c0, lead, rgba_out_dev[first]));
        } else {
            HIP_TRY(hipSetDevice(lead->device));
            PaletteParams pal;
            std::memset(&pal, 0, sizeof(pal));
            pal.len = c0->palette_len;
            for (uint32_t k = 0; k < c0->palette_len; ++k)
                for (int ch = 0; ch < 3; ++ch) pal.rgb[k][ch] = c0->palette_rgb[k][ch];
            for (int ch = 0; ch < 3; ++ch) pal.rgb[c0->palette_len][ch] = c0->palette_rgb[c0->palette_len - 1][ch];  // :416-418
            ColorizeBatch t;
            std::memset(&t, 0, sizeof(t));
            for (uint32_t i = 0; i < m; ++i) {
                t.f[i].count = rts[first + i]->d_count;
                t.f[i].steps = rts[first + i]->d_steps;
                t.f[i].scalars = rts[first + i]->d_scalars;
                t.f[i].out = rgba_out_dev[first + i];
            }
            launch_colorize_gas_batch(t, m, lead->d_lnlut, kLnLutEntries, pal, c0->brightness_offset, c0->brightness_factor, c0->transparent ? 1 : 0,
                                      lead->npix, lead->stream);
            HIP_TRY(hipGetLastError());
        }
        first += m;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_colorize(const sar_config* cfg, sar_runtime* rt, uint16_t* rgba_out_host) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    SAR_TRY(ensure_rgba(rt));
    if (rt->copy_in_flight) {  // an async frame's read-back still reads d_rgba
        HIP_TRY(hipStreamWaitEvent(rt->stream, rt->img_events[(rt->img_next - 1) % 8], 0));
        rt->copy_in_flight = false;
    }
    SAR_TRY(do_colorize(cfg, rt, rt->d_rgba));
    HIP_TRY(hipMemcpyAsync(rgba_out_host, rt->d_rgba, static_cast<size_t>(rt->npix) * 8, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_extent(const sar_config* cfg, sar_runtime* rt, uint32_t n_jobs, uint64_t iters_per_job,
                       const double* starts_xyz_host, double* out12) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_image_convert_device(sar_runtime* rt, const void* rgba16_dev, int format, void* out_dev) try {
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
} catch (...) { return sar::abi_caught(); }

// colorize + the CLI's conversion into the runtime's own buffers (d_rgba, d_export), on the launch stream; remembers where the
// image lies and how long it is
static int enqueue_colorize_convert(const sar_config* cfg, sar_runtime* rt, int format) {
    SAR_TRY(check_cfg_matches(cfg, rt));
    const size_t bytes = sar_image_bytes(format, rt->W, rt->H);
    if (bytes == 0) { set_error("sar_colorize_format: bad format"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    SAR_TRY(ensure_rgba(rt));
    if (rt->copy_in_flight) {  // the last async frame's read-back still reads d_rgba / d_export
        HIP_TRY(hipStreamWaitEvent(rt->stream, rt->img_events[(rt->img_next - 1) % 8], 0));
        rt->copy_in_flight = false;
    }
    SAR_TRY(do_colorize(cfg, rt, rt->d_rgba));
    rt->export_src = rt->d_rgba;
    if (format != SAR_FMT_RGBA16) {
        if (!rt->d_export) HIP_TRY(dev_alloc(rt, &rt->d_export, static_cast<size_t>(rt->npix) * 6));  // largest converted format
        SAR_TRY(sar_image_convert_device(rt, rt->d_rgba, format, rt->d_export));
        rt->export_src = rt->d_export;
    }
    rt->export_bytes = bytes;
    return SAR_OK;
}

// the read-back of that image: on the launch stream, or — the async form — on the copy stream behind an event, so that the next
// frame's kernels do not queue behind 20-30 MB over PCIe (d_rgba / d_export are written again only behind this copy's event)
static int enqueue_read_image(sar_runtime* rt, void* out_host, bool own_copy_stream) {
    if (!out_host || !rt->export_src) { set_error("read-back: NULL output, or no colorized image to read"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    if (!own_copy_stream || rt->readback_inline) {
        HIP_TRY(hipMemcpyAsync(out_host, rt->export_src, rt->export_bytes, hipMemcpyDeviceToHost, rt->stream));
        return SAR_OK;
    }
    if (!rt->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&rt->copy_stream, hipStreamNonBlocking));
        rt->own_copy_stream = true;
    }
    if (!rt->img_ready) HIP_TRY(hipEventCreateWithFlags(&rt->img_ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(rt->img_ready, rt->stream));
    HIP_TRY(hipStreamWaitEvent(rt->copy_stream, rt->img_ready, 0));
    HIP_TRY(hipMemcpyAsync(out_host, rt->export_src, rt->export_bytes, hipMemcpyDeviceToHost, rt->copy_stream));
    return SAR_OK;
}

static int ticket_for_read(sar_runtime* rt, uint64_t* ticket_out) {
    hipEvent_t& ev = rt->img_events[rt->img_next % 8];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, rt->readback_inline ? rt->stream : rt->copy_stream));
    rt->copy_in_flight = !rt->readback_inline;
    *ticket_out = rt->img_next++;
    return SAR_OK;
}

int sar_colorize_format(const sar_config* cfg, sar_runtime* rt, int format, void* out_host) try {
    if (!out_host) { set_error("sar_colorize_format: NULL output"); return SAR_ERR_INVALID; }
    SAR_TRY(enqueue_colorize_convert(cfg, rt, format));
    SAR_TRY(enqueue_read_image(rt, out_host, false));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_colorize_format_async(const sar_config* cfg, sar_runtime* rt, int format, void* out_host, uint64_t* ticket_out) try {
    if (out_host && !ticket_out) { set_error("sar_colorize_format_async: NULL ticket"); return SAR_ERR_INVALID; }
    SAR_TRY(enqueue_colorize_convert(cfg, rt, format));
    if (!out_host) return SAR_OK;  // the image stays in device memory: sar_runtime_read_image_async fetches it
    SAR_TRY(enqueue_read_image(rt, out_host, true));
    return ticket_for_read(rt, ticket_out);
} catch (...) { return sar::abi_caught(); }

int sar_runtime_read_image_async(sar_runtime* rt, void* out_host, uint64_t* ticket_out) try {
    if (!rt || !ticket_out) { set_error("sar_runtime_read_image_async: NULL argument"); return SAR_ERR_INVALID; }
    SAR_TRY(enqueue_read_image(rt, out_host, true));
    return ticket_for_read(rt, ticket_out);
} catch (...) { return sar::abi_caught(); }

int sar_runtime_image_done(sar_runtime* rt, uint64_t ticket, int* done_out) try {
    if (!rt || !done_out || ticket >= rt->img_next) { set_error("sar_runtime_image_done: no such ticket"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    const hipError_t e = hipEventQuery(rt->img_events[ticket % 8]);
    if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(e);
    *done_out = e == hipSuccess ? 1 : 0;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_wait_image(sar_runtime* rt, uint64_t ticket) try {
    if (!rt || ticket >= rt->img_next) { set_error("sar_runtime_wait_image: no such ticket"); return SAR_ERR_INVALID; }
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipEventSynchronize(rt->img_events[ticket % 8]));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

// Page-locked host memory for the read-backs. hipHostMalloc page-locks 4 KiB page by 4 KiB page (1.0-1.4 ms per 21.6 MB image on
// most boxes of the pool, 3-4 on some: 33-100 ms of a sweep's set-up); an anonymous mapping that asks for transparent huge pages,
// touched once and then registered with the HIP runtime, is the same memory to a copy and takes 0.4 of the time (tools/ubench/
// alloc_cost.py: 352 MiB in 20 ms against 48-55). Large blocks go that way; what fails on the way falls back to hipHostMalloc.
}  // extern "C" (an unnamed namespace inside it gives its variables C names with EXTERNAL linkage: two copies of this library in
   // one process — the product and the test-suite's hooks build, loaded RTLD_GLOBAL — would share them, and construct and destroy
   // them twice)

namespace {
struct MappedBlock { void* p; size_t bytes; };
std::mutex g_mapped_mu;
std::vector<MappedBlock> g_mapped;
constexpr size_t kHugePage = 2u << 20;

size_t mapped_len(size_t bytes) { return (bytes + kHugePage - 1) & ~(kHugePage - 1); }

// An anonymous mapping of len bytes on a 2 MiB boundary, asked to be huge pages, every page touched — no HIP call in here.
char* map_and_touch(size_t len) {
    // (over-map by one huge page so that the block can start on a 2 MiB boundary: only aligned ranges get huge pages)
    char* raw = static_cast<char*>(mmap(nullptr, len + kHugePage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (raw == MAP_FAILED) return nullptr;
    char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(raw) + kHugePage - 1) & ~static_cast<uintptr_t>(kHugePage - 1));
    if (p > raw) munmap(raw, static_cast<size_t>(p - raw));
    if (p + len < raw + len + kHugePage) munmap(p + len, static_cast<size_t>(raw + len + kHugePage - (p + len)));
    madvise(p, len, MADV_HUGEPAGE);                       // (advice: refused or unavailable, the pages are small ones)
    for (size_t off = 0; off < len; off += 4096) p[off] = 0;  // first touch: the pages exist before they are locked
    return p;
}

// Blocks announced by sar_host_reserve: helper threads map and touch them ahead of their sar_host_alloc (zeroing fresh pages is
// nine tenths of what page-locking an image costs, and needs no HIP call: the helpers never contend for the HIP runtime's locks,
// which is what made page-locking itself on a helper thread slower — profiles/dead_ends.md, Round 6).
struct HostReserve {
    static constexpr int kHelpers = 3;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<char*> ready;   // mapped + touched, not registered
    size_t len = 0;            // their size (a multiple of 2 MiB)
    uint32_t pending = 0;      // blocks nobody has started on yet
    int working = 0;           // helper threads alive
    std::vector<std::thread> helpers;
    ~HostReserve() { drop(); }
    void drop() {
        std::unique_lock<std::mutex> lock(mu);
        pending = 0;
        cv.wait(lock, [&] { return working == 0; });
        for (char* p : ready) munmap(p, len);
        ready.clear();
        lock.unlock();
        for (std::thread& t : helpers)
            if (t.joinable()) t.join();
        helpers.clear();
    }
    void work(uint64_t generation_len) {
        std::unique_lock<std::mutex> lock(mu);
        while (pending) {
            --pending;
            lock.unlock();
            char* p = map_and_touch(generation_len);
            lock.lock();
            if (!p) { pending = 0; break; }
            ready.push_back(p);
            cv.notify_all();
        }
        --working;
        cv.notify_all();
    }
    // a touched block of `want` bytes, or nullptr (none announced: the caller maps its own)
    char* take(size_t want) {
        std::unique_lock<std::mutex> lock(mu);
        if (want != len) return nullptr;
        cv.wait(lock, [&] { return !ready.empty() || working == 0; });
        if (ready.empty()) return nullptr;
        char* p = ready.front();
        ready.pop_front();
        return p;
    }
} g_reserve;

void* map_and_register(size_t bytes) {
    const size_t len = mapped_len(bytes);
    char* p = g_reserve.take(len);
    if (!p) p = map_and_touch(len);
    if (!p) return nullptr;
    if (hipHostRegister(p, len, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        munmap(p, len);
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(g_mapped_mu);
    g_mapped.push_back({p, len});
    return p;
}
}  // namespace

extern "C" {

int sar_host_reserve(size_t bytes, uint32_t count) try {
    static std::mutex one_at_a_time;                    // (announcements from several threads follow each other)
    std::lock_guard<std::mutex> serial(one_at_a_time);
    g_reserve.drop();                                   // what an earlier announcement left goes first (its helpers have ended)
    if (bytes < (4u << 20) || count == 0) return SAR_OK;  // (smaller blocks are hipHostMalloc'ed: nothing to prepare)
    if (count > 4096u) { set_error("sar_host_reserve: at most 4096 blocks"); return SAR_ERR_RANGE; }
    std::lock_guard<std::mutex> lock(g_reserve.mu);
    g_reserve.len = mapped_len(bytes);
    g_reserve.pending = count;
    const size_t len = g_reserve.len;
    const int n = count < static_cast<uint32_t>(HostReserve::kHelpers) ? static_cast<int>(count) : HostReserve::kHelpers;
    g_reserve.working = n;
    for (int i = 0; i < n; ++i) g_reserve.helpers.emplace_back([len] { g_reserve.work(len); });
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_host_alloc(size_t bytes, void** out) try {
    if (!out || bytes == 0) { set_error("sar_host_alloc: NULL output or zero size"); return SAR_ERR_INVALID; }
#ifndef SAR_EXPERIMENT_HOST_ALLOC_PLAIN  // (A/B timing only)
    if (bytes >= (4u << 20)) {
        *out = map_and_register(bytes);
        if (*out) return SAR_OK;
    }
#endif
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_host_free(void* p) try {
    if (!p) return SAR_OK;
    size_t mapped = 0;
    {
        std::lock_guard<std::mutex> lock(g_mapped_mu);
        for (size_t k = 0; k < g_mapped.size(); ++k)
            if (g_mapped[k].p == p) { mapped = g_mapped[k].bytes; g_mapped.erase(g_mapped.begin() + static_cast<long>(k)); break; }
    }
    if (mapped) {
        HIP_TRY(hipHostUnregister(p));
        munmap(p, mapped);
        return SAR_OK;
    }
    HIP_TRY(hipHostFree(p));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_count(sar_runtime* rt, uint32_t* out_host) try {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_count, static_cast<size_t>(rt->npix) * 4, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_steps(sar_runtime* rt, double* out_host) try {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_steps, static_cast<size_t>(rt->npix) * 8, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_zbuf(sar_runtime* rt, float* out_host) try {
    if (!rt || !out_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (!rt->d_ztmp) HIP_TRY(dev_alloc(rt, &rt->d_ztmp, static_cast<size_t>(rt->npix) * 4));
    launch_zbuf_out(rt->d_key, rt->d_ztmp, rt->npix, rt->stream);
    HIP_TRY(hipMemcpyAsync(out_host, rt->d_ztmp, static_cast<size_t>(rt->npix) * 4, hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_max(sar_runtime* rt, uint32_t* out_max) try {
    if (!rt || !out_max) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    uint32_t sc[SC_COUNT];
    HIP_TRY(hipMemcpyAsync(sc, rt->d_scalars, sizeof(sc), hipMemcpyDeviceToHost, rt->stream));
    HIP_TRY(hipStreamSynchronize(rt->stream));
    *out_max = sc[SC_WRAP] ? 0xFFFFFFFFu : sc[SC_MAX];
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_load(sar_runtime* rt, const uint32_t* count_host, const double* steps_host,
                     const float* zbuf_host, uint32_t max) try {
    if (!rt || !count_host || !steps_host || !zbuf_host) return SAR_ERR_INVALID;
    HIP_TRY(hipSetDevice(rt->device));
    if (!rt->d_ztmp) HIP_TRY(dev_alloc(rt, &rt->d_ztmp, static_cast<size_t>(rt->npix) * 4));
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
} catch (...) { return sar::abi_caught(); }

int sar_colorize_range_device(const sar_config* cfg, sar_runtime* rt, uint32_t first_px, uint32_t n_px, void* rgba_out_dev) try {
    SAR_TRY(check_cfg_matches(cfg, rt));
    if (!rgba_out_dev) return SAR_ERR_INVALID;
    return colorize_range(cfg, rt, first_px, n_px, rgba_out_dev, true);
} catch (...) { return sar::abi_caught(); }

// ---- measurement --------------------------------------------------------------------------------------

int sar_runtime_enable_timing(sar_runtime* rt, int enabled) try {
    if (!rt) return SAR_ERR_INVALID;
    rt->timing = enabled != 0;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_runtime_last_timing(sar_runtime* rt, sar_timing* out) try {
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
} catch (...) { return sar::abi_caught(); }

int sar_runtime_set_option(sar_runtime* rt, const char* name, uint64_t value) try {
    if (!rt || !name) return SAR_ERR_INVALID;
    const uint32_t v = static_cast<uint32_t>(value);
    if (!std::strcmp(name, "block_threads")) {
        if (v == 0) { rt->block_threads = kDefaultBlock; return SAR_OK; }
        if (v % 64 || v > 256) { set_error("block_threads must be 64, 128, 192 or 256"); return SAR_ERR_INVALID; }
        rt->block_threads = v;
    } else if (!std::strcmp(name, "checkpoint_stride")) {
        rt->ckpt_stride = v ? v : kDefaultCkptStride;
    } else if (!std::strcmp(name, "hint_bits")) {
        if (v && v != 16 && v != 32) { set_error("hint_bits must be 16 or 32"); return SAR_ERR_INVALID; }
        rt->hint_bits = v;
    } else if (!std::strcmp(name, "split_waves")) {
        if (v > 2) { set_error("split_waves must be 0, 1 or 2"); return SAR_ERR_INVALID; }
        rt->split_waves = v;
    } else if (!std::strcmp(name, "timing_accumulate")) {
        rt->timing_accumulate = v != 0;
        rt->last_iterations = 0;
        rt->iter_used = 0;
        rt->fold_used = 0;
        rt->warm_used = 0;
    } else {
        set_error("unknown option '%s' (the A/B and test options live behind sar_runtime_set_test_option: include/sar_test_hooks.h)", name);
        return SAR_ERR_INVALID;
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
