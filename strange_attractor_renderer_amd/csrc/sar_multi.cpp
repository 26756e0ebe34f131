// sar_multi.cpp — ParallelRenderer / render_parallel (reference src/lib.rs:908-1082) over one OR SEVERAL GPUs, behind
// the C ABI alone (no Python, no torch.distributed, no RCCL: the exchange is point-to-point, which is exactly what
// xGMI is).
//
// The reference owns `available_parallelism()` worker threads, each with a private Runtime (:919-1004), hands them
// T*J jobs through a shared counter (:1062), folds the runtimes with Runtime::merge (:1068-1076) and colorizes (:1080).
// Here a "worker" is a whole GPU:
//   1. the T*J jobs are cut into contiguous slices, one per device; one host thread per device draws the slice's start
//      points and enqueues the render on that device's stream (no data-path traffic between devices);
//   2. every device OWNS one slice of S consecutive pixels of the image: it packs its partial buffers into one block per
//      owner (16 B/px), the owners PULL their blocks with hipMemcpyPeerAsync — G*(G-1) copies, every pair over its own
//      xGMI link, an owner's G-1 pulls on G-1 copy streams so that the links work at the same time (on ONE stream they
//      ran one after the other, each at single-link speed: round 2) — and fold them with Runtime::merge in device order
//      (k_exch_merge_slices: device 0 is the accumulator, the earlier device wins depth ties, the running max sees every
//      intermediate sum);
//   3. four scalars (max, wrap flag, depth range): every device writes its quad onto a small board in page-locked host
//      memory, waits (on its stream, through the other devices' events) until all quads are there and reduces them itself —
//      no host round trip —, colorizes ITS slice and copies it into the caller's host image over its own PCIe link (through
//      a pinned staging buffer if that image is pageable).
// The host only enqueues: the NEXT frame's start points are drawn on one helper thread per device while the GPUs work (the
// stream is addressable in blocks of 4096 jobs, sar_start_points), uploaded from page-locked memory and announced
// (sar_runtime_prefetch_device) once the exchange is enqueued.
// With one device steps 2-3 collapse to a plain colorize.
#include <chrono>
#include <functional>
#include <cstdio>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "sar_runtime_impl.hpp"

using namespace sar;

namespace {

struct Shard {
    int device = 0;
    sar_runtime* rt = nullptr;
    // sliced exchange
    void* d_pack = nullptr;   // [G][S*16] this device's partial buffers, one block per owner
    void* d_recv = nullptr;   // [G][S*16] every device's block of the slice this device owns
    void* d_rgba = nullptr;   // [S*8] colorized slice
    // sparse form: the other devices WRITE the records of their touched segments into d_recv ([G * sps] records) and their places
    // into d_slot ([G][sps], -1 = nothing sent); d_bytes counts what this device wrote to others (statistic)
    int32_t* d_slot = nullptr;
    bool coherent = false;    // d_recv and d_slot are fine-grained device memory: other devices' kernel stores are seen by this owner's fold
    unsigned long long* d_bytes = nullptr;
    uint16_t* h_rgba = nullptr;  // pinned [S*4]: the colorized slice on its way into a pageable host image
    std::vector<hipStream_t> pull_streams;  // one per source device: this owner's pulls run side by side
    std::vector<hipEvent_t> pulled;         // ... and are joined to the owner's stream through these
    size_t slice_cap = 0;     // S the buffers were sized for
    hipEvent_t packed = nullptr, merged = nullptr, reduced = nullptr, begin = nullptr, end = nullptr;
    // job slice of the current frame
    uint32_t first_job = 0, n_jobs = 0;
    // the NEXT frame's slice of start points, uploaded and announced (sar_runtime_prefetch_device) while this frame renders:
    // its warm-up then runs under this frame's accumulate / fold / colorize
    // (two buffers in turn: the frame in flight still reads the one announced a frame ago — the start points of its later
    // launch chunks are converted on its own stream — while the next frame's points are uploaded)
    double* d_next_buf[2] = {nullptr, nullptr};
    size_t next_cap[2] = {0, 0};  // jobs
    uint32_t next_slot = 0;       // the buffer the NEXT frame's points are drawn into / uploaded to
    uint32_t cur_slot = 0;        // the buffer that holds THIS frame's points (host side)
    hipEvent_t next_read[2] = {nullptr, nullptr};  // recorded behind the frame that read buffer [i] (a caller that passes no
    bool next_read_rec[2] = {false, false};        // host image is not waited for by sar_render_parallel itself)
    double* d_next = nullptr;     // the buffer that holds the announced points
    double* h_next[2] = {nullptr, nullptr};  // page-locked: the slice's points as the helper thread drew them (upload source)
    size_t h_next_cap[2] = {0, 0};           // jobs
    hipEvent_t uploaded = nullptr;           // the upload of the announced points (the announced warm-up waits for it)
    hipStream_t up = nullptr;
    bool next_valid = false;
    uint32_t next_first = 0, next_n = 0;
    uint64_t next_iters = 0;
    int status = SAR_OK;
    char error[512] = {0};
};

}  // namespace

struct sar_renderer {
    uint32_t units = 0;
    uint64_t seed = 0;
    Rng rng;                      // the renderer's start-point stream, at the first job of the next frame to render
    // The NEXT frame's start points are drawn while the GPUs work on the current one, every device's slice on its own helper
    // thread into page-locked memory (Shard::h_next): `ahead_jobs` jobs from `rng_next` on (the stream behind the current
    // frame) are there. A next frame with another job count simply does not use them.
    uint64_t ahead_jobs = 0;
    Rng rng_next;
    long long* h_board = nullptr; // page-locked, visible to every device: [64][4] scalar quads of the exchange (step 3)
    long long* d_board = nullptr; // ... as the devices address it
    std::vector<Shard> shards;    // one per device, in fold order
    bool scattered = false;       // shard runtimes hold only their own merged slice (gather before handing one out)
    uint32_t exchange_mode = 0;   // 0: sparse (kernels push the touched segments' records over xGMI) when every pair of devices has peer
                                  // access, else dense; 1: dense (whole slices by hipMemcpyPeerAsync); 2: sparse
    uint32_t peer_access_failures = 0;  // ordered device pairs whose copies cannot go peer to peer
    sar_parallel_timing timing{};
};

namespace {

uint32_t slice_pixels(uint32_t npix, uint32_t world) {  // == sar_exchange_slice_pixels: whole 2048-pixel segments
    const uint64_t s = (static_cast<uint64_t>(npix) + world - 1) / world;
    return static_cast<uint32_t>((s + (kExchSliceAlign - 1u)) & ~static_cast<uint64_t>(kExchSliceAlign - 1u));
}

int free_shard_buffers(Shard& sh) {
    hipSetDevice(sh.device);
    if (sh.d_pack) hipFree(sh.d_pack);
    if (sh.d_recv) hipFree(sh.d_recv);
    if (sh.d_rgba) hipFree(sh.d_rgba);
    if (sh.h_rgba) hipHostFree(sh.h_rgba);
    sh.h_rgba = nullptr;
    if (sh.d_slot) hipFree(sh.d_slot);
    if (sh.d_bytes) hipFree(sh.d_bytes);
    sh.d_slot = nullptr;
    sh.d_bytes = nullptr;
    sh.d_pack = sh.d_recv = sh.d_rgba = nullptr;
    sh.slice_cap = 0;
    return SAR_OK;
}

int ensure_shard(sar_renderer* r, Shard& sh, const sar_config* cfg, uint32_t S) {
    const uint32_t G = static_cast<uint32_t>(r->shards.size());
    if (!sh.rt) {
        sar_config c0 = *cfg;
        c0.seed = r->seed;
        SAR_TRY(sar_runtime_new(&c0, sh.device, &sh.rt));
    }
    SAR_TRY(sar_runtime_set_width_height(sh.rt, cfg->width, cfg->height));  // :950
    if (G == 1) return SAR_OK;
    HIP_TRY(hipSetDevice(sh.device));
    if (!sh.packed) {
        HIP_TRY(hipEventCreate(&sh.packed));
        HIP_TRY(hipEventCreate(&sh.merged));
        HIP_TRY(hipEventCreate(&sh.reduced));
        HIP_TRY(hipEventCreate(&sh.begin));
        HIP_TRY(hipEventCreate(&sh.end));
        sh.pull_streams.assign(G, nullptr);
        sh.pulled.assign(G, nullptr);
        for (uint32_t k = 0; k < G; ++k) {
            HIP_TRY(hipStreamCreateWithFlags(&sh.pull_streams[k], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sh.pulled[k], hipEventDisableTiming));
        }
    }
    if (sh.slice_cap != S) {
        HIP_TRY(hipStreamSynchronize(sh.rt->stream));
        free_shard_buffers(sh);
        HIP_TRY(hipMalloc(&sh.d_pack, static_cast<size_t>(G) * S * 16u));
        // What the OTHER devices' kernels store into (sparse exchange): fine-grained device memory — coherent between devices, no
        // stale line of the previous frame in this device's L2 when the owner folds (plain device memory is only guaranteed
        // coherent at kernel boundaries for its own device). A device that cannot provide it gets plain memory — and the renderer
        // then exchanges the dense way (hipMemcpyPeerAsync, no peer stores): see `coherent`.
        sh.coherent = true;
        if (hipExtMallocWithFlags(&sh.d_recv, static_cast<size_t>(G) * S * 16u, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            sh.d_recv = nullptr;
            sh.coherent = false;
            HIP_TRY(hipMalloc(&sh.d_recv, static_cast<size_t>(G) * S * 16u));
        }
        HIP_TRY(hipMalloc(&sh.d_rgba, static_cast<size_t>(S) * 8u));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sh.h_rgba), static_cast<size_t>(S) * 8u, hipHostMallocDefault));
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&sh.d_slot), static_cast<size_t>(G) * (S / kExchSeg) * sizeof(int32_t),
                                  hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            sh.d_slot = nullptr;
            sh.coherent = false;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sh.d_slot), static_cast<size_t>(G) * (S / kExchSeg) * sizeof(int32_t)));
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sh.d_bytes), sizeof(unsigned long long)));
        HIP_TRY(hipMemset(sh.d_bytes, 0, sizeof(unsigned long long)));
        sh.slice_cap = S;
    }
    return SAR_OK;
}

// what one reference worker thread does with its share of the jobs (:950-988), for a whole GPU
void render_shard(sar_renderer* r, Shard* sh, const sar_config* cfg, uint64_t per_job, uint32_t S, bool use_next, bool sparse) {
    const uint32_t G = static_cast<uint32_t>(r->shards.size());
    auto run = [&]() -> int {
        HIP_TRY(hipSetDevice(sh->device));
        sar_runtime* rt = sh->rt;
        if (G > 1) HIP_TRY(hipEventRecord(sh->begin, rt->stream));
        SAR_TRY(sar_runtime_reset(rt));  // :951
        if (use_next && sh->next_valid && sh->next_first == sh->first_job && sh->next_n == sh->n_jobs && sh->next_iters == per_job) {
            HIP_TRY(hipStreamWaitEvent(rt->stream, sh->uploaded, 0));  // (long done: the announced warm-up waited for it too)
            SAR_TRY(render_chunked(cfg, rt, sh->n_jobs, per_job, sh->d_next, true));  // the points uploaded during the previous frame
            const uint32_t slot = sh->d_next == sh->d_next_buf[0] ? 0u : 1u;
            if (!sh->next_read[slot]) HIP_TRY(hipEventCreateWithFlags(&sh->next_read[slot], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(sh->next_read[slot], rt->stream));
            sh->next_read_rec[slot] = true;
        } else {
            SAR_TRY(render_chunked(cfg, rt, sh->n_jobs, per_job, sh->h_next[sh->cur_slot]));  // this slice's points as drawn (page-locked host memory)
        }
        sh->next_valid = false;
        if (G > 1 && sparse) {
            // the records of the segments this device touched go straight into their owners' buffers (peer memory)
            ExchPushArgs pa;
            std::memset(&pa, 0, sizeof(pa));
            pa.count = rt->d_count;
            pa.key = rt->d_key;
            pa.steps = rt->d_steps;
            pa.npix = rt->npix;
            pa.nseg = (rt->npix + kExchSeg - 1u) / kExchSeg;
            pa.sps = S / kExchSeg;
            pa.src = static_cast<uint32_t>(sh - r->shards.data());
            pa.G = G;
            pa.bytes = sh->d_bytes;
            for (uint32_t o = 0; o < G; ++o) {
                pa.recv[o] = static_cast<unsigned char*>(r->shards[o].d_recv);
                pa.slot[o] = r->shards[o].d_slot;
            }
            HIP_TRY(hipMemsetAsync(sh->d_bytes, 0, sizeof(unsigned long long), rt->stream));
            launch_exch_push(pa, rt->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(sh->packed, rt->stream));
        } else if (G > 1) {
            launch_exch_pack(rt->d_count, rt->d_key, rt->d_steps, rt->npix, S, G, sh->d_pack, rt->stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(sh->packed, rt->stream));
        }
        return SAR_OK;
    };
    try {
        sh->status = run();
    } catch (...) {  // (a worker thread: an exception that left it would end the process)
        sh->status = abi_caught();
    }
    if (sh->status != SAR_OK) std::snprintf(sh->error, sizeof(sh->error), "%s", sar_last_error());
}

// Makes shard 0's runtime hold the whole merged frame (every owner's slice copied over): what a caller that asks for
// "the renderer's runtime" expects to read.
int gather_into_first(sar_renderer* r) {
    if (!r->scattered) return SAR_OK;
    const uint32_t G = static_cast<uint32_t>(r->shards.size());
    Shard& s0 = r->shards[0];
    const uint32_t npix = s0.rt->npix;
    const uint32_t S = static_cast<uint32_t>(s0.slice_cap);
    HIP_TRY(hipSetDevice(s0.device));
    for (uint32_t d = 1; d < G; ++d) {
        const Shard& sd = r->shards[d];
        const uint64_t first = static_cast<uint64_t>(d) * S;
        if (first >= npix) break;
        const size_t n = (npix - first < S) ? npix - first : S;
        HIP_TRY(hipMemcpyPeerAsync(s0.rt->d_count + first, s0.device, sd.rt->d_count + first, sd.device, n * 4u, s0.rt->stream));
        HIP_TRY(hipMemcpyPeerAsync(s0.rt->d_key + first, s0.device, sd.rt->d_key + first, sd.device, n * 8u, s0.rt->stream));
        HIP_TRY(hipMemcpyPeerAsync(s0.rt->d_steps + first, s0.device, sd.rt->d_steps + first, sd.device, n * 8u, s0.rt->stream));
    }
    HIP_TRY(hipStreamSynchronize(s0.rt->stream));
    r->scattered = false;
    return SAR_OK;
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" {

int sar_renderer_new_multi(const int* devices, uint32_t n_devices, uint32_t units, uint64_t seed, sar_renderer** out) try {
    if (!out) return SAR_ERR_INVALID;
    *out = nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) { set_error("sar_renderer_new_multi: 1..64 devices"); return SAR_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (this library has no CPU fallback)");
        return SAR_ERR_NO_DEVICE;
    }
    for (uint32_t k = 0; k < n_devices; ++k)
        if (devices[k] < 0 || devices[k] >= ndev) { set_error("device %d out of range (%d devices)", devices[k], ndev); return SAR_ERR_INVALID; }
    sar_renderer* r = new (std::nothrow) sar_renderer();
    if (!r) return SAR_ERR_OOM;
    r->seed = seed;
    r->rng.seed(seed);
    r->shards.resize(n_devices);
    uint64_t lanes = 0;
    for (uint32_t k = 0; k < n_devices; ++k) {
        r->shards[k].device = devices[k];
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, devices[k]) != hipSuccess) { delete r; return SAR_ERR_HIP; }
        lanes += static_cast<uint64_t>(prop.multiProcessorCount) * 64u;
    }
    // the role available_parallelism() plays at src/lib.rs:920-922. 64 units per CU and device (16 384 per MI355X): with
    // the CLI's default of 12 jobs per thread (src/bin/main.rs:305) the job split then gives 196 608 trajectories per
    // GPU — three waves per SIMD — and 8 jobs per unit give 131 072; every job pays 1000 warm-up iterations, so a unit
    // count that multiplied typical jobs_per_unit values into millions of jobs would only add warm-up work
    r->units = units ? units : static_cast<uint32_t>(lanes > 0xFFFFFFFFull ? 0xFFFFFFFFull : lanes);
    // direct xGMI copies between every pair of distinct devices (hipMemcpyPeerAsync works without peer access too, but
    // then stages through host memory); "already enabled" is not an error
    for (uint32_t a = 0; a < n_devices; ++a)
        for (uint32_t b = 0; b < n_devices; ++b) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, devices[a], devices[b]);
            if (e == hipSuccess && can) e = hipSetDevice(devices[a]);
            if (e == hipSuccess && can) {
                e = hipDeviceEnablePeerAccess(devices[b], 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            }
            if (e != hipSuccess || !can) {  // not fatal (the copies are then staged through the host), but not silent either
                (void)hipGetLastError();
                ++r->peer_access_failures;
                set_error("no direct peer access from device %d to device %d (%s): their exchange is staged through host memory",
                          devices[a], devices[b], e != hipSuccess ? hipGetErrorString(e) : "hipDeviceCanAccessPeer: no");
            }
        }
    *out = r;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_new(int device, uint32_t units, uint64_t seed, sar_renderer** out) try {
    return sar_renderer_new_multi(&device, 1, units, seed, out);
} catch (...) { return sar::abi_caught(); }

int sar_renderer_num_units(const sar_renderer* r, uint32_t* out_units) try {
    if (!r || !out_units) return SAR_ERR_INVALID;
    *out_units = r->units;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_num_devices(const sar_renderer* r, uint32_t* out_devices) try {
    if (!r || !out_devices) return SAR_ERR_INVALID;
    *out_devices = static_cast<uint32_t>(r->shards.size());
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_shutdown(sar_renderer* r) try {
    if (!r) return SAR_OK;
    for (Shard& sh : r->shards) {
        hipSetDevice(sh.device);
        if (sh.rt) hipStreamSynchronize(sh.rt->stream);
        if (sh.up) hipStreamSynchronize(sh.up);
    }
    for (Shard& sh : r->shards) {
        if (sh.rt) sar_runtime_free(sh.rt);  // first: an announced warm-up may still read d_next on the runtime's side stream
        sh.rt = nullptr;
        free_shard_buffers(sh);
        for (double* q : sh.d_next_buf) if (q) hipFree(q);
        for (double* q : sh.h_next) if (q) hipHostFree(q);
        if (sh.uploaded) hipEventDestroy(sh.uploaded);
        for (hipEvent_t ev : sh.next_read) if (ev) hipEventDestroy(ev);
        if (sh.up) hipStreamDestroy(sh.up);
        for (hipStream_t st : sh.pull_streams) if (st) hipStreamDestroy(st);
        for (hipEvent_t ev : sh.pulled) if (ev) hipEventDestroy(ev);
        if (sh.packed) hipEventDestroy(sh.packed);
        if (sh.merged) hipEventDestroy(sh.merged);
        if (sh.reduced) hipEventDestroy(sh.reduced);
        if (sh.begin) hipEventDestroy(sh.begin);
        if (sh.end) hipEventDestroy(sh.end);
    }
    if (r->h_board) hipHostFree(r->h_board);
    delete r;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_runtime(sar_renderer* r, sar_runtime** out_borrowed) try {
    if (!r || !out_borrowed) return SAR_ERR_INVALID;
    *out_borrowed = nullptr;
    if (r->shards.empty() || !r->shards[0].rt) { set_error("the renderer has not rendered yet"); return SAR_ERR_INVALID; }
    SAR_TRY(gather_into_first(r));
    *out_borrowed = r->shards[0].rt;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_set_exchange(sar_renderer* r, uint32_t mode) try {
    if (!r || mode > 2) { set_error("sar_renderer_set_exchange: mode must be 0 (automatic), 1 (dense) or 2 (sparse)"); return SAR_ERR_INVALID; }
    r->exchange_mode = mode;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_renderer_last_timing(const sar_renderer* r, sar_parallel_timing* out) try {
    if (!r || !out) return SAR_ERR_INVALID;
    *out = r->timing;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_render_parallel(sar_renderer* r, const sar_config* cfg, uint32_t jobs_per_unit, uint16_t* rgba_out_host) try {
    if (!r) return SAR_ERR_INVALID;
    SAR_TRY(validate(cfg));
    if (jobs_per_unit == 0) { set_error("jobs_per_unit is 0"); return SAR_ERR_INVALID; }
    const uint64_t total_jobs = static_cast<uint64_t>(r->units) * jobs_per_unit;  // :1062
    if (total_jobs > 0xFFFFFFFFull) { set_error("units*jobs_per_unit exceeds 2^32-1"); return SAR_ERR_RANGE; }
    const uint64_t per_job = cfg->iterations / r->units / jobs_per_unit;  // :1058
    const uint32_t G = static_cast<uint32_t>(r->shards.size());
    const uint64_t npix64 = static_cast<uint64_t>(cfg->width) * cfg->height;
    if (npix64 > 0x7fffffffull) { set_error("width*height exceeds 2^31-1"); return SAR_ERR_RANGE; }
    const uint32_t npix = static_cast<uint32_t>(npix64);
    const uint32_t S = slice_pixels(npix, G);
    const double t0 = now_ms();
    std::memset(&r->timing, 0, sizeof(r->timing));
    r->timing.n_devices = G;
    r->scattered = false;

    for (Shard& sh : r->shards) SAR_TRY(ensure_shard(r, sh, cfg, S));
    if (G > 1 && !r->h_board) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&r->h_board), 64 * 4 * sizeof(long long), hipHostMallocPortable | hipHostMallocMapped));
        void* dev_view = nullptr;  // what the kernels use (the same address under unified addressing; asked for, not assumed)
        HIP_TRY(hipHostGetDevicePointer(&dev_view, r->h_board, 0));
        r->d_board = static_cast<long long*>(dev_view);
    }

    // contiguous job slices, sizes differ by at most one (the same partition as distributed.shard_jobs)
    {
        const uint64_t base = total_jobs / G, rem = total_jobs % G;
        uint64_t first = 0;
        for (uint32_t d = 0; d < G; ++d) {
            r->shards[d].first_job = static_cast<uint32_t>(first);
            r->shards[d].n_jobs = static_cast<uint32_t>(base + (d < rem ? 1u : 0u));
            first += r->shards[d].n_jobs;
        }
    }

    // Fresh start points for every job, in job order, from the renderer's stream (the reference's workers draw from
    // per-thread RNGs as they pick jobs up, :748; here the stream is one and the job -> point map is deterministic). Every
    // device's slice is drawn on its own host thread into page-locked memory: the stream is addressable in blocks of 4096 jobs.
    const Rng frame_rng = r->rng;                                  // the stream at this frame's first job
    const bool from_ahead = r->ahead_jobs == total_jobs;           // drawn (and, where possible, uploaded + announced) during the previous frame
    r->ahead_jobs = 0;
    auto pinned_slice = [](Shard& sh, uint32_t slot, uint32_t jobs) -> int {   // on the shard's device
        if (jobs <= sh.h_next_cap[slot]) return SAR_OK;
        if (sh.uploaded) HIP_TRY(hipEventSynchronize(sh.uploaded));            // the last upload out of this memory
        if (sh.h_next[slot]) hipHostFree(sh.h_next[slot]);
        sh.h_next[slot] = nullptr;
        sh.h_next_cap[slot] = 0;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&sh.h_next[slot]), static_cast<size_t>(jobs) * 3 * sizeof(double), hipHostMallocDefault));
        sh.h_next_cap[slot] = jobs;
        return SAR_OK;
    };
    auto draw_slice = [](const Rng& from, uint64_t skip, uint32_t jobs, double* out) {
        Rng g = from;
        g.skip_points(skip);
        for (uint32_t k = 0; k < jobs; ++k) g.start_point(out + 3 * static_cast<size_t>(k));
    };
    if (!from_ahead) {  // the first frame, or another job count than the previous one: draw now, all slices at the same time
        std::vector<std::thread> drawers;
        for (uint32_t d = 0; d < G; ++d) {
            Shard& sh = r->shards[d];
            HIP_TRY(hipSetDevice(sh.device));
            sh.next_valid = false;
            sh.cur_slot = sh.next_slot;
            if (sh.uploaded) HIP_TRY(hipEventSynchronize(sh.uploaded));  // an earlier frame's upload out of this memory
            SAR_TRY(pinned_slice(sh, sh.cur_slot, sh.n_jobs));
            sh.next_slot ^= 1u;
        }
        for (uint32_t d = 0; d < G; ++d) {
            Shard& sh = r->shards[d];
            try {
                drawers.emplace_back(draw_slice, std::cref(frame_rng), static_cast<uint64_t>(sh.first_job), sh.n_jobs, sh.h_next[sh.cur_slot]);
            } catch (const std::system_error&) {  // no thread to be had: this slice is drawn here (nothing unwinds across the ABI)
                draw_slice(frame_rng, static_cast<uint64_t>(sh.first_job), sh.n_jobs, sh.h_next[sh.cur_slot]);
            }
        }
        for (auto& t : drawers) t.join();
    }
    // The NEXT frame's points (same job count assumed: a sweep, a sequence): drawn from now on by one helper thread per
    // device while this thread enqueues the frame; joined once the exchange is enqueued.
    std::vector<std::thread> helpers;
    double draw_ms = 0.0;
    bool ahead_ok = per_job != 0;
    if (ahead_ok) {
        for (uint32_t d = 0; d < G && ahead_ok; ++d) {
            Shard& sh = r->shards[d];
            ahead_ok = hipSetDevice(sh.device) == hipSuccess && pinned_slice(sh, sh.next_slot, sh.n_jobs) == SAR_OK &&
                       (!sh.uploaded || hipEventSynchronize(sh.uploaded) == hipSuccess);  // nothing reads that memory any more
        }
        if (!ahead_ok) (void)hipGetLastError();
    }
    if (ahead_ok) {
        const double td = now_ms();
        for (uint32_t d = 0; d < G; ++d) {
            Shard* sh = &r->shards[d];
            try {
                helpers.emplace_back([=, &frame_rng, &draw_ms]() {
                    Rng nx = frame_rng;
                    nx.skip_points(total_jobs);              // the stream behind this frame
                    if (d == 0) r->rng_next = nx;
                    nx.skip_points(sh->first_job);
                    double* out = sh->h_next[sh->next_slot];
                    for (uint32_t k = 0; k < sh->n_jobs; ++k) nx.start_point(out + 3 * static_cast<size_t>(k));
                    if (d == 0) draw_ms = now_ms() - td;     // (device 0's slice: they are equal)
                });
            } catch (const std::system_error&) {  // no thread to be had: nothing is drawn ahead this frame (the next one draws its own)
                for (auto& t : helpers) t.join();
                helpers.clear();
                ahead_ok = false;
                break;
            }
        }
    }
    auto join_helpers = [&]() {
        for (auto& t : helpers) t.join();
        helpers.clear();
    };
    // ... and once they are drawn: every device gets its slice of them and is told (sar_runtime_prefetch_device), so that
    // the next frame's 1000 warm-up iterations per job run under THIS frame's accumulate / fold / colorize. The warm-up is
    // the map alone: the next frame may turn the view (a sweep does). Best effort — a failure here only costs the overlap.
    // Nothing here waits for the GPU: the upload comes out of page-locked memory and the announced warm-up waits for it on
    // its own stream.
    auto announce_next = [&]() {
        for (uint32_t d = 0; d < G; ++d) {
            Shard& sh = r->shards[d];
            const uint32_t slot = sh.next_slot;
            sh.next_valid = false;
            // a device listed several times (tests; a box with fewer GPUs than shards) runs one warm-up ahead, its first
            // shard's: the chip is busy with the other shards' frames anyway, and eight warm-ups piled onto one GPU only delay
            // the exchange they run under
            bool first_on_device = true;
            for (uint32_t e = 0; e < d; ++e) first_on_device = first_on_device && r->shards[e].device != sh.device;
            if (!first_on_device) continue;
            const uint32_t nj = sh.n_jobs;
            if (nj == 0 || per_job == 0 || hipSetDevice(sh.device) != hipSuccess) continue;
            bool ok = true;
            if (!sh.up) ok = hipStreamCreateWithFlags(&sh.up, hipStreamNonBlocking) == hipSuccess;
            if (ok && !sh.uploaded) ok = hipEventCreateWithFlags(&sh.uploaded, hipEventDisableTiming) == hipSuccess;
            if (ok && sh.next_read_rec[slot]) {  // the frame that read this buffer: two frames back, done unless nobody waited
                ok = hipEventSynchronize(sh.next_read[slot]) == hipSuccess;
                sh.next_read_rec[slot] = false;
            }
            if (ok && nj > sh.next_cap[slot]) {
                if (sh.d_next_buf[slot]) hipFree(sh.d_next_buf[slot]);  // last read two frames ago
                sh.d_next_buf[slot] = nullptr;
                sh.next_cap[slot] = 0;
                ok = hipMalloc(&sh.d_next_buf[slot], static_cast<size_t>(nj) * 3 * sizeof(double)) == hipSuccess;
                if (ok) sh.next_cap[slot] = nj;
            }
            sh.d_next = sh.d_next_buf[slot];
            // an announced warm-up that nobody consumed (another job count, a failed frame) may still read this buffer on the
            // runtime's side stream: the upload goes behind it (an event never recorded waits for nothing)
            if (ok && sh.rt->pf_done) ok = hipStreamWaitEvent(sh.up, sh.rt->pf_done, 0) == hipSuccess;
            ok = ok && hipMemcpyAsync(sh.d_next, sh.h_next[slot], static_cast<size_t>(nj) * 3 * sizeof(double), hipMemcpyHostToDevice, sh.up) == hipSuccess &&
                 hipEventRecord(sh.uploaded, sh.up) == hipSuccess;
            if (ok) {
                sh.rt->prefetch_after = sh.uploaded;  // the announced warm-up's stream waits for the upload; this thread does not
                ok = sar_runtime_prefetch_device(cfg, sh.rt, nj, per_job, sh.d_next) == SAR_OK;
                sh.rt->prefetch_after = nullptr;
            }
            if (!ok) { (void)hipGetLastError(); continue; }
            sh.next_valid = true;
            sh.next_first = sh.first_job;
            sh.next_n = nj;
            sh.next_iters = per_job;
        }
    };
    // the frame is under way: the stream moves behind it, and the points drawn meanwhile belong to the next one
    auto commit_ahead = [&]() {
        const bool had = !helpers.empty();
        join_helpers();
        r->timing.draw_ahead_ms = static_cast<float>(draw_ms);
        if (had) {
            r->rng = r->rng_next;
            announce_next();
            for (Shard& sh : r->shards) {
                sh.cur_slot = sh.next_slot;
                sh.next_slot ^= 1u;
            }
            r->ahead_jobs = total_jobs;
        } else {
            r->rng.skip_points(total_jobs);
        }
    };
    // A frame that fails leaves the renderer as it found it: the start-point stream stays at this frame's first job (the
    // next frame draws the points this one would have used), nothing drawn ahead survives, every device has finished what it
    // was given, and no shard claims to hold merged slices.
    auto failed = [&](int status) {
        char keep[512];
        std::snprintf(keep, sizeof(keep), "%s", sar_last_error());
        join_helpers();
        for (Shard& sh : r->shards) {
            if (!sh.rt) continue;
            hipSetDevice(sh.device);
            hipStreamSynchronize(sh.rt->stream);
            for (hipStream_t st : sh.pull_streams) if (st) hipStreamSynchronize(st);
        }
        (void)hipGetLastError();
        r->rng = frame_rng;
        r->ahead_jobs = 0;
        for (Shard& sh : r->shards) sh.next_valid = false;
        r->scattered = false;
        set_error("%s", keep);
        return status;
    };

    // Sparse = every device's kernels STORE into the owners' buffers: that needs direct peer access between every pair and
    // fine-grained (device-coherent) buffers on every owner — with plain memory a stale line of the previous frame in the owner's L2
    // could reach its fold. Where either is missing the automatic mode goes dense, and a sparse exchange asked for is an error.
    bool peer_stores_ok = r->peer_access_failures == 0;
    for (const Shard& sh : r->shards) peer_stores_ok = peer_stores_ok && sh.coherent;
    if (G > 1 && r->exchange_mode == 2u && !peer_stores_ok) {
        set_error("sparse exchange asked for (sar_renderer_set_exchange 2), but %s", r->peer_access_failures
                  ? "some pair of devices has no direct peer access" : "a device could not provide fine-grained memory for its receive buffers");
        return failed(SAR_ERR_INVALID);
    }
    const bool sparse = G > 1 && G <= kMaxExchDevices && peer_stores_ok && r->exchange_mode != 1u;
    if (G == 1) {
        Shard& sh = r->shards[0];
        render_shard(r, &sh, cfg, per_job, S, from_ahead, false);
        if (sh.status != SAR_OK) { set_error("%s", sh.error); return failed(sh.status); }
        r->timing.host_ms_before_exchange = static_cast<float>(now_ms() - t0);
        // the next frame's points (the helper has been drawing them since before the render was enqueued) go to the device and
        // are announced BEFORE this thread waits for the image: the announced warm-up then runs under this frame's tail
        commit_ahead();
        r->timing.host_ms_enqueue = static_cast<float>(now_ms() - t0);
        if (rgba_out_host) {
            const int st = sar_colorize(cfg, sh.rt, rgba_out_host);  // :1080 (waits for the image)
            if (st != SAR_OK) return failed(st);
        }
        r->timing.total_ms = static_cast<float>(now_ms() - t0);
        return SAR_OK;
    }

    // 1. one host thread per device (as the reference has one per core): stage, render, pack
    {
        std::vector<std::thread> workers;
        workers.reserve(G);
        for (uint32_t d = 0; d < G; ++d) {
            try {
                workers.emplace_back(render_shard, r, &r->shards[d], cfg, per_job, S, from_ahead, sparse);
            } catch (const std::system_error&) {  // no thread to be had: this device's frame is enqueued from here
                render_shard(r, &r->shards[d], cfg, per_job, S, from_ahead, sparse);
            }
        }
        for (auto& w : workers) w.join();
    }
    for (Shard& sh : r->shards)
        if (sh.status != SAR_OK) { set_error("device %d: %s", sh.device, sh.error); return failed(sh.status); }
    const double t_rendered = now_ms();

    const size_t blk = static_cast<size_t>(S) * 16u;
    // Is the caller's image pinned memory (then every device copies its slice straight into it), or pageable (an async copy
    // into pageable memory is staged by the HIP runtime and serialises the devices: each goes through its own pinned buffer
    // and a host thread moves the slice on)?
    bool pinned_out = false;
    if (rgba_out_host) {
        hipPointerAttribute_t attr;
        std::memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, rgba_out_host) == hipSuccess) pinned_out = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();
    }
    auto exchange = [&]() -> int {
        // 2. the owners pull their blocks (every pair of devices over its own link, all links at the same time) and fold them
        // in device order; every device then posts its slice's four scalars on the board
        for (uint32_t d = 0; d < G; ++d) {
            Shard& dst = r->shards[d];
            HIP_TRY(hipSetDevice(dst.device));
            hipStream_t st = dst.rt->stream;
            const uint64_t first_px = static_cast<uint64_t>(d) * S;
            const uint32_t n_px = first_px >= npix ? 0u : static_cast<uint32_t>((npix - first_px < S) ? npix - first_px : S);
            if (sparse) {
                // the records are there once every device's push kernel has ended (its own among them)
                for (uint32_t s = 0; s < G; ++s)
                    if (s != d) HIP_TRY(hipStreamWaitEvent(st, r->shards[s].packed, 0));
                if (d == 0) r->timing.host_ms_before_exchange = static_cast<float>(now_ms() - t_rendered);
                launch_exch_merge_sparse(dst.rt->d_count, dst.rt->d_key, dst.rt->d_steps, static_cast<uint32_t>(first_px >= npix ? 0 : first_px), n_px,
                                         S / kExchSeg, G, dst.d_recv, dst.d_slot, dst.rt->d_scalars, d == 0, st);
                launch_exch_scalars_export(dst.rt->d_scalars, r->d_board + 4 * d, st);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(dst.merged, st));
                continue;
            }
            for (uint32_t k = 0; k < G; ++k) {
                const uint32_t s = (d + k) % G;  // start with the local block; stagger the sources over the links
                Shard& src = r->shards[s];
                hipStream_t cs = s == d ? st : dst.pull_streams[s];
                if (s != d) HIP_TRY(hipStreamWaitEvent(cs, src.packed, 0));
                if (d == 0 && k == 0) r->timing.host_ms_before_exchange = static_cast<float>(now_ms() - t_rendered);
                HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(dst.d_recv) + s * blk, dst.device,
                                           static_cast<const char*>(src.d_pack) + d * blk, src.device, blk, cs));
                if (s != d) {
                    HIP_TRY(hipEventRecord(dst.pulled[s], cs));
                    HIP_TRY(hipStreamWaitEvent(st, dst.pulled[s], 0));
                }
            }
            const uint64_t first = static_cast<uint64_t>(d) * S;
            const uint32_t n = first >= npix ? 0u : static_cast<uint32_t>((npix - first < S) ? npix - first : S);
            launch_exch_merge_slices(dst.rt->d_count, dst.rt->d_key, dst.rt->d_steps, static_cast<uint32_t>(first >= npix ? 0 : first), n, S, G,
                                     dst.d_recv, dst.rt->d_scalars, d == 0, st);
            launch_exch_scalars_export(dst.rt->d_scalars, r->d_board + 4 * d, st);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(dst.merged, st));
        }
        r->scattered = true;

        // 3. every device waits — on its stream — for the other devices' quads, reduces the G of them itself (no host round
        // trip), colorizes its slice and copies it into the caller's image
        for (uint32_t d = 0; d < G; ++d) {
            Shard& sh = r->shards[d];
            HIP_TRY(hipSetDevice(sh.device));
            hipStream_t st = sh.rt->stream;
            for (uint32_t e = 0; e < G; ++e)
                if (e != d) HIP_TRY(hipStreamWaitEvent(st, r->shards[e].merged, 0));
            launch_exch_scalars_reduce(sh.rt->d_scalars, r->d_board, G, st);
            HIP_TRY(hipEventRecord(sh.reduced, st));  // (from here on the device works on its own slice again)
            const uint64_t first = static_cast<uint64_t>(d) * S;
            const uint32_t n = first >= npix ? 0u : static_cast<uint32_t>((npix - first < S) ? npix - first : S);
            if (rgba_out_host && n) {
                SAR_TRY(colorize_range(cfg, sh.rt, static_cast<uint32_t>(first), n, sh.d_rgba, true));  // :1080, sharded
                HIP_TRY(hipMemcpyAsync(pinned_out ? rgba_out_host + first * 4u : sh.h_rgba, sh.d_rgba, static_cast<size_t>(n) * 8u,
                                       hipMemcpyDeviceToHost, st));
            }
            HIP_TRY(hipEventRecord(sh.end, st));
        }
        r->timing.host_ms_enqueue = static_cast<float>(now_ms() - t0);
        // everything of this frame is enqueued: the next frame's points (drawn meanwhile) go to the devices
        commit_ahead();
        // wait for every device (the pack / receive buffers are reused by the next frame); a pageable image gets its slices
        // from the staging buffers, one host thread per device
        std::vector<std::thread> movers;
        std::vector<int> sync_status(G, SAR_OK);
        for (uint32_t d = 0; d < G; ++d) {
            auto move = [&, d]() {
                Shard& sh = r->shards[d];
                if (hipSetDevice(sh.device) != hipSuccess || hipStreamSynchronize(sh.rt->stream) != hipSuccess) { sync_status[d] = SAR_ERR_HIP; return; }
                const uint64_t first = static_cast<uint64_t>(d) * S;
                const uint32_t n = first >= npix ? 0u : static_cast<uint32_t>((npix - first < S) ? npix - first : S);
                if (rgba_out_host && n && !pinned_out) std::memcpy(rgba_out_host + first * 4u, sh.h_rgba, static_cast<size_t>(n) * 8u);
            };
            try {
                movers.emplace_back(move);
            } catch (const std::system_error&) {
                move();
            }
        }
        for (auto& m : movers) m.join();
        for (uint32_t d = 0; d < G; ++d)
            if (sync_status[d] != SAR_OK) { set_error("device %d: stream synchronisation failed after the exchange", r->shards[d].device); return sync_status[d]; }
        for (Shard& sh : r->shards) {
            HIP_TRY(hipSetDevice(sh.device));
            float ms = 0.f;  // per-device stream time of the three phases; the frame is as slow as the slowest device
            if (hipEventElapsedTime(&ms, sh.begin, sh.packed) == hipSuccess && ms > r->timing.render_ms) r->timing.render_ms = ms;
            // exchange: until every device's quad is reduced here (that includes waiting for the slowest device's merge);
            // colorize: this device's own slice from then on
            if (hipEventElapsedTime(&ms, sh.packed, sh.reduced) == hipSuccess && ms > r->timing.exchange_ms) r->timing.exchange_ms = ms;
            if (hipEventElapsedTime(&ms, sh.reduced, sh.end) == hipSuccess && ms > r->timing.colorize_ms) r->timing.colorize_ms = ms;
        }
        return SAR_OK;
    };
    const int st = exchange();
    if (st != SAR_OK) return failed(st);
    r->timing.total_ms = static_cast<float>(now_ms() - t0);
    r->timing.exchange_bytes_per_device = static_cast<uint64_t>(G - 1) * blk;
    if (sparse) {  // what the push kernels really wrote to other devices (the busiest device's)
        unsigned long long most = 0;
        for (Shard& sh : r->shards) {
            unsigned long long b = 0;
            HIP_TRY(hipSetDevice(sh.device));
            HIP_TRY(hipMemcpy(&b, sh.d_bytes, sizeof(b), hipMemcpyDeviceToHost));
            most = b > most ? b : most;
        }
        r->timing.exchange_bytes_per_device = most;
    }
    r->timing.peer_access_failures = r->peer_access_failures;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
