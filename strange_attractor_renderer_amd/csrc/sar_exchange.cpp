// sar_exchange.cpp — the ONE exchange step before colorize of the one-process-per-GPU path, behind one context object
// (include/sar.h: sar_exchange_*): Runtime::merge (reference src/lib.rs:708-738) folded in rank order (:1068-1076) over image
// slices. The collectives themselves (RCCL through torch.distributed, or anything else) are the caller's; everything between
// them — which granules travel, where every record goes and arrives, packing, folding, the scalars — happens here and in the
// kernels of sar_image.hip. Host logic only. (The multi-device renderer of ONE process, sar_multi.cpp, pushes its records with
// kernel stores to peer memory and needs none of this.)
#include <cstring>
#include <new>

#include "sar_plan.hpp"

using namespace sar;

struct sar_exchange {
    sar_runtime* rt = nullptr;  // borrowed: every call but sar_exchange_free needs it alive
    int device = 0;
    uint32_t world = 1, rank = 0;
    uint32_t npix = 0;          // the runtime's image size when the context was made (a resized runtime needs a new context)
    uint32_t S = 0;             // pixels per slice
    uint32_t sps = 0;           // granules per slice
    uint32_t nseg = 0;          // granules of the image
    uint32_t first = 0, n = 0;  // my slice
    int32_t* d_send_slot = nullptr;   // [nseg]
    int32_t* d_recv_slot = nullptr;   // [world * sps]
    uint32_t* d_counts = nullptr;     // [2 world + 1]
    uint32_t* h_counts = nullptr;     // page-locked copy
    hipEvent_t planned = nullptr;
    bool sparse = false;              // the form the last pack chose (merge follows it)
};

namespace {

int slice_pixels(uint32_t npix, uint32_t world, uint32_t& out) {
    // whole 2048-pixel blocks (k_fold_resolve's unit; whole granules of the sparse exchange)
    const uint64_t s = ((static_cast<uint64_t>(npix) + world - 1) / world + (kExchSliceAlign - 1u)) & ~static_cast<uint64_t>(kExchSliceAlign - 1u);
    if (s * world > 0xFFFFFFFFull) { set_error("slice geometry exceeds 2^32 pixels"); return SAR_ERR_RANGE; }
    out = static_cast<uint32_t>(s);
    return SAR_OK;
}

int same_image(const sar_exchange* ex) {
    if (ex->rt->npix == ex->npix) return SAR_OK;
    set_error("exchange context made for %u pixels, the runtime now holds %u: make a new one", ex->npix, ex->rt->npix);
    return SAR_ERR_DIM_MISMATCH;
}

}  // namespace

extern "C" {

int sar_exchange_slice_pixels(uint32_t npix, uint32_t world, uint32_t* out_slice_pixels) try {
    if (!out_slice_pixels || world == 0) return SAR_ERR_INVALID;
    return slice_pixels(npix, world, *out_slice_pixels);
} catch (...) { return sar::abi_caught(); }

int sar_exchange_new(sar_runtime* rt, uint32_t world, uint32_t rank, sar_exchange** out, sar_exchange_layout* layout_out) try {
    if (!out) return SAR_ERR_INVALID;
    *out = nullptr;
    if (!rt || world == 0 || rank >= world || world > kMaxExchRanks) { set_error("sar_exchange_new: rank %u of %u (at most %u ranks)", rank, world, kMaxExchRanks); return SAR_ERR_INVALID; }
    sar_exchange* ex = new (std::nothrow) sar_exchange();
    if (!ex) return SAR_ERR_OOM;
    ex->rt = rt;
    ex->device = rt->device;
    ex->world = world;
    ex->rank = rank;
    ex->npix = rt->npix;
    int st = slice_pixels(rt->npix, world, ex->S);
    if (st != SAR_OK) { delete ex; return st; }
    ex->sps = ex->S / kExchSeg;
    ex->nseg = (rt->npix + kExchSeg - 1u) / kExchSeg;
    const uint64_t first = static_cast<uint64_t>(rank) * ex->S;
    ex->n = first >= rt->npix ? 0u : static_cast<uint32_t>((rt->npix - first < ex->S) ? rt->npix - first : ex->S);
    ex->first = ex->n ? static_cast<uint32_t>(first) : 0u;
    auto fail = [&](int code) { sar_exchange_free(ex); return code; };
    if (hipSetDevice(rt->device) != hipSuccess) return fail(SAR_ERR_HIP);
    if (hipMalloc(reinterpret_cast<void**>(&ex->d_send_slot), static_cast<size_t>(ex->nseg) * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ex->d_recv_slot), static_cast<size_t>(world) * ex->sps * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ex->d_counts), (2u * world + 1u) * sizeof(uint32_t)) != hipSuccess) return fail(SAR_ERR_OOM);
    if (hipHostMalloc(reinterpret_cast<void**>(&ex->h_counts), (2u * world + 1u) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return fail(SAR_ERR_OOM);
    if (hipEventCreateWithFlags(&ex->planned, hipEventDisableTiming) != hipSuccess) return fail(SAR_ERR_HIP);
    if (layout_out) {
        std::memset(layout_out, 0, sizeof(*layout_out));
        layout_out->world = world;
        layout_out->rank = rank;
        layout_out->slice_pixels = ex->S;
        layout_out->first_px = ex->first;
        layout_out->n_px = ex->n;
        layout_out->granules = ex->nseg;
        layout_out->block_bytes = static_cast<uint64_t>(world) * ex->S * 16u;
    }
    *out = ex;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_free(sar_exchange* ex) try {
    if (!ex) return SAR_OK;
    // (the slot tables may still be read by a kernel in flight; the runtime itself may be gone already — it is not touched here)
    hipSetDevice(ex->device);
    hipDeviceSynchronize();
    if (ex->d_send_slot) hipFree(ex->d_send_slot);
    if (ex->d_recv_slot) hipFree(ex->d_recv_slot);
    if (ex->d_counts) hipFree(ex->d_counts);
    if (ex->h_counts) hipHostFree(ex->h_counts);
    if (ex->planned) hipEventDestroy(ex->planned);
    delete ex;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_flags(sar_exchange* ex, uint8_t* flags_out_dev) try {
    if (!ex || !flags_out_dev) return SAR_ERR_INVALID;
    sar_runtime* rt = ex->rt;
    SAR_TRY(same_image(ex));
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_flags(rt->d_count, rt->d_key, rt->npix, flags_out_dev, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_pack(sar_exchange* ex, const uint8_t* flags_all_dev, double dense_above, void* send_dev, uint64_t* send_bytes,
                      uint64_t* recv_bytes, int* sparse_out) try {
    if (!ex || !send_dev || !send_bytes || !recv_bytes) return SAR_ERR_INVALID;
    sar_runtime* rt = ex->rt;
    const uint32_t G = ex->world;
    SAR_TRY(same_image(ex));
    HIP_TRY(hipSetDevice(rt->device));
    bool sparse = false;
    if (flags_all_dev) {
        // the plan: two block scans and a count over the gathered flags, then the 2 G + 1 numbers the all-to-all needs come to the
        // host — the ONE host wait of a frame's exchange
        HIP_TRY(hipMemsetAsync(ex->d_counts, 0, (2u * G + 1u) * sizeof(uint32_t), rt->stream));
        launch_exch_plan(flags_all_dev, G, ex->rank, ex->nseg, ex->sps, ex->d_send_slot, ex->d_recv_slot, ex->d_counts, rt->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ex->h_counts, ex->d_counts, (2u * G + 1u) * sizeof(uint32_t), hipMemcpyDeviceToHost, rt->stream));
        HIP_TRY(hipEventRecord(ex->planned, rt->stream));
        HIP_TRY(hipEventSynchronize(ex->planned));
        // a frame that covers the image goes the dense way (every rank holds the same flags, so every rank decides alike)
        sparse = static_cast<double>(ex->h_counts[2u * G]) <= dense_above * static_cast<double>(G) * ex->nseg;
    }
    if (sparse) {
        for (uint32_t r = 0; r < G; ++r) {
            send_bytes[r] = static_cast<uint64_t>(ex->h_counts[r]) * kExchRecordBytes;
            recv_bytes[r] = static_cast<uint64_t>(ex->h_counts[G + r]) * kExchRecordBytes;
        }
        launch_exch_pack_sparse(rt->d_count, rt->d_key, rt->d_steps, rt->npix, ex->d_send_slot, send_dev, rt->stream);
    } else {
        for (uint32_t r = 0; r < G; ++r) send_bytes[r] = recv_bytes[r] = static_cast<uint64_t>(ex->S) * 16u;
        launch_exch_pack(rt->d_count, rt->d_key, rt->d_steps, rt->npix, ex->S, G, send_dev, rt->stream);
    }
    HIP_TRY(hipGetLastError());
    ex->sparse = sparse;
    if (sparse_out) *sparse_out = sparse ? 1 : 0;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_merge(sar_exchange* ex, const void* recv_dev, int64_t* scalars_out_dev) try {
    if (!ex || !recv_dev || !scalars_out_dev) return SAR_ERR_INVALID;
    sar_runtime* rt = ex->rt;
    SAR_TRY(same_image(ex));
    HIP_TRY(hipSetDevice(rt->device));
    if (ex->sparse)
        launch_exch_merge_sparse(rt->d_count, rt->d_key, rt->d_steps, ex->first, ex->n, ex->sps, ex->world, recv_dev, ex->d_recv_slot, rt->d_scalars,
                                 ex->rank == 0, rt->stream);
    else
        launch_exch_merge_slices(rt->d_count, rt->d_key, rt->d_steps, ex->first, ex->n, ex->S, ex->world, recv_dev, rt->d_scalars, ex->rank == 0,
                                 rt->stream);
    launch_exch_scalars_export(rt->d_scalars, scalars_out_dev, rt->stream);
    HIP_TRY(hipGetLastError());  // (the depth hints stay valid: a merge only raises zbuf)
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_finish(sar_exchange* ex, const int64_t* scalars_dev) try {
    if (!ex || !scalars_dev) return SAR_ERR_INVALID;
    sar_runtime* rt = ex->rt;
    SAR_TRY(same_image(ex));
    HIP_TRY(hipSetDevice(rt->device));
    launch_exch_scalars_import(rt->d_scalars, scalars_dev, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_exchange_rooted(sar_exchange* ex, uint32_t step, void* key_i64_dev, void* sum_i32_dev) try {
    if (!ex || !key_i64_dev || step > 2u || (step && !sum_i32_dev)) return SAR_ERR_INVALID;
    sar_runtime* rt = ex->rt;
    SAR_TRY(same_image(ex));
    HIP_TRY(hipSetDevice(rt->device));
    if (step == 0u) launch_exch_export(rt->d_key, ex->rank, key_i64_dev, rt->npix, rt->stream);
    else if (step == 1u) launch_exch_select(rt->d_count, rt->d_key, rt->d_steps, ex->rank, key_i64_dev, sum_i32_dev, rt->npix, rt->stream);
    else launch_exch_import(rt->d_count, rt->d_key, rt->d_steps, key_i64_dev, sum_i32_dev, rt->npix, rt->d_scalars, rt->stream);
    HIP_TRY(hipGetLastError());
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
