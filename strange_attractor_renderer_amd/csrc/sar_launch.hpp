// sar_launch.hpp — host-callable launch wrappers implemented in sar_iterate.hip, sar_accumulate.hip and sar_image.hip.
#pragma once

#include <hip/hip_runtime.h>

#include "sar_internal.hpp"

namespace sar {

// the one-atomic-per-visit fallback (images beyond 64 Mpx, single jobs whose record arena would not fit)
void launch_iterate(const IterArgs& a, uint32_t block, hipStream_t s);
uint32_t lean_wave_lds_bytes(uint32_t bins, uint32_t records);  // LDS staging of one wave (or wave pair) of the binned path
uint32_t chunk_bytes(uint32_t records);
// records: 12 / 20 / 28 / 60 per chunk; hint_bytes: 2 (16-bit fixed-point hints) or 4 (the depth itself as f32);
// split: producer / consumer wave pairs (k_iterate_split: 28 or 60 records) instead of the whole kernel (k_iterate_lean)
int launch_iterate_lean(const BinIterArgs& a, uint32_t block, uint32_t records, uint32_t hint_bytes, bool split, hipStream_t s);
// lists: 1 or 4 (bin, wave) lists per lane group at a time; bins of 65536 pixels are counted with packed 16-bit counters
int launch_bin_accumulate(const BinAccArgs& a, uint32_t threads, uint32_t records, uint32_t lists, hipStream_t s);
int iterate_kernel_attributes();     // sar_iterate.hip
int accumulate_kernel_attributes();  // sar_accumulate.hip
int binned_kernel_attributes();      // both
void launch_fold_resolve(const FoldArgs& a, hipStream_t s);
void launch_reset_batch(const ResetBatch& t, uint32_t n_frames, uint32_t npix, hipStream_t s);
void launch_colorize_gas_batch(const ColorizeBatch& t, uint32_t n_frames, const double* lut, uint32_t lut_len, const PaletteParams& pal, double b_offset,
                               double b_factor, int transparent, uint32_t npix, hipStream_t s);
void launch_reset(uint32_t* count, unsigned long long* key, double* steps, uint32_t npix, uint32_t* scalars, void* hints,
                  uint32_t hint_words, uint32_t hint_fill, hipStream_t s);
void launch_zbuf_out(const unsigned long long* key, float* out, uint32_t npix, hipStream_t s);
void launch_zbuf_in(const float* z, unsigned long long* key, uint32_t npix, hipStream_t s);
void launch_merge(uint32_t* count, unsigned long long* key, double* steps, const uint32_t* ocount,
                  const unsigned long long* okey, const double* osteps, uint32_t npix, uint32_t* scalars,
                  hipStream_t s);
void launch_colorize_gas(const uint32_t* count, const double* steps, const uint32_t* scalars, const double* lut,
                         uint32_t lut_len, const PaletteParams& pal, double b_offset, double b_factor,
                         int transparent, uint32_t npix, void* out, hipStream_t s);
void launch_colorize_depth(const unsigned long long* key, uint32_t* scalars, uint32_t npix, void* out,
                           hipStream_t s);
void launch_colorize_depth_range(const unsigned long long* key, const uint32_t* scalars, uint32_t n, void* out, hipStream_t s);
// sliced exchange (sar_image.hip): S = pixels per slice, G = ranks; blocks of S*16 bytes [count | sortable z | steps]
void launch_exch_pack(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t npix, uint32_t S,
                      uint32_t G, void* out, hipStream_t s);
void launch_exch_merge_slices(uint32_t* count, unsigned long long* key, double* steps, uint32_t first, uint32_t n, uint32_t S,
                              uint32_t G, const void* in, uint32_t* scalars, bool keep_max, hipStream_t s);
// sparse form: records of 64-pixel granules (kExchRecordBytes each), placed by slot tables
void launch_exch_flags(const uint32_t* count, const unsigned long long* key, uint32_t npix, void* flags, hipStream_t s);
void launch_exch_plan(const void* flags_all, uint32_t world, uint32_t rank, uint32_t nseg, uint32_t sps, int32_t* send_slot, int32_t* recv_slot,
                      uint32_t* counts, hipStream_t s);
void launch_exch_pack_sparse(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t npix, const int32_t* send_slot,
                             void* out, hipStream_t s);
void launch_exch_push(const ExchPushArgs& a, hipStream_t s);
void launch_exch_merge_sparse(uint32_t* count, unsigned long long* key, double* steps, uint32_t first, uint32_t n, uint32_t sps, uint32_t G,
                              const void* records, const int32_t* slot, uint32_t* scalars, bool keep_max, hipStream_t s);
void launch_exch_scalars_export(const uint32_t* scalars, void* out4, hipStream_t s);
void launch_exch_scalars_import(uint32_t* scalars, const void* in4, hipStream_t s);
void launch_exch_scalars_reduce(uint32_t* scalars, const void* board, uint32_t G, hipStream_t s);  // board: [G][4] int64 quads
int launch_convert(const void* rgba16, int format, void* out, uint32_t npix, hipStream_t s);
// returns the number of blocks = 12-double partial results written to `out`
uint32_t launch_extent(const MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* out, hipStream_t s);
void launch_starts_soa(const double* aos, double* soa, uint32_t m, hipStream_t s);
void launch_warmup(const WarmArgs& a, hipStream_t s);
// batched launches (sar_batch.cpp): `frames` is a table of n_frames BatchFrame in device memory, the frame is blockIdx.z
void launch_batch_clear(const BatchFrame* frames, uint32_t n_frames, uint32_t seg_words, hipStream_t s);
void launch_batch_fetch(const BatchFrame* frames, uint32_t n_frames, hipStream_t s);
void launch_warmup_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_jobs, bool first_phase, hipStream_t s);
uint32_t batch_xcd_map(uint32_t n_frames, uint32_t n_waves);
int launch_iterate_split_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_waves, uint32_t n_bins, uint32_t records,
                               uint32_t hint_bytes, uint32_t xcd_map, hipStream_t s);
int launch_bin_accumulate_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_bins, uint32_t splits, uint32_t bin_shift,
                                uint32_t threads, uint32_t records, uint32_t lists, hipStream_t s);
void launch_fold_resolve_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t npix, hipStream_t s);
void launch_dead_jobs(const uint32_t* active, uint32_t n_jobs, uint64_t iters, unsigned long long* nan_count, hipStream_t s);
void launch_exch_export(const unsigned long long* key, uint32_t rank, void* out, uint32_t npix, hipStream_t s);
void launch_exch_select(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t rank,
                        const void* reduced, void* out, uint32_t npix, hipStream_t s);
void launch_exch_import(uint32_t* count, unsigned long long* key, double* steps, const void* reduced,
                        const void* sum, uint32_t npix, uint32_t* scalars, hipStream_t s);

}  // namespace sar
