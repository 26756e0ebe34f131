// sar_accumulate.hip — gfx950 (MI355X) kernels that turn what the iterate kernel left into the persistent Runtime
// buffers: k_bin_accumulate (record lists -> per-bin LDS histograms -> partial histograms) and k_fold_resolve (partial
// histograms + depth keys -> count / key / steps, payload of the new depth winners from the trajectory checkpoints).
#include "sar_device.hpp"
#include "sar_launch.hpp"

namespace sar {

// ---------------------------------------------------------------------------------------------------
// k_bin_accumulate — records -> per-pixel hit counts, in LDS
// ---------------------------------------------------------------------------------------------------
// grid (B, splits): block (b, s) owns bin b and the waves w with w % splits == s. Every thread walks
// whole (bin, wave) chunk lists (64-byte loads, newest chunk first) and adds the records into the
// bin's LDS histogram with LDS atomics; the histogram is then written — plainly, fully — as copy s of
// the scratch count bins, which k_fold_resolve sums into Runtime::count.
// Bins of 65536 pixels (the most a 16-bit record addresses — 4096^2 in 256 bins) do not fit 32-bit counters into the LDS:
//   PACKED: two 16-bit counters per LDS word, 128 KiB for the whole bin, ONE workgroup reads the lists
//     once. A counter is 15 bits plus a guard bit: the lane whose (returning) add sets the guard bit takes 32768 out again
//     and notes the pixel in a short event list; every event is worth 32768 hits when the histogram is written out.
//     No carry reaches the neighbouring counter WHILE A GUARD BIT IS SET, by a bound, not by timing: an add that FINDS the guard
//     bit set (the setter's subtraction is still on its way) takes itself out again and counts its hit in memory instead
//     (out[pixel] + 1, the write-out then adds to what is there). While the guard bit is set the counter therefore holds only adds
//     that have not yet undone themselves — one per lane (a lane waits for every add before its next): 1024, and never more than
//     the hardware lets a workgroup have in flight, 15 LDS operations per wave (lgkmcnt is four bits) x 64 lanes x 16 waves =
//     15360 < 32768. What the bound does NOT cover are undo-subtractions that land AFTER the setter cleared the guard: they are
//     still inside the counter, and a second guard event before they land leaves the counter below zero — a borrow from the
//     neighbouring counter (the word stays right modulo 2^32, the two 16-bit halves do not). Between an add and its undo lie two
//     trips through the LDS unit's queue — with every lane of the workgroup on ONE counter some 2 000 operations — against the
//     32768 adds a second event needs: a margin of 15x, and a TIMING argument, said to be one (round 6 measured where it ends:
//     eight adds in flight per lane stretch the window tenfold and the hot-pixel test fails). The tests drive one pixel through
//     9000 guard events and past 2^32 hits; a build with -DSAR_ACC_GUARD_CHECK asserts the bound (traps on a guarded counter above 0x4000).
//     (Round 2 counted such a bin in two halves, two workgroups reading every list: 4.35 ms per launch of configs[3] on one
//     GPU against 2.82 ms, which is the rate of isolated 64-byte reads, 2.7 of the 3.4 TB/s MI355X serves; that form left the
//     tree in round 4. Rounds 2-4 let the adds under a pending subtraction stand — correct unless 32768 of them fell into
//     that sub-microsecond window: a timing argument, replaced in round 5.)
constexpr uint32_t kAccEvents = 2040u;  // event list of the PACKED mode (u16 records), next to two counters
template <uint32_t R, uint32_t K, bool PACKED>
__device__ __forceinline__ void bin_accumulate_body(const BinAccArgs& a, uint32_t* hist) {
    constexpr uint32_t Q = kChunkQuads(R);       // 16-byte quads per chunk
    constexpr uint32_t G = kChunkLanes(R);       // lanes that share one list: lane q of a group reads quad q
    // PACKED: behind the 32768 words of counters: [0] events noted, [1] "an event went straight to memory", then the events
    uint32_t* const ev_ctl = hist + 32768u;
    unsigned short* const ev = (unsigned short*)(ev_ctl + 2);
    uint32_t* const out = a.scratch_count + (size_t)blockIdx.y * a.npix;
    const uint32_t b = blockIdx.x;
    const uint32_t s = blockIdx.y;
    const uint32_t hist_px = 1u << a.bin_shift;
    const uint32_t hist_words = PACKED ? hist_px / 2u : hist_px;
    auto pixel_of = [&](uint32_t rec) {  // (bin, record) -> image position (BinMap)
        return (rec & a.map.low_mask) | (b << a.map.seg_shift) | ((rec & ~a.map.low_mask) << a.map.hi_shift);
    };
    const uint32_t q = threadIdx.x % G;
    const uint32_t group = threadIdx.x / G, groups = blockDim.x / G;
    // Most bins of a frame are empty (the attractor covers a band of the image): a block with no chunk at all
    // leaves its partial histogram untouched — the scratch copies are all-zero between launches because
    // k_fold_resolve clears what it reads.
    int any = 0;
    for (uint32_t w = s + a.splits * threadIdx.x; w < a.n_waves; w += a.splits * blockDim.x)
        any |= a.heads[(size_t)b * a.n_waves + w] != kNoChunk;
    if (!__syncthreads_or(any)) return;
    for (uint32_t k = threadIdx.x; k < hist_words; k += blockDim.x) hist[k] = 0u;
    if (PACKED && threadIdx.x < 2u) ev_ctl[threadIdx.x] = 0u;
    __syncthreads();
    const uint4* arena = (const uint4*)a.arena;
    // One (bin, wave) list per group of G lanes: a chunk is ONE 16-byte load per lane and one cache line per
    // group (a lane that walked a list alone needed Q loads over 64 different lines per wave instruction).
    // A list is a chain of dependent loads (the next chunk's number is in this chunk's header): a group walks K lists at
    // the same time — K loads in flight per lane. (One workgroup per CU with the 128 KiB histogram: K = 2.)
    const uint32_t stride = a.splits * groups;
    for (uint32_t w0 = s + a.splits * group; w0 < a.n_waves; w0 += K * stride) {
        uint32_t chunk[K];
        const uint4* base[K];
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t w = w0 + k * stride;
            chunk[k] = w < a.n_waves ? a.heads[(size_t)b * a.n_waves + w] : kNoChunk;
            base[k] = arena + (size_t)(w < a.n_waves ? w : 0u) * a.chunks_per_wave * kChunkStride(R) + (q < Q ? q : 0u);
        }
        for (;;) {
            bool live = false;
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) live |= chunk[k] != kNoChunk;
            if (!live) break;
            uint4 v[K];
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                v[k] = make_uint4(kNoChunk, 0u, 0u, 0u);  // an ended list: no predecessor, no record
                if (chunk[k] != kNoChunk) v[k] = base[k][(size_t)chunk[k] * kChunkStride(R)];
            }
#pragma unroll
            for (uint32_t k = 0; k < K; ++k) {
                // the chunk header {previous chunk of the list, record count} sits in lane 0's quad
                const uint32_t prev = __shfl(v[k].x, 0, G);
                const uint32_t nrec = q < Q ? __shfl(v[k].y, 0, G) : 0u;
                // records held by this lane: lane 0 -> records 0..3 (its .z/.w), lane q -> 8q-4 .. 8q+3
                const uint32_t first = q == 0u ? 0u : 8u * q - 4u;
                const uint32_t r0 = q == 0u ? v[k].z : v[k].x, r1 = q == 0u ? v[k].w : v[k].y;
                // PACKED: one returning add per record; the counter's 15 bits were all set before it <=> this add set the
                // guard bit (inc = 1 or 1 << 16, so inc * 0x7FFF masks the counter)
                // what an add does that saw its counter's guard bit after itself (old: the word before the add): either the bit was set
                // already — its setter's subtraction is pending: this add steps back out, so that the counter never holds more than the
                // adds in flight (< 32768: no carry), and the hit is counted in memory — or this add set it: 32768 hits leave the
                // counter as one event
                auto packed_guard = [&](uint32_t rec, uint32_t inc, uint32_t old) {
                    const uint32_t guard = inc << 15;
                    if (old & guard) {
#ifdef SAR_ACC_GUARD_CHECK  // the bound itself: a guarded counter holds at most the adds in flight
                        if ((old & __umul24(inc, 0x7FFFu)) >= __umul24(inc, 0x4000u)) __builtin_trap();
#endif
#ifdef SAR_ACC_GUARD_CHECK  // ... and what the bound does not cover: an undo must never find its counter's 15 bits at zero (a borrow)
                        if ((atomicSub(&hist[rec >> 1], inc) & __umul24(inc, 0x7FFFu)) == 0u) __builtin_trap();
#else
                        atomicSub(&hist[rec >> 1], inc);
#endif
                        atomicAdd(&out[pixel_of(rec)], 1u);
                        ev_ctl[1] = 1u;
                    } else {
                        atomicSub(&hist[rec >> 1], guard);
                        const uint32_t e = atomicAdd(&ev_ctl[0], 1u);
                        if (e < kAccEvents) {
                            ev[e] = (unsigned short)rec;
                        } else {  // more than 2040 x 32768 hits on a handful of pixels in one block: straight to memory
                            atomicAdd(&out[pixel_of(rec)], 32768u);
                            ev_ctl[1] = 1u;
                        }
                    }
                };
                // PACKED: one returning add per record; ONE test on the common path: the counter's half of the word shows its guard
                // bit after this add — either this add set it (the 15 bits were all set) or it was set already (inc = 1 or 1 << 16)
                auto packed_add = [&](uint32_t rec) {
                    const uint32_t inc = __umul24(rec & 1u, 0xFFFFu) + 1u;
                    const uint32_t old = atomicAdd(&hist[rec >> 1], inc);
#ifdef SAR_EXPERIMENT_ACC_NO_UNDO  // A/B timing only: rounds 2-4's form (adds under a pending subtraction stand)
                    if (__builtin_expect((old & __umul24(inc, 0x7FFFu)) == __umul24(inc, 0x7FFFu), 0)) packed_guard(rec, inc, old & ~(inc << 15));
#else
                    if (__builtin_expect(((old + inc) & (inc << 15)) != 0u, 0)) packed_guard(rec, inc, old);
#endif
                };
                // (Round 6 issued the quad's eight returning adds back to back and looked at them together — one wait, one branch. Dropped:
                // 3.55 against 3.21 ms per launch of configs[3]'s frame — the LDS atomic unit is what is busy, a deeper queue in front
                // of it buys nothing — and WRONG on the hot-pixel test: with eight operations per lane queued in front of one counter,
                // the window between an add that found the guard set and its undo grows from ~2 000 to ~20 000 other operations, a
                // second guard event meets undos still pending and the counter ends below zero — a borrow from the neighbour that
                // stays. profiles/dead_ends.md, "Round 6".)
                if (PACKED && nrec == R) {
                    if (q < Q) {
                        packed_add(r0 & 0xFFFFu);
                        packed_add(r0 >> 16);
                        packed_add(r1 & 0xFFFFu);
                        packed_add(r1 >> 16);
                        if (q != 0u) {
                            packed_add(v[k].z & 0xFFFFu);
                            packed_add(v[k].z >> 16);
                            packed_add(v[k].w & 0xFFFFu);
                            packed_add(v[k].w >> 16);
                        }
                    }
                    chunk[k] = prev;
                    continue;
                }
                auto count = [&](bool valid, uint32_t rec) {  // rec: 16 bits
                    if (PACKED) {
                        if (valid) packed_add(rec);
                    } else if (valid) {
                        atomicAdd(&hist[rec], 1u);
                    }
                };
                count(first < nrec, r0 & 0xFFFFu);
                count(first + 1u < nrec, r0 >> 16);
                count(first + 2u < nrec, r1 & 0xFFFFu);
                count(first + 3u < nrec, r1 >> 16);
                if (q != 0u) {
                    count(first + 4u < nrec, v[k].z & 0xFFFFu);
                    count(first + 5u < nrec, v[k].z >> 16);
                    count(first + 6u < nrec, v[k].w & 0xFFFFu);
                    count(first + 7u < nrec, v[k].w >> 16);
                }
                chunk[k] = prev;
            }
        }
    }
    __syncthreads();
    // The non-zero counts go out as partial histogram s (the scratch copies are all-zero between launches), at the image
    // position the map gives (bin, record); 2048 consecutive records are 2048 consecutive pixels under both maps, and a
    // step of this loop (256 / 512 / 1024 threads) stays inside one such segment: its flag tells k_fold_resolve that the
    // segment has something to fold.
    const bool direct = PACKED && ev_ctl[1] != 0u;  // (wave-uniform) some events are already in `out`: add, do not overwrite
    if (direct) __threadfence();
    for (uint32_t k = threadIdx.x; k < hist_px; k += blockDim.x) {
        uint32_t v = PACKED ? (hist[k >> 1] >> ((k & 1u) << 4)) & 0xFFFFu : hist[k];
        const uint32_t px = pixel_of(k);
        if (direct && px < a.npix) v += __hip_atomic_load(&out[px], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool live = v != 0u && px < a.npix;
        if (live) out[px] = v;
        if (wave_ballot(live) && (threadIdx.x & 63u) == 0u) a.seg_any[px >> 11] = 1u;  // lane 0 holds the wave's lowest pixel
    }
    if (PACKED && ev_ctl[0] != 0u) {  // (block-uniform) 32768 hits per event, on top of what was just stored
        __threadfence();
        __syncthreads();
        const uint32_t n_ev = ev_ctl[0] < kAccEvents ? ev_ctl[0] : kAccEvents;
        for (uint32_t e = threadIdx.x; e < n_ev; e += blockDim.x) {
            const uint32_t px = pixel_of(ev[e]);
            atomicAdd(&out[px], 32768u);
            a.seg_any[px >> 11] = 1u;
        }
    }
}

template <uint32_t R, uint32_t K, bool PACKED>
__global__ void __launch_bounds__(1024) k_bin_accumulate(const BinAccArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    bin_accumulate_body<R, K, PACKED>(a, hist);
}
// F frames in one launch (BatchFrame): the frame is blockIdx.z, (bin, split) stay blockIdx.x / .y
template <uint32_t R, uint32_t K>
__global__ void __launch_bounds__(1024) k_bin_accumulate_batch(const BatchFrame* frames) {
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const BinAccArgs a = load_frame_args(&frames[blockIdx.z].acc);
    bin_accumulate_body<R, K, false>(a, hist);
}

// ---------------------------------------------------------------------------------------------------
// k_fold_resolve — scratch bins -> persistent Runtime buffers, then the payload of new depth winners
// ---------------------------------------------------------------------------------------------------
// Each block owns FOLD_PIX contiguous pixels: it folds the scratch copies into count / key (count add
// with the running max, depth test where the value already held wins ties), re-zeroes the scratch,
// compacts the pixels whose depth winner changed into LDS and then recomputes their colour-transform
// payload (the rare branch of render, :821-834) from the nearest trajectory checkpoint — the visit
// ordinal in the key names the job and the iteration. No global atomics except one max per block.
constexpr uint32_t FOLD_PIX = 2048;

__device__ __forceinline__ void fold_resolve_body(const FoldArgs& a) {
    __shared__ unsigned long long s_key[FOLD_PIX];
    __shared__ uint32_t s_pix[FOLD_PIX];
    __shared__ uint32_t s_n, s_wrap;
    __shared__ uint32_t s_tmp[4];
    if (threadIdx.x == 0) { s_n = 0; s_wrap = 0; }
    const uint32_t base = blockIdx.x * FOLD_PIX;
    // Binned path: most of a frame is empty (the attractor touches a fifth of the pixels). k_bin_accumulate flags the
    // 2048-pixel segments that received a count; no visit in the segment means no partial count and no depth key to fold —
    // only block 0 must always run (the NaN iterations land on pixel 0).
    static_assert(FOLD_PIX == 2048u, "seg_any is per 2048 pixels");
    if (a.seg_any && blockIdx.x != 0 && a.seg_any[blockIdx.x] == 0u) return;
    __syncthreads();

    uint32_t local_max = 0;
    // what one pixel does with the hits and the best depth key this launch left for it
    auto commit = [&](uint32_t px, unsigned long long add, unsigned long long kbest) {
        if (px == 0 && a.nan_count) {  // diverged trajectories: every iteration after the NaN hits (0,0)
            add += *a.nan_count;
            *a.nan_count = 0;
        }
        if (add) {
            // count += hits, wrapping like the release build (:811); if the u32 wraps, the reference's
            // running max (:813-815) has seen u32::MAX on the way.
            const unsigned long long total = (unsigned long long)a.count[px] + add;
            if (total >> 32) s_wrap = 1;
            const uint32_t c32 = (uint32_t)total;
            a.count[px] = c32;
            local_max = c32 > local_max ? c32 : local_max;
        }
        // depth test (:821): strictly greater than what the runtime already holds (an earlier render
        // call or launch chunk wins ties; within the chunk the lowest ordinal already won the atomic max)
        if (kbest && (uint32_t)(kbest >> 32) > (uint32_t)(a.key[px] >> 32)) {
            const uint32_t pos = atomicAdd(&s_n, 1u);
            s_key[pos] = kbest;
            s_pix[pos] = px;
        }
    };
    if ((a.npix & 3u) == 0u) {
        // four pixels per thread: 16-byte loads of every partial histogram (the copies are 16-byte aligned when the pixel
        // count is a multiple of four)
        for (uint32_t q = threadIdx.x; q < FOLD_PIX / 4u; q += blockDim.x) {
            const uint32_t px0 = base + 4u * q;
            if (px0 >= a.npix) break;
            unsigned long long add[4] = {0, 0, 0, 0}, kb[4] = {0, 0, 0, 0};
            for (uint32_t c = 0; c < a.copies; ++c) {
                uint4* sp = (uint4*)(a.scratch_count + (size_t)c * a.npix + px0);
                const uint4 v = *sp;
                if (v.x | v.y | v.z | v.w) {
                    add[0] += v.x; add[1] += v.y; add[2] += v.z; add[3] += v.w;
                    *sp = make_uint4(0u, 0u, 0u, 0u);
                }
            }
            for (uint32_t c = 0; c < a.key_copies; ++c) {
                ulonglong2* kp = (ulonglong2*)(a.scratch_key + (size_t)c * a.npix + px0);
                const ulonglong2 k0 = kp[0], k1 = kp[1];
                if (k0.x | k0.y | k1.x | k1.y) {
                    kb[0] = k0.x > kb[0] ? k0.x : kb[0]; kb[1] = k0.y > kb[1] ? k0.y : kb[1];
                    kb[2] = k1.x > kb[2] ? k1.x : kb[2]; kb[3] = k1.y > kb[3] ? k1.y : kb[3];
                    kp[0] = make_ulonglong2(0ull, 0ull);
                    kp[1] = make_ulonglong2(0ull, 0ull);
                }
            }
#pragma unroll
            for (uint32_t e = 0; e < 4u; ++e) commit(px0 + e, add[e], kb[e]);
        }
    } else {
        for (uint32_t k = threadIdx.x; k < FOLD_PIX; k += blockDim.x) {
            const uint32_t px = base + k;
            if (px >= a.npix) break;
            unsigned long long add = 0, kbest = 0;
            for (uint32_t c = 0; c < a.copies; ++c) {
                const size_t o = (size_t)c * a.npix + px;
                const uint32_t sc = a.scratch_count[o];
                if (sc) { add += sc; a.scratch_count[o] = 0; }
            }
            for (uint32_t c = 0; c < a.key_copies; ++c) {
                const size_t o = (size_t)c * a.npix + px;
                const unsigned long long sk = a.scratch_key[o];
                if (sk) { kbest = sk > kbest ? sk : kbest; a.scratch_key[o] = 0; }
            }
            commit(px, add, kbest);
        }
    }
    __syncthreads();

    const uint32_t total = s_n;
    const uint32_t n = (uint32_t)a.iters;
    const size_t cs = a.n_jobs;
    for (uint32_t w = threadIdx.x; w < total; w += blockDim.x) {
        const unsigned long long wk = s_key[w];
        const uint32_t ord = 0xFFFFFFFFu - (uint32_t)wk;
        const uint32_t job = ord / n;
        const uint32_t t = ord - job * n;
        const uint32_t k = t / a.ckpt_stride;
        const uint32_t r = t - k * a.ckpt_stride;
        const double* ck = a.ckpt + (size_t)k * 3 * cs + job;
        double x = ck[0], y = ck[cs], z = ck[2 * cs];
        for (uint32_t s = 0; s < r; ++s) next_point(a.p, x, y, z);
        const double px = x, py = y, pz = z;  // previous_point (:766 / :836)
        next_point(a.p, x, y, z);             // current_point (:770)
        double sx, sy, sz;
        screen_space(a.p, x, y, z, sx, sy, sz);
        a.steps[s_pix[w]] = color_transform(a.ct, x - px, y - py, z - pz, sx, sy, sz);  // :822-830
        a.key[s_pix[w]] = wk | 0xFFFFFFFFull;                                            // :832
    }

    const uint32_t m = block_max_u32(local_max, s_tmp);
    if (threadIdx.x == 0) {
        if (m) raise_scalar(&a.scalars[SC_MAX], m);
        if (s_wrap) atomicOr(&a.scalars[SC_WRAP], 1u);
    }
}

__global__ void __launch_bounds__(256) k_fold_resolve(const FoldArgs a) { fold_resolve_body(a); }
__global__ void __launch_bounds__(256) k_fold_resolve_batch(const BatchFrame* frames) {
    const FoldArgs a = load_frame_args(&frames[blockIdx.z].fold);
    fold_resolve_body(a);
}

// lists: (bin, wave) lists a lane group walks at the same time — 4 with the 128 KiB histograms (one workgroup per CU: four
// loads in flight per lane make up for the missing second workgroup), 1 otherwise
#define SAR_FOR_EACH_ACC(X) X(12u, 1u) X(12u, 4u) X(20u, 1u) X(20u, 4u) X(28u, 1u) X(28u, 4u) X(60u, 1u) X(60u, 4u)
int launch_bin_accumulate(const BinAccArgs& a, uint32_t threads, uint32_t records, uint32_t lists, hipStream_t s) {
    const bool packed = a.bin_shift == 16u;
    const size_t lds = packed ? (size_t)(32768u + 2u) * 4u + kAccEvents * 2u : (size_t)4u << a.bin_shift;
    // a list takes a group of 2, 4 or 8 lanes: 1024 threads walk 128..512 lists per block, `lists` per group at a time
    if (threads == 0) threads = 1024u;
    const dim3 grid(a.n_bins, a.splits);
    bool launched = false;
#define SAR_ACC(RR, KK)                                                                                         \
    if (!launched && records == RR && lists == KK) {                                                             \
        if (packed) hipLaunchKernelGGL((k_bin_accumulate<RR, KK, true>), grid, dim3(threads), lds, s, a);        \
        else hipLaunchKernelGGL((k_bin_accumulate<RR, KK, false>), grid, dim3(threads), lds, s, a);              \
        launched = true;                                                                                        \
    }
    SAR_FOR_EACH_ACC(SAR_ACC)
#undef SAR_ACC
    return launched ? 0 : 1;
}

// the batched form exists for bins of up to 32768 pixels (32-bit counters)
int launch_bin_accumulate_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_bins, uint32_t splits, uint32_t bin_shift,
                                uint32_t threads, uint32_t records, uint32_t lists, hipStream_t s) {
    if (bin_shift > 15u) return 1;
    const size_t lds = (size_t)4u << bin_shift;
    if (threads == 0) threads = 1024u;
    const dim3 grid(n_bins, splits, n_frames);
    bool launched = false;
#define SAR_ACC_BATCH(RR, KK)                                                                                   \
    if (!launched && records == RR && lists == KK) {                                                             \
        hipLaunchKernelGGL((k_bin_accumulate_batch<RR, KK>), grid, dim3(threads), lds, s, frames);               \
        launched = true;                                                                                        \
    }
    SAR_FOR_EACH_ACC(SAR_ACC_BATCH)
#undef SAR_ACC_BATCH
    return launched ? 0 : 1;
}

void launch_fold_resolve_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_fold_resolve_batch, dim3((npix + FOLD_PIX - 1) / FOLD_PIX, 1, n_frames), dim3(256), 0, s, frames);
}

int accumulate_kernel_attributes() {
    // a bin's histogram needs more dynamic LDS than the 64 KiB default window when the bin has 32768 pixels
    hipError_t e = hipSuccess;
#define SAR_ATTR(RR, KK) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bin_accumulate<RR, KK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bin_accumulate<RR, KK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    SAR_FOR_EACH_ACC(SAR_ATTR)
#undef SAR_ATTR
#define SAR_ATTR_BATCH(RR, KK) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bin_accumulate_batch<RR, KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    SAR_FOR_EACH_ACC(SAR_ATTR_BATCH)
#undef SAR_ATTR_BATCH
    return (int)e;
}

int binned_kernel_attributes() {
    const int e = iterate_kernel_attributes();
    return e ? e : accumulate_kernel_attributes();
}

void launch_fold_resolve(const FoldArgs& a, hipStream_t s) {
    const uint32_t grid = (a.npix + FOLD_PIX - 1) / FOLD_PIX;
    hipLaunchKernelGGL(k_fold_resolve, dim3(grid), dim3(256), 0, s, a);
}

}  // namespace sar
