// sar_iterate.hip — gfx950 (MI355X) kernels that run the map: k_warmup, k_iterate_split / k_iterate_lean (the hot loop as
// wave pairs / whole), k_iterate (one global atomic per visit: the fallback beyond 64 Mpx), k_extent, k_starts_soa. DESIGN.md section 3 has the measurements
// behind every choice; sar_device.hpp states the bit-exactness contract.
#include "sar_device.hpp"
#include "sar_launch.hpp"

namespace sar {

// ---------------------------------------------------------------------------------------------------
// k_iterate — render's loop (src/lib.rs:747-838) with ONE global atomic per visit: the fallback for images beyond
// 64 Mpx (more bins than the LDS staging holds) and for single jobs whose record arena would not fit. The chip retires
// ~2.1e10 scattered atomics per second: 45 ms per 1e9 visits, an order of magnitude behind the binned path.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_iterate(const IterArgs a) {
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= a.n_jobs) return;
    // 30 coefficients + 9 matrix entries + 10 projection constants are 98 SGPRs as kernel arguments —
    // more than the scalar file holds next to pointers and exec masks, and the compiler then spills
    // SGPRs to VGPR lanes inside the loop (v_readlane per use). The coefficients stay scalar operands;
    // the matrix and the projection constants are pinned into (plentiful) VGPRs instead.
    MapParams p = a.p;
#pragma unroll
    for (int k = 0; k < 9; ++k) p.m[k] = vgpr_pin(p.m[k]);
    p.sin_v = vgpr_pin(p.sin_v);
    p.cos_v = vgpr_pin(p.cos_v);
    p.ccx = vgpr_pin(p.ccx);
    p.ccy = vgpr_pin(p.ccy);
    p.ccz = vgpr_pin(p.ccz);
    p.width = vgpr_pin(p.width);
    p.height = vgpr_pin(p.height);
    p.half_height = vgpr_pin(p.half_height);
    p.width_scaled = vgpr_pin(p.width_scaled);
    p.scale_adjusted_mid = vgpr_pin(p.scale_adjusted_mid);

    double x = a.starts[job];
    double y = a.starts[a.n_jobs + job];
    double z = a.starts[2u * a.n_jobs + job];

    // "skip first 1000 to get good values in the attractor" (:750-752) — unless this launch continues a trajectory
    if (!a.resume)
        for (int w = 0; w < 1000; ++w) next_point(p, x, y, z);

    const uint32_t n = (uint32_t)a.iters;
    // visit ordinal = job*n + t (job-major, iteration-minor == the sequential order of the reference);
    // the key's low word is 0xFFFFFFFF - ordinal so that the EARLIEST visit wins a depth tie.
    const uint32_t lo_base = 0xFFFFFFFFu - job * n;
    const uint32_t C = a.ckpt_stride;
    const size_t cs = a.n_jobs;  // checkpoint component stride

    uint32_t t = 0;
    double* ck = a.ckpt + job;
    bool ended = false;
    while (t < n && !ended) {
        // checkpoint: the state BEFORE iteration t (coalesced 512-B rows per wave)
        ck[0] = x;
        ck[cs] = y;
        ck[2 * cs] = z;
        ck += 3 * cs;
        const uint32_t tend = (n - t > C) ? t + C : n;
        for (; t < tend; ++t) {
            next_point(p, x, y, z);  // :770
            if (x != x) {
                // Absorbing state: a NaN x makes every coordinate NaN from the next iteration on, and
                // already makes all of screen space NaN now, so this and every remaining iteration
                // passes the bounds test (:789, all comparisons false), casts to pixel (0,0)
                // (:800-802) and never wins the depth test. Add them in one go instead of hammering
                // one address n-t times.
                __hip_atomic_fetch_add(a.scratch_count, n - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ended = true;
                break;
            }
            double sx, sy, sz;
            screen_space(p, x, y, z, sx, sy, sz);  // :773
            const double ax = sx + p.ccx;          // center_camera.x with screen_space.x
            const double az = sz + p.ccy;          // center_camera.y with screen_space.z (:776-779)
            const double x2 = ax * p.cos_v + az * p.sin_v;
            const double z2 = ax * p.sin_v - az * p.cos_v;
            const double fi = (p.scale_adjusted_mid - x2) * p.width_scaled;  // :783
            const double fj = p.half_height - (sy + p.ccz) * p.width_scaled; // :786
            if (fi >= p.width || fj >= p.height || fi < 0. || fj < 0.) continue;  // :789-795
            const uint32_t i = (fi == fi) ? (uint32_t)fi : 0u;  // Rust `as u32`: NaN -> 0
            const uint32_t j = (fj == fj) ? (uint32_t)fj : 0u;
            const uint32_t idx = j * a.width + i;
            __hip_atomic_fetch_add(a.scratch_count + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // :807-812
            float zf = (float)z2;  // `z2 as f32`
            // strict `>` against an initial -1.0 (:693, :821): z <= -1 and NaN can never win
            if (zf > -1.0f) {
                zf = zf + 0.0f;  // -0.0 -> +0.0 so the integer order agrees with the float order
                const unsigned long long k =
                    ((unsigned long long)f32_sortable(zf) << 32) | (unsigned long long)(lo_base - t);
                __hip_atomic_fetch_max(a.scratch_key + idx, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (a.state_out) {  // a job of more than 2^32-2 iterations continues from here in the next launch (NaN stays NaN)
        a.state_out[job] = x;
        a.state_out[a.n_jobs + job] = y;
        a.state_out[2u * a.n_jobs + job] = z;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_iterate_lean — the hot loop without a global atomic per visit
// ---------------------------------------------------------------------------------------------------
// Measured on MI355X: the chip retires ~2.1e10 scattered global atomics per second whatever their
// scope or width, while the fp64 arithmetic of this loop alone runs at ~3.3e11 iterations/s. So a
// visit must not cost a global atomic. Here every visit becomes a 2-byte RECORD instead:
//
//   * the image is cut into B bins (BinMap: 2^bin_shift pixels each, dealt round-robin in 2048-pixel segments); a record
//     is the pixel's position inside its bin (u16);
//   * each WAVE owns B staging buffers of R records in LDS; a visit takes a slot with one LDS atomic (ds_add_rtn) and writes
//     its u16 there one iteration later;
//   * a full buffer leaves the LDS as ONE chunk {link to the previous chunk of this (wave, bin), count, records} into the
//     wave's private arena in HBM — position from a wave-local cursor, so no global atomic and nothing to wait for;
//   * k_bin_accumulate later walks the per-(bin, wave) chunk lists and histograms them in LDS.
//
// Depth: see DepthPipe::settle_depth — two filters (this XCD's hint, then the chip-wide key) in front of the
// 64-bit atomic max, as a software pipeline U visits deep. A stale or lost hint only costs an extra atomic,
// never a wrong result.
//
// Control flow: lanes without a visit are masked off for the slot request and the record write (two scalar instructions each:
// vector issue is the scarcer resource), the hint of a lane without a depth candidate is loaded from element 0 (a load under
// the candidates' exec mask costs 8 % at 4096^2), and only the rare-per-lane events keep a wave-level branch: "some lane
// filled a buffer", "some lane passed a depth filter". A trajectory that ended in NaN is looked for once per checkpoint,
// not per iteration: until then its iterations land on pixel (0,0) through the ordinary record path, exactly where the
// reference counts them.
//
// PoolStager below is everything a wave needs to turn a stream of visits into staged records + depth candidates: one visit
// per lane per step(); all per-visit state lives in registers, the staging buffers in the wave's LDS slice. (Rounds 1-2
// also had a stager whose filling lane copied its buffer out by itself — ~30 instructions that 90 % of a wave's iterations
// ran with 2-3 lanes active; it left the tree in round 4, see profiles/dead_ends.md.)

// The depth path: two filters (this XCD's hint, then the chip-wide key) in front of the 64-bit
// atomic max, as a software pipeline U visits deep — visit t uses slot t % U, whose previous occupant (visit t - U) is
// settled first. A hint or key load therefore has U whole iterations to arrive, and because the loop is unrolled U times
// every slot is a fixed set of registers: no copies that would have to wait for a load. A stale or lost hint only costs
// an extra atomic, never a wrong result. H is the hint type: unsigned short = 16-bit fixed point (depth_q16; half the
// cache footprint, but every visit within 2^-14 of the best depth passes stage 1), uint32_t = the sortable image of the
// f32 depth itself (only true improvements and exact ties pass: 3x fewer waves have to wait for a stage-2 key load).
// The host picks by image size.
template <bool DEPTH, uint32_t U, typename H>
struct DepthPipe {
    static constexpr bool kWide = sizeof(H) == 4;
    H* zhint;
    unsigned long long* key;
    uint32_t lo_base;
    HintQuant hq;         // narrow hints: the quantiser (wave-uniform)
    HintTile ht;          // narrow hints: their layout (wave-uniform)
    // Stage-1 slot of a visit that waits for its hint. Per visit only what the filter itself needs is computed: the wide
    // hint is the f32 depth itself and stage 1 is ONE float compare (the hints start at the smallest float above -1.0 (nextafter(-1, +inf) = 0xBF7FFFFF), so
    // `z >= hint` is the reference's strict `z > -1.0` while nobody has been there, and NaN fails); the sortable key, the
    // visit ordinal and the -0.0 fix are only formed for the ~1 % that pass. (Round 2 formed all of them for every visit:
    // six vector instructions on the consumer wave of k_iterate_split, which is that kernel's critical path.)
    bool pv[U];           // wide hints: the visit exists; narrow hints: it is a candidate (z > -1)
    uint32_t p_idx[U], p_zf[U], p_hint[U], p_q[U], n_sent, n_pass;
    uint32_t p_t[U];      // wave-uniform: the visit's iteration
    bool gv[U];           // stage-2 candidate, waiting for the chip-wide key
    uint32_t g_idx[U], g_q[U];
    unsigned long long g_mine[U], g_cur[U];

    // where pixel idx keeps its narrow hint (HintTile): four full-rate instructions, the identity for mask2 = 0
    __device__ __forceinline__ uint32_t hint_index(uint32_t idx) const {
        if (kWide) return idx;
        return bfi(ht.mask2, bfi(ht.mask1, idx >> ht.shift1, idx << 3), idx);
    }
    __device__ __forceinline__ void depth_init(H* zhint_, unsigned long long* key_, uint32_t lo_base_, const uint32_t* hint_range_, const HintTile& tile_) {
        ht = tile_;
        zhint = zhint_;
        key = key_;
        lo_base = lo_base_;
        hq = hint_quant(kWide ? nullptr : hint_range_);
        n_sent = n_pass = 0;
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            pv[k] = gv[k] = false;
            p_idx[k] = p_zf[k] = p_hint[k] = p_q[k] = p_t[k] = 0;
            g_idx[k] = g_q[k] = 0;
            g_mine[k] = g_cur[k] = 0;
        }
    }

    // Depth candidates go through two filters before they cost a global atomic (the chip retires only ~2.1e10 of those
    // per second):
    //   stage 1  this XCD's private hint (L2-resident, loaded U visits ahead);
    //   stage 2  the chip-wide 64-bit key itself, read at device scope U visits after stage 1 passed: the atomic is sent
    //            only if this visit beats what ANY XCD has sent — and the private hint learns the chip-wide depth on the way.
    // k is a compile-time constant after unrolling.
#ifdef SAR_EXPERIMENT_PROF  // wave-cycles of settle_depth: [0] stage 2 (key wait, atomic, hint store), [1] stage 1 (hint wait, compare, key load)
    unsigned long long dprof[2] = {0, 0};
#define SAR_DMARK(i, t0) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_readcyclecounter(); dprof[i] += n_ - t0; t0 = n_; } while (0)
#else
#define SAR_DMARK(i, t0)
#endif
    __device__ __forceinline__ void settle_depth(uint32_t k) {
#ifdef SAR_EXPERIMENT_PROF
        // (this build measures how long the memory takes, not how well the pipeline hides it: first everything in flight is
        // waited for under [1]'s clock — the hint and key loads of U steps ago —, then each stage's own work is timed)
        unsigned long long t0_ = __builtin_readcyclecounter();
#endif
        if (gv[k]) {
            if (g_mine[k] > g_cur[k]) {
                atomicMax(key + g_idx[k], g_mine[k]);
                ++n_sent;
            }
            const uint32_t seen = (uint32_t)(g_cur[k] >> 32);  // 0 while nobody has sent this pixel
            uint32_t learnt;
            if (kWide) {  // the hint is the depth as f32: the larger of this visit's and the chip-wide best
                const float mine = __uint_as_float(g_q[k]), theirs = sortable_f32(seen);
                learnt = (seen && theirs > mine) ? __float_as_uint(theirs) : g_q[k];
            } else {
                const uint32_t qs = seen ? depth_q16(sortable_f32(seen), hq) : 0u;
                learnt = qs > g_q[k] ? qs : g_q[k];
            }
            zhint[hint_index(g_idx[k])] = (H)learnt;
        }
        SAR_DMARK(0, t0_);
        // p_hint is the raw dword holding this pixel's hint and its neighbour's: it is unpacked only HERE, U visits
        // after the load was issued. (Unpacking next to the load makes the compiler wait for the load right there.)
        if (kWide) {
            gv[k] = pv[k] && __uint_as_float(p_zf[k]) >= __uint_as_float(p_hint[k]);
        } else {
            const uint32_t hint = (p_idx[k] & 1u) ? (p_hint[k] >> 16) : (p_hint[k] & 0xFFFFu);
            gv[k] = pv[k] && p_q[k] >= hint;
        }
        if (gv[k]) {
            ++n_pass;  // statistic: visits that passed stage 1 (one key load each)
            const float zc = __uint_as_float(p_zf[k]) + 0.0f;  // -0.0 -> +0.0: integer order == float order
            g_idx[k] = p_idx[k];
            g_q[k] = kWide ? __float_as_uint(zc) : p_q[k];
            // visit ordinal = job*n + t; the key's low word is 0xFFFFFFFF - ordinal so that the EARLIEST visit wins a tie
            g_mine[k] = ((unsigned long long)f32_sortable(zc) << 32) | (unsigned long long)(lo_base - p_t[k]);
            g_cur[k] = __hip_atomic_load(key + p_idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SAR_DMARK(1, t0_);
    }

    // Settles the candidate of visit t - U (its hint was requested U whole steps ago) and files this visit's. Returns
    // whether the hint of this visit's pixel is wanted.
    __device__ __forceinline__ bool depth_candidate(uint32_t k, bool inb, uint32_t idx, float zf, uint32_t t) {
        if (!DEPTH) return false;
        settle_depth(k);
        p_idx[k] = idx;
        p_zf[k] = __float_as_uint(zf);
        p_t[k] = t;
        if (kWide) {
            pv[k] = inb;  // z > -1 is what stage 1's compare against the hint says (:693, :821)
            return inb;
        }
        // strict `>` against the initial -1.0 (:693, :821); NaN fails
        const bool cand = inb && zf > -1.0f;
        p_q[k] = depth_q16(zf + 0.0f, hq);
        pv[k] = cand;
        return cand;
    }

    // The hint load is the LAST vector-memory operation of a step: the counter the hardware offers for "has my load
    // returned" (vmcnt) counts operations in issue order, so anything issued after a load that is still wanted in flight
    // would have to be waited for as well.
    __device__ __forceinline__ void depth_request(uint32_t k, bool cand, uint32_t idx) {
        // every lane loads (the others from entry 0): a load under the candidates' exec mask instead costs 8 % at 4096^2,
        // where the hints miss the L2 — the partial register write makes the previous load of that register a dependency
        // (the tiling leaves bit 0 of the index alone: the pixel's hint is the half of the dword its own parity names)
        if (DEPTH) p_hint[k] = *(const uint32_t*)(zhint + (cand ? (kWide ? idx : (hint_index(idx) & ~1u)) : 0u));
    }

    // After the last visit: settle what is in flight.
    __device__ __forceinline__ void depth_drain(uint32_t lane, unsigned long long* stats) {
        if (!DEPTH) return;
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            settle_depth(k);  // moves the slot's stage-1 candidate to stage 2
            pv[k] = false;
        }
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) settle_depth(k);  // settles it
        uint32_t tot = n_sent, pas = n_pass;  // statistics: depth atomics issued / stage-1 passers of this wave
        for (int off = 32; off > 0; off >>= 1) {
            tot += __shfl_down(tot, off);
            pas += __shfl_down(pas, off);
        }
        if (lane == 0 && tot) atomicAdd(stats + 1, (unsigned long long)tot);
        if (lane == 0 && pas) atomicAdd(stats + 14, (unsigned long long)pas);
    }
};

// ---------------------------------------------------------------------------------------------------
// PoolStager — the staging, with the copy-out taken off the per-iteration path
// ---------------------------------------------------------------------------------------------------
// A wave fills 64 / R buffers per iteration, so whatever a filling lane does runs in most iterations with 2-3 of 64 lanes
// active. Here a full buffer is only SWAPPED against a spare one:
//   * the staged chunk already has its final form in LDS: {previous chunk of this (wave, bin), n, R x u16};
//   * ctl[bin] = (LDS address of the bin's current buffer << 7) | fill: one ds_add_rtn hands out the slot AND names the
//     buffer, so swapping a buffer is one more atomic add on that word;
//   * a ring of P = 16 entries holds the spare buffers: the lane that fills a buffer takes the next chunk number c of
//     the wave (ballot + mbcnt), exchanges ring[c % P] (a free buffer) against its full one — which thereby becomes
//     "pending chunk c" — writes the header of the NEXT chunk of this list into the fresh buffer (its predecessor is c)
//     and re-points ctl[bin]: three LDS operations, one of them returning, no copy, no list-head array;
//   * when the ring is full (every ~7 iterations) the WHOLE wave copies the pending chunks out: lane l moves quad l % 4
//     of pending chunk l / 4 — one 16-byte LDS read and one 16-byte store per lane for 16 chunks, to consecutive
//     addresses of the wave's arena (1 KiB runs) — and the buffers are free again where they stand in the ring.
// Records that overflow a buffer within one slot request (several lanes, same bin, across the R boundary) are placed
// in the new buffer by a generation loop (rare).
template <bool DEPTH, uint32_t R, uint32_t U, typename H>
struct PoolStager : DepthPipe<DEPTH, U, H> {
    using DepthPipe<DEPTH, U, H>::depth_init;
    using DepthPipe<DEPTH, U, H>::depth_candidate;
    using DepthPipe<DEPTH, U, H>::depth_request;
    using DepthPipe<DEPTH, U, H>::depth_drain;
    static constexpr uint32_t CB = kPoolChunkBytes(R);   // staged chunk == final chunk: 8-byte header + R records
    static constexpr uint32_t Q = CB / 16u;              // 16-byte quads per chunk
    static constexpr uint32_t P = kPoolSpareOf(R);       // spare buffers == most chunks that wait for the copy-out
    static constexpr uint32_t kFillBits = 7u, kFillMask = 127u;  // fill < R + 64 <= 92
    uint32_t* ctl;        // [B] (LDS address of the records of the bin's buffer << 7) | fill
    uint32_t* ring;       // [P] LDS addresses (records) of the spare / pending buffers
    uint32_t pool0;       // LDS address of buffer 0's records
    uint32_t lane, n_bins;
    uint4* arena;         // this wave's chunk arena
    uint32_t cursor;      // wave-uniform: next chunk number
    uint32_t drained;     // wave-uniform: chunks below this one are in the arena
    BinMap map;
    uint32_t bin_bits_v;  // map.bin_bits, held in a vector register
    bool b_have;          // previous visit, waiting for its LDS slot
    uint32_t b_bin, b_old, b_local;
#ifdef SAR_EXPERIMENT_PROF
    // timing experiment: wave-cycles per segment of the loop body (s_memtime; every mark drains lgkmcnt, so the LDS round
    // trips that normally overlap the next segment are charged to the segment that issued them)
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0;
    __device__ __forceinline__ void mark(int i) {
        asm volatile("" ::: "memory");
        const unsigned long long now = __builtin_readcyclecounter();
        asm volatile("" ::: "memory");
        prof[i] += now - prof_last;
        prof_last = now;
    }
#define SAR_MARK(i) this->mark(i)
#else
#define SAR_MARK(i)
#endif

    static __device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
    static __device__ __forceinline__ char* lds_ptr(uint32_t a) { return (char*)(__attribute__((address_space(3))) char*)(uintptr_t)a; }

    __device__ __forceinline__ void init(char* wbase, uint32_t bins, uint32_t lane_, uint4* arena_, H* zhint_,
                                         unsigned long long* key_, const BinMap& map_, uint32_t lo_base_, const uint32_t* hint_range_, const HintTile& tile_) {
        n_bins = bins;
        lane = lane_;
        // layout: buffers (bins + P) * CB | ctl bins | ring P (lanes without a visit are masked off, not redirected)
        char* pool = wbase;
        ctl = (uint32_t*)(wbase + (bins + P) * CB);
        ring = ctl + bins;
        pool0 = lds_addr(pool) + 8u;
        for (uint32_t b = lane; b < bins; b += 64u) {
            ctl[b] = (pool0 + b * CB) << kFillBits;
            *(uint2*)(pool + b * CB) = make_uint2(kNoChunk, R);  // header of the list's first chunk: no predecessor
        }
        if (lane < P) ring[lane] = pool0 + (bins + lane) * CB;
        arena = arena_;
        cursor = drained = 0;
        depth_init(zhint_, key_, lo_base_, hint_range_, tile_);
        map = map_;
        bin_bits_v = map_.bin_bits;
        asm volatile("" : "+v"(bin_bits_v));  // stays in a VGPR: v_bfe_u32 takes one scalar operand, the field offset
        b_have = false;
        b_bin = b_old = b_local = 0;
    }

    // The whole wave copies the pending chunks [drained, cursor) out — at most P of them (16; 8 of the 128-byte chunks), so
    // one pass: lane l moves quad l % G of pending chunk l / G (G = 2 / 4 / 8 lanes per chunk).
    __device__ __forceinline__ void drain_all() {
        constexpr uint32_t G = kChunkLanes(R);
        static_assert(P <= 64u / G, "one pass must cover the ring");
        const uint32_t q = lane % G;
        const uint32_t e = drained + lane / G;
        if ((int32_t)(cursor - e) > 0 && q < Q) {
            const uint32_t rec = ring[e % P];
            const u32x4 v = *(const u32x4*)lds_ptr(rec - 8u + q * 16u);
            __builtin_nontemporal_store(v, (u32x4*)(arena + (size_t)e * kChunkStride(R)) + q);
        }
        drained = cursor;
    }

    // One round of swaps: the lanes with `mine` (at most P of them, ranked 0.. by `rank`) have just filled the buffer at
    // LDS address `rec` of bin `bin`.
    __device__ __forceinline__ void swap_round(bool mine, uint32_t rank, uint32_t count, uint32_t bin, uint32_t rec) {
        if (cursor - drained + count > P) drain_all();
        if (mine) {
            // The header of a staged chunk — {previous chunk of this (wave, bin) list, R} — is written when its buffer is
            // INSTALLED, by the lane that filled the bin's previous buffer: that lane knows the predecessor's number (its
            // own). So a fill costs one returning LDS operation (the ring exchange), not two, and no list-head array.
            const uint32_t chunk = cursor + rank;
            const uint32_t fresh = __hip_atomic_exchange(&ring[chunk % P], rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            *(uint2*)lds_ptr(fresh - 8u) = make_uint2(chunk, R);
            // buffer address and fill live in one word: re-point the bin and take R off the fill (records that
            // overflowed in the same request keep their count)
            __hip_atomic_fetch_add(&ctl[bin], ((fresh - rec) << kFillBits) - R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        cursor = __builtin_amdgcn_readfirstlane(cursor + count);
    }
    __device__ __forceinline__ void swap_full(bool fl, unsigned long long fb, uint32_t bin, uint32_t rec) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0u));
        const uint32_t nfill = (uint32_t)__popcll(fb);
        if (nfill <= P) {
            swap_round(fl, rank, nfill, bin, rec);
        } else {  // more than P lanes fill a buffer in one request: P at a time
            for (uint32_t base = 0; base < nfill; base += P)
                swap_round(fl && rank - base < P, rank - base, nfill - base < P ? nfill - base : P, bin, rec);
        }
    }

    // Places the pending record: b_old is the control word its slot request returned.
    __device__ __forceinline__ void place_visit() {
        uint32_t slot = b_old & kFillMask;
        uint32_t rec = b_old >> kFillBits;
        const bool w0 = b_have && slot < R;
        if (w0) *(unsigned short*)lds_ptr(rec + 2u * slot) = (unsigned short)b_local;
        // lanes that took the last slot: the mask comes straight from one compare, the lanes without a visit drop out of it
        // in the scalar unit
        const unsigned long long fb = lanes_eq(slot, R - 1u) & wave_ballot(b_have);
        const bool fl = b_have && slot == R - 1u;
        if (fb) {
            swap_full(fl, fb, b_bin, rec);
            // Lanes whose slot lies beyond the buffer: it filled within this very request, and their record belongs to a later
            // generation of it. Rare — kept out of the path above, which 2 of 3 iterations take.
            if (__builtin_expect(lanes_ge(b_have ? slot : 0u, R) != 0ull, 0)) place_overflow(slot, rec);
        }
    }
    __device__ __forceinline__ void place_overflow(uint32_t slot, uint32_t rec) {
        bool over = b_have && slot >= R;
        for (;;) {
            slot -= over ? R : 0u;
            if (over) rec = ctl[b_bin] >> kFillBits;  // the buffer the swap installed
            const bool w = over && slot < R;
            if (w) *(unsigned short*)lds_ptr(rec + 2u * slot) = (unsigned short)b_local;
            const bool fl = w && slot == R - 1u;
            over = over && slot >= R;
            const unsigned long long fb = wave_ballot(fl);
            if (!fb) break;
            swap_full(fl, fb, b_bin, rec);
            if (!wave_ballot(over)) break;
        }
    }

    __device__ __forceinline__ void step(uint32_t k, bool inb, uint32_t idx, float zf, uint32_t t) {
        // the staging phase is a chain of short dependent steps with memory round trips at its end: let it win the SIMD's
        // issue arbitration against the other waves' long arithmetic phase, so that its loads start early
        __builtin_amdgcn_s_setprio(3);
        place_visit();
        SAR_MARK(2);
        const bool cand = depth_candidate(k, inb, idx, zf, t);
        SAR_MARK(3);
        b_have = inb;
        b_bin = __builtin_amdgcn_ubfe(idx, map.seg_shift, bin_bits_v);
        b_local = bfi(map.low_mask, idx, idx >> map.hi_shift);
        if (inb) b_old = atomicAdd(&ctl[b_bin], 1u);  // ds_add_rtn_u32: slot and buffer in one word
        depth_request(k, cand, idx);
        __builtin_amdgcn_s_setprio(0);
        SAR_MARK(4);
    }

    __device__ __forceinline__ void finish(uint32_t* heads, uint32_t n_waves, uint32_t wave, unsigned long long* stats) {
        place_visit();
        drain_all();
        depth_drain(lane, stats);
        // the partly filled buffers: one lane per bin writes {list head, fill, records} as the list's last chunk
        for (uint32_t b0 = 0; b0 < n_bins; b0 += 64u) {
            const uint32_t b = b0 + lane;
            const uint32_t word = (b < n_bins) ? ctl[b] : (pool0 << kFillBits);
            const uint32_t have = (b < n_bins) ? (word & kFillMask) : 0u;
            const bool flusher = have != 0u;
            const unsigned long long fb = wave_ballot(flusher);
            const uint32_t rec = word >> kFillBits;
            uint32_t head = (b < n_bins) ? *(const uint32_t*)lds_ptr(rec - 8u) : kNoChunk;  // the list's last full chunk
            if (flusher) {
                const uint32_t chunk = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0u));
                *(uint32_t*)lds_ptr(rec - 4u) = have;
                u32x4* dst = (u32x4*)(arena + (size_t)chunk * kChunkStride(R));
#pragma unroll
                for (uint32_t q = 0; q < Q; ++q)
                    __builtin_nontemporal_store(*(const u32x4*)lds_ptr(rec - 8u + q * 16u), dst + q);
                head = chunk;
            }
            cursor += (uint32_t)__popcll(fb);
            if (b < n_bins) heads[(size_t)b * n_waves + wave] = head;
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// k_warmup — the 1000 uncounted iterations every job starts with (reference src/lib.rs:750-752), and the packing of
// the survivors. NaN is absorbing: a job whose x is NaN after the warm-up spends all its counted iterations on pixel
// (0,0) without ever winning a depth test, so its n iterations go straight to the NaN counter and the
// job never occupies a lane of the hot kernel. The packed order depends on which wave's atomic lands first; results
// do not (the visit ordinal is formed from the job index, which travels in `joblist`).
// ---------------------------------------------------------------------------------------------------
constexpr int kWarmupRangeIters = 64;  // the last warm-up iterations whose depths k_warmup looks at for the hint quantiser
__device__ __forceinline__ void warmup_body(const WarmArgs& a) {
    const double* __restrict__ starts = a.starts;
    double* __restrict__ warm = a.warm;
    uint32_t* __restrict__ joblist = a.joblist;
    uint32_t* const hint_range = a.hint_range;
    const uint32_t n_jobs = a.n_jobs;
    const uint32_t slot_in = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_in = a.in_active ? *a.in_active : n_jobs;  // (second phase: what the first phase packed)
    if (blockIdx.x * blockDim.x >= n_in) return;
    const bool valid = slot_in < n_in;
    const uint32_t job = (valid && a.in_joblist) ? a.in_joblist[slot_in] : slot_in;
    const int n_iter = (int)a.n_iter;
    MapParams p = a.p;
    pin_map_params(p);
    double x = 0., y = 0., z = 0.;
    float zlo = __builtin_inff(), zhi = -__builtin_inff();  // depth range of this lane's candidates (neutral for lanes without a job)
    if (valid) {
        x = starts[slot_in];
        y = starts[n_jobs + slot_in];
        z = starts[2u * n_jobs + slot_in];
        if (!hint_range) {
            for (int w = 0; w < n_iter; ++w) next_point(p, x, y, z);
        } else {
            // the same iterations; the last ones also project the point and note the depth of every visit that could
            // win a depth test (in bounds, z > -1): the range the narrow depth hints quantise (HintQuant)
            const int n_range = n_iter < kWarmupRangeIters ? n_iter : kWarmupRangeIters;
            for (int w = 0; w < n_iter - n_range; ++w) next_point(p, x, y, z);
            for (int w = 0; w < n_range; ++w) {
                bool inb;
                uint32_t idx;
                float zf;
                iterate_once(p, a.width, x, y, z, inb, idx, zf);
                if (inb && zf > -1.0f) {
                    zlo = fminf(zlo, zf);
                    zhi = fmaxf(zhi, zf);
                }
            }
        }
    }
    if (hint_range) {  // (kernel-uniform) the wave's range: every lane takes part in the shuffles, the job-less ones with +-inf
        for (int off = 32; off > 0; off >>= 1) {
            zlo = fminf(zlo, __shfl_down(zlo, off));
            zhi = fmaxf(zhi, __shfl_down(zhi, off));
        }
        if ((threadIdx.x & 63u) == 0u && zhi >= zlo) {
            atomicMax(hint_range, ~f32_sortable(zlo + 0.0f));
            atomicMax(hint_range + 1, f32_sortable(zhi + 0.0f));
        }
    }
    const bool live = valid && x == x;
    const unsigned long long lm = wave_ballot(live), dm = wave_ballot(valid && !live);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) {
        if (lm) base = atomicAdd(a.active, (uint32_t)__popcll(lm));
        if (dm) atomicAdd(a.nan_count, a.iters * (unsigned long long)__popcll(dm));
    }
    base = __builtin_amdgcn_readfirstlane(base);
    if (live) {
        const uint32_t slot = base + rank;
        warm[slot] = x;
        warm[n_jobs + slot] = y;
        warm[2u * n_jobs + slot] = z;
        joblist[slot] = job;
    }
}
__global__ void __launch_bounds__(256) k_warmup(const WarmArgs a) { warmup_body(a); }

// ---------------------------------------------------------------------------------------------------
// Batched launches: F frames of one shape through ONE launch of each kernel (BatchFrame, sar_internal.hpp). The frame is
// blockIdx.z; its argument block is read from a table in device memory through the CONSTANT address space — the address is
// wave-uniform and nothing writes the table while a launch runs, so these are scalar loads into SGPRs, exactly what the
// by-value kernel arguments of the single-frame kernels are — and the kernel body is the single-frame body unchanged.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_warmup_batch(const BatchFrame* frames, uint32_t first_phase) {
    const WarmArgs a = load_frame_args(first_phase ? &frames[blockIdx.z].warm_first : &frames[blockIdx.z].warm);
    warmup_body(a);
}
// what a render call clears before its kernels (three small memsets per frame otherwise): survivor counter and dead-job
// iterations, the segment flags of k_bin_accumulate, the measured depth range of the narrow hints
__global__ void __launch_bounds__(256) k_batch_clear(const BatchFrame* frames) {
    const BatchFrame* f = frames + blockIdx.y;
    uint32_t* const seg = f->seg_any;
    const uint32_t n = f->seg_words;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) seg[k] = 0u;
    if (blockIdx.x == 0u && threadIdx.x < 4u) f->warm.active[threadIdx.x] = 0u;
    if (blockIdx.x == 0u && threadIdx.x < 4u && f->warm_first.n_iter) f->warm_first.active[threadIdx.x] = 0u;
    if (blockIdx.x == 0u && threadIdx.x < 2u && f->clear_hint_range) f->warm.hint_range[threadIdx.x] = 0u;
}

template <bool DEPTH, uint32_t R, uint32_t U, typename H>
__global__ void __launch_bounds__(256) k_iterate_lean(const BinIterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // lanes take the packed trajectories k_warmup left (those that survived the warm-up), not raw job indices: a
    // preset like solar-sail loses 38 % of its start points to NaN there, and they would sit in every wave as idle lanes
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = slot >> 6;
    const uint32_t active = *a.active;
    if (a.warm_nan && slot == 0u) {  // the jobs the warm-up dropped (k_warmup): their iterations all land on pixel (0,0)
        const unsigned long long dead = *a.warm_nan;
        if (dead) atomicAdd(a.nan_count, dead);
    }
    if ((slot & ~63u) >= active) {  // nothing left for this wave: publish empty lists
        for (uint32_t b = lane; b < a.n_bins; b += 64u) a.heads[(size_t)b * a.n_waves + wave] = kNoChunk;
        return;
    }
    bool alive = slot < active;
    const uint32_t job = alive ? a.joblist[slot] : 0u;
    const uint32_t n = (uint32_t)a.it.iters;

    PoolStager<DEPTH, R, U, H> st;
    // visit ordinal = job*n + t (job-major, iteration-minor == the sequential order of the reference); the key's
    // low word is 0xFFFFFFFF - ordinal so that the EARLIEST visit wins a depth tie
    st.init((char*)smem + (threadIdx.x >> 6) * kPoolWaveLds(a.n_bins, R), a.n_bins, lane,
            (uint4*)a.arena + (size_t)wave * a.chunks_per_wave * kChunkStride(R),
            (H*)a.zhint + (size_t)(xcc_id() & a.hint_copy_mask) * kHintStride(a.it.npix), a.it.scratch_key, a.map, 0xFFFFFFFFu - job * n, a.hint_range, a.tile);

    MapParams p = a.it.p;
    pin_map_params(p);
#pragma unroll
    for (int k = 0; k < 10; ++k) p.cy[k] = vgpr_pin(p.cy[k]);  // room enough: the y coefficients stay out of the (spilling) scalar file as well
    double x = 0., y = 0., z = 0.;
    if (alive) {  // the point after the warm-up (:750-752), from k_warmup
        x = a.warm[slot];
        y = a.warm[a.it.n_jobs + slot];
        z = a.warm[2u * a.it.n_jobs + slot];
    }
    const uint32_t C = a.it.ckpt_stride;  // a multiple of U (the host rounds it)
    const size_t cs = a.it.n_jobs;
    uint32_t t = 0;
    double* ck = a.it.ckpt + job;
    auto checkpoint = [&]() {  // the state BEFORE iteration t (coalesced 512-B rows per wave)
        // Diverged trajectories are looked for HERE, once per checkpoint, not in every iteration: NaN is absorbing, and
        // an iteration whose point is NaN passes the bounds test (:789) and lands on pixel (0,0) (:800-802) through the
        // ordinary record path (count is a sum: it does not matter which way an iteration is counted). From the first
        // checkpoint that sees the NaN on, the remaining iterations are added in one go.
        {
            const bool ended = alive && x != x;
            if (wave_ballot(ended)) {
                if (ended) atomicAdd(a.nan_count, (unsigned long long)(n - t));
                alive = alive && !ended;
            }
        }
        if (alive) {
            __builtin_nontemporal_store(x, ck);
            __builtin_nontemporal_store(y, ck + cs);
            __builtin_nontemporal_store(z, ck + 2 * cs);
        }
        ck += 3 * cs;
    };
    // one iteration of the loop body; k = t % U is a compile-time constant where this is instantiated
    auto iteration = [&](uint32_t k) {
        bool inb;
        uint32_t idx;
        float zf;
        iterate_once(p, a.it.width, x, y, z, inb, idx, zf);  // every lane, finished or not: no divergence
        inb = inb && alive;  // idx is only ever used under inb (slot request, hint address, depth candidate)
#ifdef SAR_EXPERIMENT_PROF
        asm volatile("" : "+v"(idx), "+v"(zf));  // the map and the projection belong to segment 0
#endif
#ifdef SAR_EXPERIMENT_PROF
        st.mark(0);
#endif
        st.step(k, inb, idx, zf, t);
        ++t;
    };
    // Whole passes of U iterations first: the pass is the unit of the depth pipeline, and a loop that contains
    // nothing else lets the compiler count exactly which loads may still be in flight at each use.
#ifdef SAR_EXPERIMENT_PROF
    st.prof_last = __builtin_readcyclecounter();
#endif
    const uint32_t n_full = n - n % U;
    while (t < n_full) {
        checkpoint();
        const uint32_t tend = (n_full - t > C) ? t + C : n_full;
        while (t < tend) {
#pragma unroll
            for (uint32_t k = 0; k < U; ++k) iteration(k);
        }
    }
    if (t < n) {  // the last n % U iterations of the job
        if (t % C == 0u) checkpoint();
#pragma unroll
        for (uint32_t k = 0; k + 1 < U; ++k)
            if (t < n) iteration(k);
    }
#ifdef SAR_EXPERIMENT_PROF
    if (lane == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(a.nan_count + 2 + i, st.prof[i ? i + 1 : i]);  // PoolStager marks 0, 2, 3, 4
#endif
    if (a.warm_out && slot < active) {  // the next segment of a > 2^32-2-iteration job starts here (a NaN state stays NaN
        a.warm_out[slot] = x;           // and is found again by that segment's first checkpoint)
        a.warm_out[a.it.n_jobs + slot] = y;
        a.warm_out[2u * a.it.n_jobs + slot] = z;
    }
    st.finish(a.heads, a.n_waves, wave, a.nan_count);
}

// ---------------------------------------------------------------------------------------------------
// k_iterate_split — the hot loop cut in two by FUNCTION: a producer wave and a consumer wave per 64 trajectories
// ---------------------------------------------------------------------------------------------------
// k_iterate_lean keeps two waves per SIMD resident (the staging buffers fill the LDS), and each wave alternates between a
// long run of dependent fp64 arithmetic and a short run of LDS / memory round trips: a quarter of the time both waves of a
// SIMD wait. Here a workgroup is TWO waves for one set of 64 trajectories: wave 0 runs the map and the projection —
// arithmetic only, it never waits for memory — and hands {pixel, depth} of every visit to wave 1 through 1 KiB of LDS
// (double-buffered, one s_barrier per iteration); wave 1 owns the staging buffers and does what k_iterate_lean's stager
// does. The LDS holds the same eight staging sets per CU, but the SIMD now has four waves to pick from, two of which are
// always ready to issue arithmetic. Same arithmetic, same visit order, same records.
// PH = iterations per barrier phase: U (2 KiB of visits in flight) or 1 (1 KiB, where the staging leaves no more).
template <bool DEPTH, uint32_t R, uint32_t U, typename H, uint32_t PH>
__device__ __forceinline__ void iterate_split_body(const BinIterArgs& a, uint32_t* smem, const uint32_t wave) {
    static_assert(PH == 1u || PH == U, "a phase is one iteration or one pass of the depth pipeline");
    const uint32_t lane = threadIdx.x & 63u;
    const bool producer = threadIdx.x < 64u;  // wave-uniform; `wave`: one set of 64 trajectories per workgroup
    const uint32_t slot = wave * 64u + lane;
    const uint32_t active = *a.active;
    if (a.warm_nan && wave == 0u && threadIdx.x == 0u) {  // as in k_iterate_lean
        const unsigned long long dead = *a.warm_nan;
        if (dead) atomicAdd(a.nan_count, dead);
    }
    if (wave * 64u >= active) {  // nothing left for this workgroup: publish empty lists
        if (!producer)
            for (uint32_t b = lane; b < a.n_bins; b += 64u) a.heads[(size_t)b * a.n_waves + wave] = kNoChunk;
        return;
    }
    bool alive = slot < active;
    const uint32_t job = alive ? a.joblist[slot] : 0u;
    const uint32_t n = (uint32_t)a.it.iters;
    // visits in flight between the two waves: [2 phases][PH visits][64 lanes] {pixel index or ~0 (no visit), depth as f32 bits};
    // one s_barrier per phase
    uint2* hand = (uint2*)((char*)smem + kPoolWaveLds(a.n_bins, R));
    const uint32_t n_full = n - n % U;  // whole phases; the last n % U iterations form a short one
    if (producer) {
        MapParams p = a.it.p;
        pin_map_params(p);
#pragma unroll
        for (int k = 0; k < 10; ++k) p.cy[k] = vgpr_pin(p.cy[k]);
        double x = 0., y = 0., z = 0.;
        if (alive) {  // the point after the warm-up (:750-752), from k_warmup
            x = a.warm[slot];
            y = a.warm[a.it.n_jobs + slot];
            z = a.warm[2u * a.it.n_jobs + slot];
        }
        const uint32_t C = a.it.ckpt_stride;  // a multiple of U (the host rounds it)
        const size_t cs = a.it.n_jobs;
        double* ck = a.it.ckpt + job;
        uint32_t t = 0;
        auto checkpoint = [&]() {  // as in k_iterate_lean: the state BEFORE iteration t; NaN is looked for here
            const bool ended = alive && x != x;
            if (wave_ballot(ended)) {
                if (ended) atomicAdd(a.nan_count, (unsigned long long)(n - t));
                alive = alive && !ended;
            }
            if (alive) {
                __builtin_nontemporal_store(x, ck);
                __builtin_nontemporal_store(y, ck + cs);
                __builtin_nontemporal_store(z, ck + 2 * cs);
            }
            ck += 3 * cs;
        };
        auto produce = [&](uint2* dst) {
            bool inb;
            uint32_t idx;
            float zf;
            iterate_once(p, a.it.width, x, y, z, inb, idx, zf);
            inb = inb && alive;
            *dst = make_uint2(inb ? idx : 0xFFFFFFFFu, __float_as_uint(zf));
            ++t;
        };
        uint32_t phase = 0;
#ifdef SAR_EXPERIMENT_PROF  // wave-cycles of the producer: [0] map + projection + hand-over write, [1] waiting at the barrier
        unsigned long long pp[2] = {0, 0}, pl = __builtin_readcyclecounter();
#define SAR_PMARK(i) do { asm volatile("" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); pp[i] += now_ - pl; pl = now_; } while (0)
#else
#define SAR_PMARK(i)
#endif
        while (t < n_full) {
            checkpoint();
            const uint32_t tend = (n_full - t > C) ? t + C : n_full;
            while (t < tend) {
#pragma unroll
                for (uint32_t h = 0; h < U / PH; ++h) {
                    uint2* dst = hand + (phase & 1u) * (PH * 64u) + lane;
#pragma unroll
                    for (uint32_t k = 0; k < PH; ++k) produce(dst + k * 64u);
                    // the visits are in LDS before the consumer is let past the barrier
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    SAR_PMARK(0);
                    asm volatile("s_barrier" ::: "memory");
                    SAR_PMARK(1);
                    ++phase;
                }
            }
        }
#ifdef SAR_EXPERIMENT_PROF
        if (lane == 0) { atomicAdd(a.nan_count + 6, pp[0]); atomicAdd(a.nan_count + 7, pp[1]); }
#endif
        if (t < n) {  // the last n % U iterations of the job
            if (t % C == 0u) checkpoint();
#pragma unroll
            for (uint32_t k = 0; k + 1 < U; ++k)  // one iteration per phase here, whatever PH
                if (t < n) {
                    produce(hand + (phase & 1u) * (PH * 64u) + lane);
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    ++phase;
                }
        }
        if (a.warm_out && slot < active) {
            a.warm_out[slot] = x;
            a.warm_out[a.it.n_jobs + slot] = y;
            a.warm_out[2u * a.it.n_jobs + slot] = z;
        }
    } else {
        PoolStager<DEPTH, R, U, H> st;
        st.init((char*)smem, a.n_bins, lane, (uint4*)a.arena + (size_t)wave * a.chunks_per_wave * kChunkStride(R),
                (H*)a.zhint + (size_t)(xcc_id() & a.hint_copy_mask) * kHintStride(a.it.npix), a.it.scratch_key, a.map, 0xFFFFFFFFu - job * n, a.hint_range, a.tile);
        uint32_t t = 0, phase = 0;
#ifdef SAR_EXPERIMENT_PROF  // wave-cycles of the consumer: [0] barrier, [1] hand-over read, [2] place_visit, [3] depth, [4] requests
        st.prof_last = __builtin_readcyclecounter();
#endif
        while (t < n_full) {
#pragma unroll
            for (uint32_t h = 0; h < U / PH; ++h) {
                // phase `phase` is in its half of `hand`; the producer writes that half again after the NEXT barrier
                asm volatile("s_barrier" ::: "memory");
#ifdef SAR_EXPERIMENT_PROF
                st.mark(0);
#endif
                const uint2* src = hand + (phase & 1u) * (PH * 64u) + lane;
                uint2 v[PH];
#pragma unroll
                for (uint32_t k = 0; k < PH; ++k) v[k] = src[k * 64u];
#ifdef SAR_EXPERIMENT_PROF
#pragma unroll
                for (uint32_t k = 0; k < PH; ++k) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
                st.mark(1);
#endif
#pragma unroll
                for (uint32_t k = 0; k < PH; ++k) {
                    st.step(h * PH + k, v[k].x != 0xFFFFFFFFu, v[k].x, __uint_as_float(v[k].y), t);
                    ++t;
                }
                ++phase;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k + 1 < U; ++k)
            if (t < n) {
                asm volatile("s_barrier" ::: "memory");
                const uint2 v = hand[(phase & 1u) * (PH * 64u) + lane];
                st.step(k, v.x != 0xFFFFFFFFu, v.x, __uint_as_float(v.y), t);
                ++t;
                ++phase;
            }
#ifdef SAR_EXPERIMENT_PROF
        if (lane == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(a.nan_count + 8 + i, st.prof[i]);
            atomicAdd(a.nan_count + 13, st.dprof[0]);
            atomicAdd(a.nan_count + 15, st.dprof[1]);
        }
#endif
        st.finish(a.heads, a.n_waves, wave, a.nan_count);
    }
}
template <bool DEPTH, uint32_t R, uint32_t U, typename H, uint32_t PH>
__global__ void __launch_bounds__(128) k_iterate_split(const BinIterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    iterate_split_body<DEPTH, R, U, H, PH>(a, smem, blockIdx.x);
}
// F frames in one launch, a 1-D grid of F x n_waves workgroups. Which (frame, wave pair) a workgroup is follows the XCDs: the
// dispatcher deals consecutive workgroups to the eight XCDs round-robin, and every frame has its own depth hints and depth
// keys — F working sets in every XCD's 4 MiB L2 if every frame ran everywhere (three frames of configs[4]: 1.0 us per
// iteration step against 0.65 alone). With xcd_map the frames are dealt to the XCDs instead: 8 / F XCDs per frame (F = 2, 4,
// 8), F / 8 frames per XCD (16, 24, ...), or — any other F of three or more — eight equal runs of consecutive wave pairs: an XCD's
// L2 sees the hints of its own frames only, one frame at a time. One or two frames whose pairs do not divide: frame after frame.
template <bool DEPTH, uint32_t R, uint32_t U, typename H, uint32_t PH>
__global__ void __launch_bounds__(128) k_iterate_split_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_waves, uint32_t xcd_map) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lin = blockIdx.x;
    uint32_t frame, wave;
    if (xcd_map == 1u) {         // 8 / F XCDs per frame
        const uint32_t xpf = 8u / n_frames, xcd = lin & 7u, pos = lin >> 3;
        frame = xcd / xpf;
        wave = pos * xpf + xcd % xpf;
    } else if (xcd_map == 2u) {  // F / 8 frames per XCD, one after the other: XCD x runs frames x, x + 8, ... — its L2 holds one frame's
        const uint32_t xcd = lin & 7u, pos = lin >> 3, turn = pos / n_waves;  // hints at a time, and its CUs never wait for a round of
        frame = xcd + 8u * turn;                                              // equally long workgroups to end before the next frame starts
        wave = pos - turn * n_waves;
    } else if (xcd_map == 3u) {  // any F: the F * n_waves wave pairs in frame order, cut into eight runs of consecutive pairs — XCD x takes
        const uint32_t total = n_frames * n_waves, xcd = lin & 7u, pos = lin >> 3;  // run x (the workgroups lin = x mod 8 are q + (x < r) many):
        const uint32_t q = total >> 3, r = total & 7u;                            // an XCD works through F / 8 frames' worth of pairs one frame
        const uint32_t g = xcd * q + (xcd < r ? xcd : r) + pos;                   // after the other, a frame lies on one or two XCDs (more below F = 8)
        frame = g / n_waves;
        wave = g - frame * n_waves;
    } else {
        frame = lin / n_waves;
        wave = lin - frame * n_waves;
    }
    const BinIterArgs a = load_frame_args(&frames[frame].it);
    iterate_split_body<DEPTH, R, U, H, PH>(a, smem, wave);
}

// ---------------------------------------------------------------------------------------------------
// k_extent — the "first pass" the reference leaves as a TODO (src/lib.rs:326-333): bounds of the attractor in screen
// space (what the comment at :329-333 lists) and in raw coordinates. One trajectory per lane: 1000 warm-up
// iterations, then `iters` iterations with 12 running bounds in registers; a bound moves through `<` / `>` only, so NaN
// never moves one and the result does not depend on the order of the reduction. Output: 12 doubles per block.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bound(double v, double& lo, double& hi) {
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
}

__global__ void __launch_bounds__(256) k_extent(const MapParams pin, const double* __restrict__ starts, uint32_t n_jobs,
                                                uint64_t iters, double* __restrict__ out) {
    __shared__ double part[4][12];
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    MapParams p = pin;
    pin_map_params(p);
    double b[12];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        b[2 * k] = __builtin_inf();
        b[2 * k + 1] = -__builtin_inf();
    }
    if (job < n_jobs) {
        double x = starts[job], y = starts[n_jobs + job], z = starts[2u * n_jobs + job];
        for (int w = 0; w < 1000; ++w) next_point(p, x, y, z);  // :750-752
        for (uint64_t t = 0; t < iters; ++t) {
            next_point(p, x, y, z);
            double sx, sy, sz;
            screen_space(p, x, y, z, sx, sy, sz);  // :773
            bound(sx, b[0], b[1]);
            bound(sy, b[2], b[3]);
            bound(sz, b[4], b[5]);
            bound(x, b[6], b[7]);
            bound(y, b[8], b[9]);
            bound(z, b[10], b[11]);
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = b[k];
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_down(v, off);
            v = (k & 1) ? (o > v ? o : v) : (o < v ? o : v);
        }
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const int k = threadIdx.x;
        double v = part[0][k];
        for (uint32_t w = 1; w < blockDim.x / 64u; ++w) {
            const double o = part[w][k];
            v = (k & 1) ? (o > v ? o : v) : (o < v ? o : v);
        }
        out[(size_t)blockIdx.x * 12u + k] = v;
    }
}

// start points [m][3] (as the ABI takes them) -> the kernel's SoA block x[m] y[m] z[m]
__global__ void __launch_bounds__(256) k_starts_soa(const double* __restrict__ aos, double* __restrict__ soa, uint32_t m) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < m) {
        soa[k] = aos[3u * k];
        soa[m + k] = aos[3u * k + 1u];
        soa[2u * m + k] = aos[3u * k + 2u];
    }
}

void launch_iterate(const IterArgs& a, uint32_t block, hipStream_t s) {
    hipLaunchKernelGGL(k_iterate, dim3((a.n_jobs + block - 1) / block), dim3(block), 0, s, a);
}

uint32_t lean_wave_lds_bytes(uint32_t bins, uint32_t records) { return kPoolWaveLds(bins, records); }
uint32_t chunk_bytes(uint32_t records) { return kChunkStride(records) * 16u; }

// The instantiations of the hot kernel are exactly the shapes the host's planning can pick (sar_plan.cpp): chunk size
// (records per chunk: 12 / 20 / 28 / 60 = 32- / 48-on-64- / 64- / 128-byte chunks) x hint type, depth pipeline of two visits;
// the wave-pair form for 64- and 128-byte chunks, with one or two iterations per barrier phase.
#define SAR_FOR_EACH_LEAN(X)                                                                                         \
    X(12u, unsigned short) X(20u, unsigned short) X(28u, unsigned short) X(60u, unsigned short)                       \
    X(12u, uint32_t) X(20u, uint32_t) X(28u, uint32_t) X(60u, uint32_t)
#define SAR_FOR_EACH_SPLIT(X)                                                                                        \
    X(28u, unsigned short, 1u) X(28u, unsigned short, 2u) X(28u, uint32_t, 1u) X(28u, uint32_t, 2u)                   \
    X(60u, unsigned short, 1u) X(60u, unsigned short, 2u) X(60u, uint32_t, 1u) X(60u, uint32_t, 2u)

int launch_iterate_lean(const BinIterArgs& a, uint32_t block, uint32_t records, uint32_t hint_bytes, bool split, hipStream_t s) {
    const uint32_t stage = lean_wave_lds_bytes(a.n_bins, records);
    bool launched = false;
    if (split) {  // producer / consumer wave pairs: one workgroup of 128 threads per launched wave (a.n_waves of them)
        const uint32_t ph = (stage + 2048u) * 8u <= 160u * 1024u ? 2u : 1u;  // visits in flight: two iterations if they fit
        const size_t lds2 = stage + ph * 1024u;
#define SAR_LAUNCH_SPLIT(RR, HH, PP)                                                                                  \
    if (!launched && records == RR && hint_bytes == sizeof(HH) && ph == PP) {                                          \
        hipLaunchKernelGGL((k_iterate_split<true, RR, 2u, HH, PP>), dim3(a.n_waves), dim3(128), lds2, s, a);           \
        launched = true;                                                                                              \
    }
        SAR_FOR_EACH_SPLIT(SAR_LAUNCH_SPLIT)
#undef SAR_LAUNCH_SPLIT
        return launched ? 0 : 1;
    }
    const uint32_t grid = (a.it.n_jobs + block - 1) / block;
    const size_t lds = (size_t)(block / 64u) * stage;
#define SAR_LAUNCH_LEAN(RR, HH)                                                                                       \
    if (!launched && records == RR && hint_bytes == sizeof(HH)) {                                                      \
        hipLaunchKernelGGL((k_iterate_lean<true, RR, 2u, HH>), dim3(grid), dim3(block), lds, s, a);                    \
        launched = true;                                                                                              \
    }
    SAR_FOR_EACH_LEAN(SAR_LAUNCH_LEAN)
#undef SAR_LAUNCH_LEAN
    return launched ? 0 : 1;
}

// how k_iterate_split_batch deals F frames of n_waves wave pairs to the XCDs: 1 = 8 / F XCDs per frame (F = 1, 2, 4, 8), 2 = F / 8
// frames per XCD (F = 16, 24, ...), 3 = any other F of three frames or more: eight equal runs of consecutive wave pairs (13 frames:
// 1.625 frames' worth per XCD — the tail of a sweep is dealt like its full batches), 0 = frame after frame on every XCD (what is
// left: one or two frames whose pairs do not divide)
uint32_t batch_xcd_map(uint32_t n_frames, uint32_t n_waves) {
    if (n_frames <= 8u && 8u % n_frames == 0u && n_waves % (8u / n_frames) == 0u) return 1;
    if (n_frames % 8u == 0u) return 2;
    if (n_frames >= 3u) return 3;
    return 0;
}

int launch_iterate_split_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_waves, uint32_t n_bins, uint32_t records,
                               uint32_t hint_bytes, uint32_t xcd_map, hipStream_t s) {
    const uint32_t total = n_frames * n_waves;
    const uint32_t stage = lean_wave_lds_bytes(n_bins, records);
    const uint32_t ph = (stage + 2048u) * 8u <= 160u * 1024u ? 2u : 1u;
    const size_t lds2 = stage + ph * 1024u;
    bool launched = false;
#define SAR_LAUNCH_SPLIT_BATCH(RR, HH, PP)                                                                                        \
    if (!launched && records == RR && hint_bytes == sizeof(HH) && ph == PP) {                                                      \
        hipLaunchKernelGGL((k_iterate_split_batch<true, RR, 2u, HH, PP>), dim3(total), dim3(128), lds2, s, frames, n_frames, n_waves, xcd_map); \
        launched = true;                                                                                                          \
    }
    SAR_FOR_EACH_SPLIT(SAR_LAUNCH_SPLIT_BATCH)
#undef SAR_LAUNCH_SPLIT_BATCH
    return launched ? 0 : 1;
}

int iterate_kernel_attributes() {
    // the staging buffers need more dynamic LDS than the 64 KiB default window
    hipError_t e = hipSuccess;
#define SAR_ATTR_LEAN(RR, HH) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_iterate_lean<true, RR, 2u, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SAR_FOR_EACH_LEAN(SAR_ATTR_LEAN)
#undef SAR_ATTR_LEAN
#define SAR_ATTR_SPLIT(RR, HH, PP) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_iterate_split<true, RR, 2u, HH, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SAR_FOR_EACH_SPLIT(SAR_ATTR_SPLIT)
#undef SAR_ATTR_SPLIT
#define SAR_ATTR_SPLIT_BATCH(RR, HH, PP) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_iterate_split_batch<true, RR, 2u, HH, PP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SAR_FOR_EACH_SPLIT(SAR_ATTR_SPLIT_BATCH)
#undef SAR_ATTR_SPLIT_BATCH
    return (int)e;
}

uint32_t launch_extent(const MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* out, hipStream_t s) {
    const uint32_t blocks = (n_jobs + 255u) / 256u;
    hipLaunchKernelGGL(k_extent, dim3(blocks), dim3(256), 0, s, p, starts, n_jobs, iters, out);
    return blocks;
}

void launch_starts_soa(const double* aos, double* soa, uint32_t m, hipStream_t s) {
    hipLaunchKernelGGL(k_starts_soa, dim3((m + 255u) / 256u), dim3(256), 0, s, aos, soa, m);
}

// a later segment of a long job: the jobs that died in the warm-up (never packed) spend this segment's iterations on
// pixel (0,0) as well
__global__ void k_dead_jobs(const uint32_t* active, uint32_t n_jobs, uint64_t iters, unsigned long long* nan_count) {
    const uint32_t dead = n_jobs - *active;
    if (dead) atomicAdd(nan_count, iters * (unsigned long long)dead);
}
void launch_dead_jobs(const uint32_t* active, uint32_t n_jobs, uint64_t iters, unsigned long long* nan_count, hipStream_t s) {
    hipLaunchKernelGGL(k_dead_jobs, dim3(1), dim3(1), 0, s, active, n_jobs, iters, nan_count);
}

void launch_warmup(const WarmArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_warmup, dim3((a.n_jobs + 255u) / 256u), dim3(256), 0, s, a);
}
void launch_warmup_batch(const BatchFrame* frames, uint32_t n_frames, uint32_t n_jobs, bool first_phase, hipStream_t s) {
    hipLaunchKernelGGL(k_warmup_batch, dim3((n_jobs + 255u) / 256u, 1, n_frames), dim3(256), 0, s, frames, first_phase ? 1u : 0u);
}
// The start points of a batch, page-locked host memory -> device memory, by a kernel instead of the copy engine (whose queue
// holds the previous frames' read-backs) and instead of the warm-up kernel reading them in place: 1.5 MB per frame over PCIe
// take longer than the first warm-up phase computes, and a warm-up wave that waits for PCIe holds registers nothing else can
// use — these few waves hold almost none, and the other lane's kernels run beside them.
__global__ void __launch_bounds__(256) k_batch_fetch(const BatchFrame* frames) {
    const BatchFrame* f = frames + blockIdx.y;
    const u32x4* __restrict__ src = (const u32x4*)f->starts_host;
    u32x4* __restrict__ dst = (u32x4*)f->starts_dev;
    const uint32_t n = f->n_start_quads;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) dst[k] = src[k];
}
void launch_batch_fetch(const BatchFrame* frames, uint32_t n_frames, hipStream_t s) {
    hipLaunchKernelGGL(k_batch_fetch, dim3(32, n_frames), dim3(256), 0, s, frames);
}
void launch_batch_clear(const BatchFrame* frames, uint32_t n_frames, uint32_t seg_words, hipStream_t s) {
    hipLaunchKernelGGL(k_batch_clear, dim3((seg_words + 255u) / 256u, n_frames), dim3(256), 0, s, frames);
}

}  // namespace sar
