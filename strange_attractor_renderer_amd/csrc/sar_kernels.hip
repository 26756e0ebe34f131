// sar_kernels.hip — gfx950 (MI355X) kernels of the iterate/accumulate path.
//
// Bit-exactness contract: every floating-point operation below is the reference's operation, in the
// reference's order, with separate multiply and add (the file is compiled with -ffp-contract=off and
// the build checks that k_iterate contains no v_fma_f64). The chaotic map amplifies a 1-ulp
// deviation exponentially, so "close" does not exist here: either the op sequence is identical or
// the images differ.
//
// Kernels (DESIGN.md section 3 has the measurements behind every choice)
//   k_warmup              the 1000 uncounted iterations of every job, and the packing of the jobs that survive them.
//   k_iterate_lean<DEPTH,R,U,H>  the hot loop: one surviving trajectory per lane, fp64 state in registers; a visit
//                         becomes a 2-byte record staged per (wave, bin) in LDS and copied out in chunks of R records
//                         to the wave's arena; the depth test goes through per-XCD hints (type H) and a pipeline U
//                         visits deep, so that only ~0.6 % of the visits send the 64-bit key atomic
//                         (sortable(z as f32) << 32 | ~ordinal); trajectory checkpoints every `ckpt_stride` iterations.
//   k_bin_accumulate<R>   walks the per-(bin, wave) chunk lists and counts the records in an LDS histogram per bin.
//   k_iterate<XCD_LOCAL,MODE>  the first correct version — one no-return global atomic add and one atomic max per
//                         visit — kept for images beyond the binned path's 32 Mpx and as an A/B reference.
//   k_fold_resolve        folds the scratch bins into the persistent Runtime buffers (count add, running max, depth
//                         test with "earlier visit wins ties"), re-zeroes the scratch, compacts the pixels whose depth
//                         winner changed (LDS) and recomputes their colour-transform payload from the nearest
//                         checkpoint (the visit ordinal in the key names job and iteration).
//   k_merge, k_colorize_gas, k_zrange + k_colorize_depth, exchange pack/unpack, k_convert (export formats),
//   k_extent (attractor bounds), k_reset, k_zbuf_in/out, k_starts_soa.
#include <hip/hip_runtime.h>

#include "sar_internal.hpp"

#pragma STDC FP_CONTRACT OFF
#pragma clang fp contract(off)

namespace sar {

// ---------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_sortable(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float sortable_f32(uint32_t s) {
    const uint32_t b = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
    return __uint_as_float(b);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The narrow depth hints are 16-bit fixed point: q(z) = clamp(floor((z + 1) * 2^14), 0, 65535). Every step is monotone
// non-decreasing in z, so q(a) < q(b) implies a < b: a visit whose q is below the stored q of an already-sent visit
// cannot win the depth test. Half the footprint of the 32-bit (sortable f32) hints, at the price of passing every
// visit within 2^-14 of the best depth: the host picks the type by the view's footprint (sar_runtime.cpp).
__device__ __forceinline__ uint32_t depth_q16(float zf) {
    const float s = (zf + 1.0f) * 16384.0f;
    const uint32_t q = (uint32_t)fminf(fmaxf(s, 0.0f), 65535.0f);
    return q;
}

// Forces a wave-uniform value into a VGPR (opaque to the optimiser, no instruction emitted).
__device__ __forceinline__ double vgpr_pin(double v) {
    asm volatile("" : "+v"(v));
    return v;
}

// PolynomialSprott2Degree::next_point (reference src/lib.rs:583-621).
// sum = ((((c0 + x*c1) + x²*c2) + xy*c3) + ... + z²*c9), strictly left to right, no FMA.
// (`0. + 1.*c0` is exactly c0 once the host has canonicalised a -0.0 coefficient to +0.0.)
__device__ __forceinline__ void next_point(const MapParams& p, double& x, double& y, double& z) {
    const double xx = x * x;
    const double xy = x * y;
    const double xz = x * z;
    const double yy = y * y;
    const double yz = y * z;
    const double zz = z * z;
    double sx = p.cx[0], sy = p.cy[0], sz = p.cz[0];
    sx = sx + x * p.cx[1];  sy = sy + x * p.cy[1];  sz = sz + x * p.cz[1];
    sx = sx + xx * p.cx[2]; sy = sy + xx * p.cy[2]; sz = sz + xx * p.cz[2];
    sx = sx + xy * p.cx[3]; sy = sy + xy * p.cy[3]; sz = sz + xy * p.cz[3];
    sx = sx + xz * p.cx[4]; sy = sy + xz * p.cy[4]; sz = sz + xz * p.cz[4];
    sx = sx + y * p.cx[5];  sy = sy + y * p.cy[5];  sz = sz + y * p.cz[5];
    sx = sx + yy * p.cx[6]; sy = sy + yy * p.cy[6]; sz = sz + yy * p.cz[6];
    sx = sx + yz * p.cx[7]; sy = sy + yz * p.cy[7]; sz = sz + yz * p.cz[7];
    sx = sx + z * p.cx[8];  sy = sy + z * p.cy[8];  sz = sz + z * p.cz[8];
    sx = sx + zz * p.cx[9]; sy = sy + zz * p.cy[9]; sz = sz + zz * p.cz[9];
    x = sx; y = sy; z = sz;
}

// Matrix3x3::mul_right (src/lib.rs:205-216): (m0*x + m1*y) + m2*z per row.
__device__ __forceinline__ void screen_space(const MapParams& p, double x, double y, double z,
                                             double& sx, double& sy, double& sz) {
    sx = p.m[0] * x + p.m[1] * y + p.m[2] * z;
    sy = p.m[3] * x + p.m[4] * y + p.m[5] * z;
    sz = p.m[6] * x + p.m[7] * y + p.m[8] * z;
}

// color transforms (src/lib.rs:507-516, 520-558); only evaluated for depth winners.
__device__ __forceinline__ double color_transform(const ColorTransformParams& ct, double dx, double dy,
                                                  double dz, double sx, double sy, double sz) {
    const double mag = sqrt(dx * dx + dy * dy + dz * dz);  // Vec3::magnitude, :129-131
    if (ct.kind == SAR_CT_ADJUSTED_VELOCITY) {
        return (mag + ct.offset) * ct.factor;  // :514
    }
    const double COS = 0.7009092642998509;  // literal at :530
    const double SIN = 0.7132504491541816;  // literal at :536
    const double x2 = (sx + ct.ccx) * COS + (sz + ct.ccy) * SIN;  // :538-539
    double part = 1.;
    if (x2 < -0.0839 || 10.55 * x2 + sy < 0.46 - 1.0941 || 1.0426 * x2 + sy < 0.179 - 0.1576 ||
        0.5139 * x2 - sy > -0.04 - 0.04092) {
        part = 0.;
    }
    const double color = (part + mag) / 2.;  // :556
    return (color - 0.1) / 0.9;              // :557
}

// ---------------------------------------------------------------------------------------------------
// k_iterate — the hot loop (render, src/lib.rs:747-838)
// ---------------------------------------------------------------------------------------------------
template <bool XCD_LOCAL>
__device__ __forceinline__ void bin_count(uint32_t* addr, uint32_t v) {
    if (XCD_LOCAL) {
        // this scratch copy is only ever touched by CUs of ONE XCD (copy index = hardware XCC id),
        // so the XCD's own L2 is a sufficient coherence point: workgroup scope keeps the atomic in L2.
        __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_fetch_add(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <bool XCD_LOCAL>
__device__ __forceinline__ void bin_key(unsigned long long* addr, unsigned long long v) {
    if (XCD_LOCAL) {
        __hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_fetch_max(addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ uint32_t xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): id 20, offset 0, size 4 -> simm16 = (3<<11)|(0<<6)|20
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
}

template <bool XCD_LOCAL, int MODE>
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

    // "skip first 1000 to get good values in the attractor" (:750-752)
    for (int w = 0; w < 1000; ++w) next_point(p, x, y, z);

    uint32_t* const count = a.scratch_count + (XCD_LOCAL ? (size_t)xcc_id() * a.npix : 0);
    unsigned long long* const key = a.scratch_key + (XCD_LOCAL ? (size_t)xcc_id() * a.npix : 0);

    const uint32_t n = (uint32_t)a.iters;
    // visit ordinal = job*n + t (job-major, iteration-minor == the sequential order of the reference);
    // the key's low word is 0xFFFFFFFF - ordinal so that the EARLIEST visit wins a depth tie.
    const uint32_t lo_base = 0xFFFFFFFFu - job * n;
    const uint32_t C = a.ckpt_stride;
    const size_t cs = a.n_jobs;  // checkpoint component stride

    uint32_t t = 0;
    double* ck = a.ckpt + job;
    while (t < n) {
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
                if (MODE != 0) bin_count<XCD_LOCAL>(count, n - t);
                return;
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
            if (MODE != 0) bin_count<XCD_LOCAL>(count + idx, 1u);  // :807-812
            if (MODE == 2) {
                float zf = (float)z2;  // `z2 as f32`
                // strict `>` against an initial -1.0 (:693, :821): z <= -1 and NaN can never win
                if (zf > -1.0f) {
                    zf = zf + 0.0f;  // -0.0 -> +0.0 so the integer order agrees with the float order
                    const unsigned long long k =
                        ((unsigned long long)f32_sortable(zf) << 32) | (unsigned long long)(lo_base - t);
                    bin_key<XCD_LOCAL>(key + idx, k);
                }
            }
        }
    }
    if (MODE == 0) {  // measurement-only variant: keep the arithmetic alive
        if (x + y + z == 12345.678) a.scratch_count[0] = 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_iterate_lean — the hot loop without a global atomic per visit
// ---------------------------------------------------------------------------------------------------
// Measured on MI355X: the chip retires ~2.1e10 scattered global atomics per second whatever their
// scope or width, while the fp64 arithmetic of this loop alone runs at ~3.3e11 iterations/s. So a
// visit must not cost a global atomic. Here every visit becomes a 2-byte RECORD instead:
//
//   * the image is cut into B bins of 2^bin_shift consecutive pixels; a record is the pixel's offset
//     inside its bin (u16);
//   * each WAVE owns B staging buffers of R records in LDS (2R + 8 bytes per bin: records, counter, link);
//     a visit takes a slot with one LDS atomic (ds_add_rtn) and writes its u16 there one iteration later;
//   * the lane that takes the last slot copies the R records + {link to the previous chunk of this
//     (wave, bin), count} as ONE chunk to the wave's private arena in HBM — position from a
//     wave-local cursor, so no global atomic and nothing to wait for — and resets the buffer;
//   * k_bin_accumulate later walks the per-(bin, wave) chunk lists and histograms them in LDS.
//
// Depth: see Stager::settle_depth — two filters (this XCD's hint, then the chip-wide key) in front of the
// 64-bit atomic max, as a software pipeline U visits deep. A stale or lost hint only costs an extra atomic,
// never a wrong result.
//
// Control flow: the per-visit operations are issued unconditionally with a select on the ADDRESS instead
// of a branch (every `if` around an LDS or memory operation costs s_and_saveexec / s_cbranch_execz / s_or):
//   * a lane without a visit requests its slot from a private dummy counter and writes its record to
//     a private scratch slot (cnt[B + lane], rec[B*R + lane]);
//   * the hint of a lane without a depth candidate is loaded from element 0;
//   * only the rare-per-lane events keep a wave-level branch: "some lane filled a buffer" (copy-out,
//     which also places the records that overflowed into the next buffer generation), "some lane passed
//     a depth filter", "a trajectory ended in NaN".
// LDS per wave: B buffers of R records + B counters + B links + 64 scratch records + 64 dummy counters
// per-XCD hint arrays: an even number of entries each, so that the dword holding a 16-bit hint is aligned
__host__ __device__ constexpr size_t kHintStride(uint32_t npix) { return ((size_t)npix + 1u) & ~(size_t)1u; }
constexpr uint32_t kLeanWaveLds(uint32_t bins, uint32_t R) { return bins * (2u * R + 8u) + 384u; }
constexpr uint32_t kChunkQuads(uint32_t R) { return (8u + 2u * R) / 16u; }  // 16-byte quads of data per chunk: R = 12, 20, 28
// Chunks never straddle a 64-byte sector of the arena: the 48-byte chunk (R = 20) is laid out on a 64-byte stride.
// The accumulate kernel's chunk reads are isolated, and an isolated read moves whole sectors (measured, tools/ubench/
// gather_runs.hip: 3.4 TB/s for 64-byte pieces) — a straddling chunk would cost two.
constexpr uint32_t kChunkStride(uint32_t R) { return R == 12u ? 2u : 4u; }

// a * b for operands below 2^24 as ONE full-rate instruction. (Through the __umul24 builtin the optimiser knows the
// operand ranges, turns the product back into a 32-bit multiply and picks quarter-rate v_mul_lo_u32 / v_mad_u64_u32.)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t wave_uniform_b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(wave_uniform_b), "v"(a));  // src0 may be scalar, src1 is a VGPR
    return r;
}

// Lane mask of a predicate. (HIP's __ballot goes through an integer: v_cndmask + v_cmp_ne per call.)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Everything a wave needs to turn a stream of visits into staged records + depth candidates. One visit
// per lane per step(); all per-visit state lives in registers, the staging buffers in the wave's LDS slice.
// H is the hint type: unsigned short = 16-bit fixed point (depth_q16; half the cache footprint, but every visit within
// 2^-14 of the best depth passes stage 1), uint32_t = the sortable image of the f32 depth itself (only true improvements
// and exact ties pass: 3x fewer waves have to wait for a stage-2 key load). The host picks by image size.
template <bool DEPTH, uint32_t R, uint32_t U, typename H>
struct Stager {
    static constexpr bool kWide = sizeof(H) == 4;
    static constexpr uint32_t Q = kChunkQuads(R);  // 16-byte quads per chunk
    unsigned short* rec;  // [B][R] staged records + 64 scratch slots
    uint32_t* cnt;        // [B] fill counters + 64 dummy counters
    uint32_t* prv;        // [B] previous chunk of this (wave, bin) list
    uint32_t trash, dummy, lane, n_bins;
    uint4* arena;         // this wave's chunk arena
    uint32_t cursor;      // wave-uniform: next free chunk
    H* zhint;
    unsigned long long* key;
    uint32_t bin_shift, bin_mask, lo_base;
    // The depth path is a software pipeline U visits deep: visit t uses slot t % U, whose previous occupant (visit
    // t - U) is settled first. A hint or key load therefore has U whole iterations to arrive, and because the loop
    // is unrolled U times every slot is a fixed set of registers: no copies that would have to wait for a load.
    bool pv[U];           // stage-1 candidate, waiting for its hint
    uint32_t p_idx[U], p_zkey[U], p_lo[U], p_hint[U], p_q[U], n_sent;
    bool gv[U];           // stage-2 candidate, waiting for the chip-wide key
    uint32_t g_idx[U], g_q[U];
    unsigned long long g_mine[U], g_cur[U];
    bool b_have;          // previous visit, waiting for its LDS slot
    uint32_t b_bin, b_slot, b_local;
#ifdef SAR_EXPERIMENT_PROF
    // timing experiment: wave-cycles per segment of the loop body (s_memtime; every mark drains lgkmcnt, so the LDS
    // round trips that normally overlap the next segment are charged to the segment that issued them)
    unsigned long long prof[4] = {0, 0, 0, 0}, prof_last = 0;
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
    bool f_on;            // a filled buffer whose 2R bytes sit in registers, waiting to be stored
    uint32_t f_chunk, f_prev;
    uint2 fpend[R / 4u];

    __device__ __forceinline__ void init(char* wbase, uint32_t bins, uint32_t lane_, uint4* arena_, H* zhint_,
                                         unsigned long long* key_, uint32_t shift, uint32_t lo_base_) {
        n_bins = bins;
        lane = lane_;
        rec = (unsigned short*)wbase;
        cnt = (uint32_t*)(wbase + bins * 2u * R + 128u);
        prv = cnt + bins + 64u;
        for (uint32_t b = lane; b < bins + 64u; b += 64u) cnt[b] = 0u;
        for (uint32_t b = lane; b < bins; b += 64u) prv[b] = kNoChunk;
        trash = bins * R + lane;
        dummy = bins + lane;
        arena = arena_;
        cursor = 0;
        zhint = zhint_;
        key = key_;
        bin_shift = shift;
        bin_mask = (1u << shift) - 1u;
        lo_base = lo_base_;
        b_have = f_on = false;
        n_sent = 0;
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            pv[k] = gv[k] = false;
            p_idx[k] = p_zkey[k] = p_lo[k] = p_hint[k] = p_q[k] = 0;
            g_idx[k] = g_q[k] = 0;
            g_mine[k] = g_cur[k] = 0;
        }
        b_bin = b_slot = b_local = 0;
        f_chunk = f_prev = 0;
#pragma unroll
        for (uint32_t k = 0; k < R / 4u; ++k) fpend[k] = make_uint2(0u, 0u);
    }

    // one chunk: {previous chunk of this (wave, bin), record count, records}
    __device__ __forceinline__ void store_chunk(uint32_t chunk, uint32_t prev, uint32_t count, const uint2 (&f)[R / 4u]) {
        uint32_t w[4u * Q];
        w[0] = prev;
        w[1] = count;
#pragma unroll
        for (uint32_t k = 0; k < R / 4u; ++k) {
            w[2u + 2u * k] = f[k].x;
            w[3u + 2u * k] = f[k].y;
        }
        u32x4* dst = (u32x4*)(arena + (size_t)chunk * kChunkStride(R));
#pragma unroll
        for (uint32_t q = 0; q < Q; ++q)  // streamed once, read once: keep them out of the L2 the hints live in
            __builtin_nontemporal_store((u32x4){w[4u * q], w[4u * q + 1u], w[4u * q + 2u], w[4u * q + 3u]}, dst + q);
    }
    // immediate copy-out (rare path)
    __device__ __forceinline__ void flush_full(uint32_t bin, uint32_t chunk) {
        const uint2* r = (const uint2*)(rec + mul24(bin, R));  // 2R bytes, 8-byte aligned
        uint2 f[R / 4u];
#pragma unroll
        for (uint32_t k = 0; k < R / 4u; ++k) f[k] = r[k];
        store_chunk(chunk, prv[bin], R, f);
        prv[bin] = chunk;
        __hip_atomic_fetch_sub(&cnt[bin], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // The common copy-out is split: the lane that filled a buffer ISSUES the LDS reads (and frees the buffer:
    // LDS executes a wave's operations in order, so later writes cannot overtake them); the global stores go
    // out later in the step, when the reads have long returned.
    __device__ __forceinline__ void flush_store_pending() {
        if (f_on) store_chunk(f_chunk, f_prev, R, fpend);
        f_on = false;
    }

    // Places the pending record. slot = R*gen + pos: slots are handed out consecutively per bin, so the
    // quotient says which refill generation of the R-record buffer a record belongs to. The generation-0
    // write is unconditional (scratch slot for lanes without one); everything else only exists when some lane
    // filled a buffer in the same slot request.
    __device__ __forceinline__ void place_visit() {
        // slot < R + 64 (a counter is below R whenever a slot request finds it), so slot / R is exact through a
        // 24-bit multiply: full-rate v_mul_u32_u24 / v_mad_u32_u24 instead of the quarter-rate 32-bit multiplies
        constexpr uint32_t kInvR = (65536u + R - 1u) / R;
        const uint32_t gen = mul24(b_slot, kInvR) >> 16;
        const uint32_t pos = b_slot - mul24(gen, R);
        const uint32_t base = mul24(b_bin, R);
        const uint32_t at = base + pos;
        const bool w0 = b_have && gen == 0u;
        rec[w0 ? at : trash] = (unsigned short)b_local;
        const bool fl = w0 && pos == R - 1u;
        const unsigned long long fb = wave_ballot(fl);
        if (fb) {
            if (fl) {
                const uint2* r = (const uint2*)(rec + base);
#pragma unroll
                for (uint32_t k = 0; k < R / 4u; ++k) fpend[k] = r[k];
                f_chunk = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0u));
                f_prev = prv[b_bin];
                f_on = true;
                prv[b_bin] = f_chunk;
                __hip_atomic_fetch_sub(&cnt[b_bin], R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            cursor += (uint32_t)__popcll(fb);
            // records that overflowed into the next generation of a buffer that was just emptied
            const bool e1 = b_have && gen == 1u && pos < R - 1u;
            rec[e1 ? at : trash] = (unsigned short)b_local;
            bool pend = b_have && gen >= 1u && !e1;  // a later generation's last slot, or generation >= 2: rare
            for (uint32_t g = 1; wave_ballot(pend); ++g) {
                const bool mine = pend && gen == g;
                if (mine) rec[at] = (unsigned short)b_local;
                const bool fl2 = mine && pos == R - 1u;
                const unsigned long long fb2 = wave_ballot(fl2);
                if (fb2) {
                    if (fl2) flush_full(b_bin, cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(fb2 >> 32),
                                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)fb2, 0u)));
                    cursor += (uint32_t)__popcll(fb2);
                }
                const bool early = pend && gen == g + 1u && pos < R - 1u;
                if (early) rec[at] = (unsigned short)b_local;
                pend = pend && !(mine || early);
            }
        }
    }

    // Depth candidates go through two filters before they cost a global atomic (the chip retires only
    // ~2.1e10 of those per second):
    //   stage 1  this XCD's private 16-bit hint (L2-resident, loaded U visits ahead);
    //   stage 2  the chip-wide 64-bit key itself, read at device scope U visits after stage 1 passed
    //            (~5 % of the visits): the atomic is sent only if this visit beats what ANY XCD has sent —
    //            and the private hint learns the chip-wide depth on the way.
    // k is a compile-time constant after unrolling.
    __device__ __forceinline__ void settle_depth(uint32_t k) {
        if (gv[k]) {
            if (g_mine[k] > g_cur[k]) {
                atomicMax(key + g_idx[k], g_mine[k]);
                ++n_sent;
            }
            const uint32_t seen = (uint32_t)(g_cur[k] >> 32);  // 0 while nobody has sent this pixel
            const uint32_t qs = kWide ? seen : (seen ? depth_q16(sortable_f32(seen)) : 0u);
            zhint[g_idx[k]] = (H)(qs > g_q[k] ? qs : g_q[k]);
        }
        // p_hint is the raw dword holding this pixel's hint and its neighbour's: it is unpacked only HERE, U visits
        // after the load was issued. (Unpacking next to the load makes the compiler wait for the load right there.)
        const uint32_t hint = kWide ? p_hint[k] : ((p_idx[k] & 1u) ? (p_hint[k] >> 16) : (p_hint[k] & 0xFFFFu));
        gv[k] = pv[k] && p_q[k] >= hint;
        if (gv[k]) {
            g_idx[k] = p_idx[k];
            g_q[k] = p_q[k];
            g_mine[k] = ((unsigned long long)p_zkey[k] << 32) | (unsigned long long)p_lo[k];
            g_cur[k] = __hip_atomic_load(key + p_idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // One visit of this lane: inb = the iteration landed inside the image at pixel idx with depth zf
    // (reference src/lib.rs:807-834); t = iteration number (for the visit ordinal); k = t % U, a compile-time
    // constant after unrolling.
    __device__ __forceinline__ void step(uint32_t k, bool inb, uint32_t idx, float zf, uint32_t t) {
        // the staging phase is a chain of short dependent steps with memory round trips at its end: let it win the
        // SIMD's issue arbitration against the other waves' long arithmetic phase, so that its loads start early
        __builtin_amdgcn_s_setprio(3);
        // the previous visit's record first: pure LDS work
        place_visit();
        SAR_MARK(1);
        bool cand = false;
        if (DEPTH) {
            settle_depth(k);  // the candidate of visit t - U: its hint was requested U whole steps ago
            // this visit's candidate: strict `>` against the initial -1.0 (:693, :821); NaN fails
            cand = inb && zf > -1.0f;
            const float zc = zf + 0.0f;  // -0.0 -> +0.0: integer order == float order
            p_zkey[k] = f32_sortable(zc);
            p_q[k] = kWide ? p_zkey[k] : depth_q16(zc);
            p_idx[k] = idx;
            p_lo[k] = lo_base - t;
            pv[k] = cand;
        }
        SAR_MARK(2);
        // chunk stores of a buffer that filled up (their LDS reads were issued by place_visit above), then this
        // visit's slot request
        flush_store_pending();
        b_have = inb;
        b_bin = idx >> bin_shift;
        b_local = idx & bin_mask;
        b_slot = atomicAdd(&cnt[inb ? b_bin : dummy], 1u);  // ds_add_rtn_u32
        // the hint load is the LAST vector-memory operation of the step: the counter the hardware offers for "has
        // my load returned" (vmcnt) counts operations in issue order, so anything issued after a load that is still
        // wanted in flight would have to be waited for as well
        if (DEPTH) p_hint[k] = *(const uint32_t*)(zhint + (cand ? (kWide ? idx : (idx & ~1u)) : 0u));
        __builtin_amdgcn_s_setprio(0);
        SAR_MARK(3);
    }

    // After the last visit: settle what is in flight, flush the partly filled buffers, publish the list heads.
    __device__ __forceinline__ void finish(uint32_t* heads, uint32_t n_waves, uint32_t wave, unsigned long long* stats) {
        place_visit();
        flush_store_pending();
        if (DEPTH) {
#pragma unroll
            for (uint32_t k = 0; k < U; ++k) {
                settle_depth(k);  // moves the slot's stage-1 candidate to stage 2
                pv[k] = false;
            }
#pragma unroll
            for (uint32_t k = 0; k < U; ++k) settle_depth(k);  // settles it
            uint32_t tot = n_sent;  // statistics: depth atomics issued by this wave
            for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
            if (lane == 0 && tot) atomicAdd(stats + 1, (unsigned long long)tot);
        }
        for (uint32_t b0 = 0; b0 < n_bins; b0 += 64u) {
            const uint32_t b = b0 + lane;
            const uint32_t have = (b < n_bins) ? cnt[b] : 0u;
            const bool flusher = have != 0u;
            const unsigned long long fb = wave_ballot(flusher);
            uint32_t head = (b < n_bins) ? prv[b] : kNoChunk;
            if (flusher) {
                const uint32_t chunk = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(fb >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)fb, 0u));
                const uint2* r = (const uint2*)(rec + b * R);
                uint2 f[R / 4u];
#pragma unroll
                for (uint32_t k = 0; k < R / 4u; ++k) f[k] = r[k];
                store_chunk(chunk, head, have, f);
                head = chunk;
            }
            cursor += (uint32_t)__popcll(fb);
            if (b < n_bins) heads[(size_t)b * n_waves + wave] = head;
        }
    }
};

// One iteration of render's loop body up to the visit (reference src/lib.rs:770-802), without a branch: advances the
// point and reports whether the iteration passes the bounds test, the pixel it would land on and its depth as f32.
// The caller masks the result for lanes whose trajectory has ended (NaN is absorbing: x != x after this call).
__device__ __forceinline__ void iterate_once(const MapParams& p, uint32_t width, double& x, double& y, double& z,
                                             bool& inb, uint32_t& idx, float& zf) {
    next_point(p, x, y, z);  // :770
    double sx, sy, sz;
    screen_space(p, x, y, z, sx, sy, sz);  // :773
    const double ax = sx + p.ccx;          // center_camera.x with screen_space.x
    const double az = sz + p.ccy;          // center_camera.y with screen_space.z (:776-779)
    const double x2 = ax * p.cos_v + az * p.sin_v;
    const double z2 = ax * p.sin_v - az * p.cos_v;
    const double fi = (p.scale_adjusted_mid - x2) * p.width_scaled;   // :783
    const double fj = p.half_height - (sy + p.ccz) * p.width_scaled;  // :786
    // :789 — `|` instead of `||`: four compares and three mask ORs, not four nested branches
    inb = !((int)(fi >= p.width) | (int)(fj >= p.height) | (int)(fi < 0.) | (int)(fj < 0.));
    const uint32_t i = (fi == fi) ? (uint32_t)fi : 0u;  // Rust `as u32`: NaN -> 0 (non-finite coordinates pass :789)
    const uint32_t j = (fj == fj) ? (uint32_t)fj : 0u;
    idx = __umul24(j, width) + i;  // v_mad_u32_u24 (full rate); exact for every in-bounds (i, j): width, height < 2^24
    zf = (float)z2;  // `z2 as f32`
}

__device__ __forceinline__ void pin_map_params(MapParams& p) {
    // 30 coefficients + 9 matrix entries + 10 projection constants are 98 SGPRs as kernel arguments — more than
    // the scalar file holds next to pointers and exec masks (the compiler then spills SGPRs to VGPR lanes inside
    // the loop). The x/y coefficients stay scalar operands; the rest is pinned into (plentiful) VGPRs.
#pragma unroll
    for (int k = 0; k < 10; ++k) p.cz[k] = vgpr_pin(p.cz[k]);
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
}

// ---------------------------------------------------------------------------------------------------
// k_warmup — the 1000 uncounted iterations every job starts with (reference src/lib.rs:750-752), and the packing of
// the survivors. NaN is absorbing: a job whose x is NaN after the warm-up spends all its counted iterations on pixel
// (0,0) without ever winning a depth test, so its n iterations go straight to the NaN counter and the
// job never occupies a lane of the hot kernel. The packed order depends on which wave's atomic lands first; results
// do not (the visit ordinal is formed from the job index, which travels in `joblist`).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_warmup(const MapParams pin, const double* __restrict__ starts, uint32_t n_jobs,
                                                uint64_t iters, double* __restrict__ warm, uint32_t* __restrict__ joblist,
                                                uint32_t* active, unsigned long long* nan_count) {
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = job < n_jobs;
    MapParams p = pin;
    pin_map_params(p);
    double x = 0., y = 0., z = 0.;
    if (valid) {
        x = starts[job];
        y = starts[n_jobs + job];
        z = starts[2u * n_jobs + job];
        for (int w = 0; w < 1000; ++w) next_point(p, x, y, z);
    }
    const bool live = valid && x == x;
    const unsigned long long lm = wave_ballot(live), dm = wave_ballot(valid && !live);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u));
    uint32_t base = 0;
    if ((threadIdx.x & 63u) == 0u) {
        if (lm) base = atomicAdd(active, (uint32_t)__popcll(lm));
        if (dm) atomicAdd(nan_count, iters * (unsigned long long)__popcll(dm));
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

template <bool DEPTH, uint32_t R, uint32_t U, typename H>
__global__ void __launch_bounds__(256) k_iterate_lean(const BinIterArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t lane = threadIdx.x & 63u;
    // lanes take the packed trajectories k_warmup left (those that survived the warm-up), not raw job indices: a
    // preset like solar-sail loses 38 % of its start points to NaN there, and they would sit in every wave as idle lanes
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = slot >> 6;
    const uint32_t active = *a.active;
    if ((slot & ~63u) >= active) {  // nothing left for this wave: publish empty lists
        for (uint32_t b = lane; b < a.n_bins; b += 64u) a.heads[(size_t)b * a.n_waves + wave] = kNoChunk;
        return;
    }
    bool alive = slot < active;
    const uint32_t job = alive ? a.joblist[slot] : 0u;
    const uint32_t n = (uint32_t)a.it.iters;

    Stager<DEPTH, R, U, H> st;
    // visit ordinal = job*n + t (job-major, iteration-minor == the sequential order of the reference); the key's
    // low word is 0xFFFFFFFF - ordinal so that the EARLIEST visit wins a depth tie
    st.init((char*)smem + (threadIdx.x >> 6) * kLeanWaveLds(a.n_bins, R), a.n_bins, lane,
            (uint4*)a.arena + (size_t)wave * a.chunks_per_wave * kChunkStride(R),
            (H*)a.zhint + (size_t)xcc_id() * kHintStride(a.it.npix), a.it.scratch_key, a.bin_shift, 0xFFFFFFFFu - job * n);

    MapParams p = a.it.p;
    pin_map_params(p);
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
        const bool ended = alive && x != x;
        if (wave_ballot(ended)) {
            // absorbing NaN state: this and all remaining iterations pass the bounds test (:789), land on pixel
            // (0,0) (:800-802) and never win the depth test — add them in one go
            if (ended) atomicAdd(a.nan_count, (unsigned long long)(n - t));
            alive = alive && !ended;
        }
        inb = inb && alive;
        idx = inb ? idx : 0u;
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
        for (int i = 0; i < 4; ++i) atomicAdd(a.nan_count + 2 + i, st.prof[i]);
#endif
    st.finish(a.heads, a.n_waves, wave, a.nan_count);
}

// ---------------------------------------------------------------------------------------------------
// k_bin_accumulate — records -> per-pixel hit counts, in LDS
// ---------------------------------------------------------------------------------------------------
// grid (B, splits): block (b, s) owns bin b and the waves w with w % splits == s. Every thread walks
// whole (bin, wave) chunk lists (64-byte loads, newest chunk first) and adds the records into the
// bin's LDS histogram with LDS atomics; the histogram is then written — plainly, fully — as copy s of
// the scratch count bins, which k_fold_resolve sums into Runtime::count.
template <uint32_t R>
__global__ void __launch_bounds__(1024) k_bin_accumulate(const BinAccArgs a) {
    constexpr uint32_t Q = kChunkQuads(R);       // 16-byte quads per chunk
    constexpr uint32_t G = Q == 2u ? 2u : 4u;    // lanes that share one list: lane q of a group reads quad q
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const uint32_t b = blockIdx.x, s = blockIdx.y;
    const uint32_t bin_px = 1u << a.bin_shift;
    const uint32_t q = threadIdx.x % G;
    const uint32_t group = threadIdx.x / G, groups = blockDim.x / G;
    // Most bins of a frame are empty (the attractor covers a band of the image): a block with no chunk at all
    // leaves its partial histogram untouched — the scratch copies are all-zero between launches because
    // k_fold_resolve clears what it reads.
    int any = 0;
    for (uint32_t w = s + a.splits * threadIdx.x; w < a.n_waves; w += a.splits * blockDim.x)
        any |= a.heads[(size_t)b * a.n_waves + w] != kNoChunk;
    if (!__syncthreads_or(any)) return;
    for (uint32_t k = threadIdx.x; k < bin_px; k += blockDim.x) hist[k] = 0u;
    __syncthreads();
    const uint4* arena = (const uint4*)a.arena;
    // One (bin, wave) list per group of G lanes: a chunk is ONE 16-byte load per lane and one cache line per
    // group (a lane that walked a list alone needed Q loads over 64 different lines per wave instruction).
    for (uint32_t w = s + a.splits * group; w < a.n_waves; w += a.splits * groups) {
        uint32_t chunk = a.heads[(size_t)b * a.n_waves + w];
        const uint4* base = arena + (size_t)w * a.chunks_per_wave * kChunkStride(R);
        while (chunk != kNoChunk) {
            uint4 v = make_uint4(kNoChunk, 0u, 0u, 0u);
            if (q < Q) v = base[(size_t)chunk * kChunkStride(R) + q];
            // the chunk header {previous chunk of the list, record count} sits in lane 0's quad
            const uint32_t prev = __shfl(v.x, 0, G);
            const uint32_t nrec = __shfl(v.y, 0, G);
            // records held by this lane: lane 0 -> records 0..3 (its .z/.w), lane q -> 8q-4 .. 8q+3
            const uint32_t first = q == 0u ? 0u : 8u * q - 4u;
            const uint32_t w0 = q == 0u ? v.z : v.x, w1 = q == 0u ? v.w : v.y;
            if (first < nrec) atomicAdd(&hist[w0 & 0xFFFFu], 1u);
            if (first + 1u < nrec) atomicAdd(&hist[w0 >> 16], 1u);
            if (first + 2u < nrec) atomicAdd(&hist[w1 & 0xFFFFu], 1u);
            if (first + 3u < nrec) atomicAdd(&hist[w1 >> 16], 1u);
            if (q != 0u) {
                if (first + 4u < nrec) atomicAdd(&hist[v.z & 0xFFFFu], 1u);
                if (first + 5u < nrec) atomicAdd(&hist[v.z >> 16], 1u);
                if (first + 6u < nrec) atomicAdd(&hist[v.w & 0xFFFFu], 1u);
                if (first + 7u < nrec) atomicAdd(&hist[v.w >> 16], 1u);
            }
            chunk = prev;
        }
    }
    __syncthreads();
    const uint32_t px0 = b << a.bin_shift;
    uint32_t* out = a.scratch_count + (size_t)s * a.npix;
    for (uint32_t k = threadIdx.x; k < bin_px; k += blockDim.x)
        if (px0 + k < a.npix) out[px0 + k] = hist[k];
}

// ---------------------------------------------------------------------------------------------------
// block-level reductions (result valid in thread 0)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* s_tmp /* [4] */) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(v, off);
        v = o > v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) v = s_tmp[w] > v ? s_tmp[w] : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t block_min_u32(uint32_t v, uint32_t* s_tmp /* [4] */) {
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(v, off);
        v = o < v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) s_tmp[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) v = s_tmp[w] < v ? s_tmp[w] : v;
    }
    return v;
}
// raise scalars[slot] to at least m (one lane); skips the atomic when the slot is already there
__device__ __forceinline__ void raise_scalar(uint32_t* slot, uint32_t m) {
    if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, m);
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

__global__ void __launch_bounds__(256) k_fold_resolve(const FoldArgs a) {
    __shared__ unsigned long long s_key[FOLD_PIX];
    __shared__ uint32_t s_pix[FOLD_PIX];
    __shared__ uint32_t s_n, s_wrap;
    __shared__ uint32_t s_tmp[4];
    if (threadIdx.x == 0) { s_n = 0; s_wrap = 0; }
    __syncthreads();

    uint32_t local_max = 0;
    const uint32_t base = blockIdx.x * FOLD_PIX;
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

// ---------------------------------------------------------------------------------------------------
// state management
// ---------------------------------------------------------------------------------------------------
__global__ void k_reset(uint32_t* count, unsigned long long* key, double* steps, uint32_t npix,
                        uint32_t* scalars) {
    const unsigned long long init = ((unsigned long long)f32_sortable(-1.0f) << 32) | 0xFFFFFFFFull;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        count[p] = 0u;   // :687
        steps[p] = 0.;   // :690
        key[p] = init;   // zbuf = -1.0, :693
    }
    if (blockIdx.x == 0 && threadIdx.x < SC_COUNT) scalars[threadIdx.x] = 0u;  // max = 0, :694
}

__global__ void k_zbuf_out(const unsigned long long* key, float* out, uint32_t npix) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x)
        out[p] = sortable_f32((uint32_t)(key[p] >> 32));
}

__global__ void k_zbuf_in(const float* z, unsigned long long* key, uint32_t npix) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x)
        key[p] = ((unsigned long long)f32_sortable(z[p] + 0.0f) << 32) | 0xFFFFFFFFull;
}

// Runtime::merge (:708-738)
__global__ void __launch_bounds__(256) k_merge(uint32_t* count, unsigned long long* key, double* steps,
                                               const uint32_t* ocount, const unsigned long long* okey,
                                               const double* osteps, uint32_t npix, uint32_t* scalars) {
    uint32_t local_max = 0;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const uint32_t merged = count[p] + ocount[p];  // wrapping, :719
        count[p] = merged;
        local_max = merged > local_max ? merged : local_max;  // :721-723
        const unsigned long long ok = okey[p];
        if ((uint32_t)(ok >> 32) > (uint32_t)(key[p] >> 32)) {  // strict: self wins ties, :728
            steps[p] = osteps[p];
            key[p] = ok;
        }
    }
    __shared__ uint32_t s_tmp[4];
    const uint32_t m = block_max_u32(local_max, s_tmp);
    if (threadIdx.x == 0 && m) raise_scalar(&scalars[SC_MAX], m);
}

// ---------------------------------------------------------------------------------------------------
// colorize (:841-904)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t as_u16(double v) {  // Rust `as u16`: saturating, NaN -> 0
    if (!(v == v)) return 0;
    if (v <= 0.) return 0;
    if (v >= 65535.) return 65535;
    return (uint16_t)(uint32_t)v;
}
__device__ __forceinline__ uint16_t as_u16_f32(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.f) return 0;
    if (v >= 65535.f) return 65535;
    return (uint16_t)(uint32_t)v;
}

// ln(c) for an integer-valued u32 c: table of host-libm values where it exists (bit-identical to the
// oracle/reference on the same host), device log beyond it (<= 1 ulp).
__device__ __forceinline__ double ln_u32(uint32_t c, const double* lut, uint32_t lut_len) {
    const uint32_t k = c - 1u;  // c == 0 (u32 wrap of count+1) -> huge index -> log(0) = -inf
    return (k < lut_len) ? lut[k] : log((double)c);
}

__global__ void __launch_bounds__(256) k_colorize_gas(const uint32_t* count, const double* steps,
                                                      const uint32_t* scalars, const double* lut,
                                                      uint32_t lut_len, const PaletteParams pal,
                                                      double b_offset, double b_factor, int transparent,
                                                      uint32_t npix, ushort4* out) {
    __shared__ double s_pal[(SAR_PALETTE_MAX + 1) * 3];
    for (uint32_t k = threadIdx.x; k < (pal.len + 1) * 3; k += blockDim.x) s_pal[k] = pal.rgb[k / 3][k % 3];
    __syncthreads();
    const uint32_t rmax = scalars[SC_WRAP] ? 0xFFFFFFFFu : scalars[SC_MAX];
    const double ln_base = ln_u32(rmax + 1u, lut, lut_len);  // ln(max + 1), :860
    const double count_f64 = (double)pal.len;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        // Palette::interpolate (:442-472)
        double v = steps[p];
        if (v < 0.) v = 0.;
        else if (v >= 1.) v = 0.999999;
        v = v * count_f64;
        const double fl = floor(v);
        uint32_t n = (fl == fl) ? (uint32_t)fl : 0u;
        if (n >= pal.len) n = pal.len - 1;  // unreachable for non-NaN
        const double t = v - fl;            // == v % 1. for v >= 0 (exact)
        const double t1 = 1.0 - t;
        const double* c1 = &s_pal[n * 3];
        const double* c2 = &s_pal[(n + 1) * 3];
        const double r = sqrt(c2[0] * t + c1[0] * t1);
        const double g = sqrt(c2[1] * t + c1[1] * t1);
        const double b = sqrt(c2[2] * t + c1[2] * t1);
        // factor = ln(count+1) / ln(max+1)  (f64::log(self, base), :860)
        const double factor = ln_u32(count[p] + 1u, lut, lut_len) / ln_base;
        ushort4 o;
        o.x = as_u16((r * factor + b_offset) * b_factor * 65535.);
        o.y = as_u16((g * factor + b_offset) * b_factor * 65535.);
        o.z = as_u16((b * factor + b_offset) * b_factor * 65535.);
        o.w = transparent ? as_u16(factor * 65535.) : (uint16_t)65535;
        out[p] = o;
    }
}

// fold (max, min) over zbuf != -1.0 with seeds (0.0, f32::MAX) (:877-882); the sortable image turns
// f32 max/min into u32 atomics.
__global__ void k_zrange_init(uint32_t* scalars) {
    scalars[SC_ZMAX] = f32_sortable(0.0f);
    scalars[SC_ZMIN] = f32_sortable(3.40282346638528859811704183484516925e+38f);
}
__global__ void __launch_bounds__(256) k_zrange(const unsigned long long* key, uint32_t npix, uint32_t* scalars) {
    const uint32_t unset = f32_sortable(-1.0f);
    uint32_t mx = f32_sortable(0.0f);
    uint32_t mn = f32_sortable(3.40282346638528859811704183484516925e+38f);
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const uint32_t s = (uint32_t)(key[p] >> 32);
        if (s != unset) {
            mx = s > mx ? s : mx;
            mn = s < mn ? s : mn;
        }
    }
    __shared__ uint32_t s_tmp[4];
    mx = block_max_u32(mx, s_tmp);
    mn = block_min_u32(mn, s_tmp);
    if (threadIdx.x == 0) {
        atomicMax(&scalars[SC_ZMAX], mx);
        atomicMin(&scalars[SC_ZMIN], mn);
    }
}
__global__ void __launch_bounds__(256) k_colorize_depth(const unsigned long long* key, const uint32_t* scalars,
                                                        uint32_t npix, ushort4* out) {
    const float zmax = sortable_f32(scalars[SC_ZMAX]);
    const float zmin = sortable_f32(scalars[SC_ZMIN]);
    const float diff = zmax - zmin;  // :883
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        float z = sortable_f32((uint32_t)(key[p] >> 32));
        if (z == -1.0f) z = 0.0f;
        else z = __fdiv_rn(z - zmin, diff);  // f32 reverse lerp, :893
        const uint16_t v = as_u16_f32(z * 65535.0f);
        ushort4 o;
        o.x = v; o.y = v; o.z = v; o.w = 65535;
        out[p] = o;
    }
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU exchange (merge folded in rank order, expressed as MAX / SUM reductions)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long exch_key(unsigned long long key, uint32_t rank) {
    const unsigned long long k = (key & 0xFFFFFFFF00000000ull) | (unsigned long long)(0xFFFFFFFFu - rank);
    return (long long)(k ^ 0x8000000000000000ull);  // unsigned order -> signed order
}
__global__ void k_exch_export(const unsigned long long* key, uint32_t rank, long long* out, uint32_t npix) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x)
        out[p] = exch_key(key[p], rank);
}
__global__ void k_exch_select(const uint32_t* count, const unsigned long long* key, const double* steps,
                              uint32_t rank, const long long* reduced, int* out, uint32_t npix) {
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        out[p] = (int)count[p];
        const bool mine = exch_key(key[p], rank) == reduced[p];
        const unsigned long long bits = mine ? (unsigned long long)__double_as_longlong(steps[p]) : 0ull;
        out[(size_t)npix + 2 * (size_t)p] = (int)(uint32_t)bits;
        out[(size_t)npix + 2 * (size_t)p + 1] = (int)(uint32_t)(bits >> 32);
    }
}
__global__ void __launch_bounds__(256) k_exch_import(uint32_t* count, unsigned long long* key, double* steps,
                                                     const long long* reduced, const int* sum, uint32_t npix,
                                                     uint32_t* scalars) {
    uint32_t local_max = 0;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        const uint32_t c = (uint32_t)sum[p];
        count[p] = c;
        local_max = c > local_max ? c : local_max;
        const unsigned long long k = (unsigned long long)reduced[p] ^ 0x8000000000000000ull;
        key[p] = k | 0xFFFFFFFFull;
        const unsigned long long bits = (unsigned long long)(uint32_t)sum[(size_t)npix + 2 * (size_t)p] |
                                        ((unsigned long long)(uint32_t)sum[(size_t)npix + 2 * (size_t)p + 1] << 32);
        steps[p] = __longlong_as_double((long long)bits);
    }
    __shared__ uint32_t s_tmp[4];
    const uint32_t m = block_max_u32(local_max, s_tmp);
    if (threadIdx.x == 0 && m) raise_scalar(&scalars[SC_MAX], m);
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers (called from sar_runtime.cpp)
// ---------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t n, uint32_t block, uint32_t cap) {
    uint32_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    return g ? g : 1;
}

// ---------------------------------------------------------------------------------------------------
// k_convert — RGBA16 -> RGB16 / RGBA8 / RGB8 (src/bin/main.rs:52-57: DynamicImage::to_rgb16 / to_rgba8 / to_rgb8).
// image 0.25's channel conversion u16 -> u8 is ((c + 128) / 257) (rounding, exact inverse of c * 257); alpha is
// dropped, not pre-multiplied. Streaming: 8 B/px in, 3-6 B/px out; four pixels per thread keep stores 4-byte aligned.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t to8(uint32_t c16) { return (c16 + 128u) / 257u; }

template <int FORMAT>
__global__ void __launch_bounds__(256) k_convert(const ushort4* __restrict__ in, void* __restrict__ out, uint32_t npix) {
    const uint32_t quads = (npix + 3u) / 4u;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < quads; g += gridDim.x * blockDim.x) {
        const uint32_t p0 = 4u * g;
        ushort4 px[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) px[k] = (p0 + k < npix) ? in[p0 + k] : make_ushort4(0, 0, 0, 0);
        const bool full = p0 + 4u <= npix;
        if (FORMAT == SAR_FMT_RGB16) {
            unsigned short* o = (unsigned short*)out + (size_t)p0 * 3u;
            if (full) {  // 12 u16 = three 8-byte stores
                uint2* o2 = (uint2*)o;
                o2[0] = make_uint2(px[0].x | ((uint32_t)px[0].y << 16), px[0].z | ((uint32_t)px[1].x << 16));
                o2[1] = make_uint2(px[1].y | ((uint32_t)px[1].z << 16), px[2].x | ((uint32_t)px[2].y << 16));
                o2[2] = make_uint2(px[2].z | ((uint32_t)px[3].x << 16), px[3].y | ((uint32_t)px[3].z << 16));
            } else {
                for (uint32_t k = 0; p0 + k < npix; ++k) {
                    o[3u * k] = px[k].x;
                    o[3u * k + 1u] = px[k].y;
                    o[3u * k + 2u] = px[k].z;
                }
            }
        } else if (FORMAT == SAR_FMT_RGBA8) {
            uint32_t w[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) w[k] = to8(px[k].x) | (to8(px[k].y) << 8) | (to8(px[k].z) << 16) | (to8(px[k].w) << 24);
            uint32_t* o = (uint32_t*)out + p0;
            if (full) *(uint4*)o = make_uint4(w[0], w[1], w[2], w[3]);
            else
                for (uint32_t k = 0; p0 + k < npix; ++k) o[k] = w[k];
        } else {  // RGB8: 12 bytes per four pixels
            unsigned char* o = (unsigned char*)out + (size_t)p0 * 3u;
            if (full) {
                uint32_t b[12];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    b[3u * k] = to8(px[k].x);
                    b[3u * k + 1u] = to8(px[k].y);
                    b[3u * k + 2u] = to8(px[k].z);
                }
                uint32_t* o4 = (uint32_t*)o;
                o4[0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                o4[1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
                o4[2] = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
            } else {
                for (uint32_t k = 0; p0 + k < npix; ++k) {
                    o[3u * k] = (unsigned char)to8(px[k].x);
                    o[3u * k + 1u] = (unsigned char)to8(px[k].y);
                    o[3u * k + 2u] = (unsigned char)to8(px[k].z);
                }
            }
        }
    }
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

void launch_iterate(const IterArgs& a, uint32_t block, bool xcd_local, int mode, hipStream_t s) {
    const uint32_t grid = (a.n_jobs + block - 1) / block;
    if (xcd_local) {
        if (mode == 2) hipLaunchKernelGGL((k_iterate<true, 2>), dim3(grid), dim3(block), 0, s, a);
        else if (mode == 1) hipLaunchKernelGGL((k_iterate<true, 1>), dim3(grid), dim3(block), 0, s, a);
        else hipLaunchKernelGGL((k_iterate<true, 0>), dim3(grid), dim3(block), 0, s, a);
    } else {
        if (mode == 2) hipLaunchKernelGGL((k_iterate<false, 2>), dim3(grid), dim3(block), 0, s, a);
        else if (mode == 1) hipLaunchKernelGGL((k_iterate<false, 1>), dim3(grid), dim3(block), 0, s, a);
        else hipLaunchKernelGGL((k_iterate<false, 0>), dim3(grid), dim3(block), 0, s, a);
    }
}

uint32_t lean_wave_lds_bytes(uint32_t bins, uint32_t records) { return kLeanWaveLds(bins, records); }
uint32_t chunk_bytes(uint32_t records) { return kChunkStride(records) * 16u; }

// the instantiations of the hot kernel: chunk size x depth-pipeline length x hint type (count-only kernels have neither)
#define SAR_FOR_EACH_LEAN(X)                                                                                          \
    X(true, 12u, 1u, unsigned short) X(true, 12u, 2u, unsigned short) X(true, 20u, 1u, unsigned short)                \
    X(true, 20u, 2u, unsigned short) X(true, 28u, 1u, unsigned short) X(true, 28u, 2u, unsigned short)                \
    X(true, 12u, 1u, uint32_t) X(true, 12u, 2u, uint32_t) X(true, 20u, 1u, uint32_t) X(true, 20u, 2u, uint32_t)         \
    X(true, 28u, 1u, uint32_t) X(true, 28u, 2u, uint32_t)                                                              \
    X(false, 12u, 1u, unsigned short) X(false, 20u, 1u, unsigned short) X(false, 28u, 1u, unsigned short)

int launch_iterate_lean(const BinIterArgs& a, uint32_t block, uint32_t records, uint32_t pipe, uint32_t hint_bytes, bool depth,
                        hipStream_t s) {
    const uint32_t grid = (a.it.n_jobs + block - 1) / block;
    const size_t lds = (size_t)(block / 64u) * kLeanWaveLds(a.n_bins, records);
    if (!depth) {
        pipe = 1;
        hint_bytes = 2;
    }
    bool launched = false;
#define SAR_LAUNCH_LEAN(DD, RR, UU, HH)                                                                    \
    if (!launched && depth == DD && records == RR && pipe == UU && hint_bytes == sizeof(HH)) {             \
        hipLaunchKernelGGL((k_iterate_lean<DD, RR, UU, HH>), dim3(grid), dim3(block), lds, s, a);          \
        launched = true;                                                                                   \
    }
    SAR_FOR_EACH_LEAN(SAR_LAUNCH_LEAN)
#undef SAR_LAUNCH_LEAN
    return launched ? 0 : 1;
}

int launch_bin_accumulate(const BinAccArgs& a, uint32_t threads, uint32_t records, hipStream_t s) {
    const size_t lds = (size_t)4u << a.bin_shift;
    // a list takes a group of 2 or 4 lanes: 1024 threads walk 256..512 lists per block
    if (threads == 0) threads = 1024u;
    switch (records) {
        case 12: hipLaunchKernelGGL(k_bin_accumulate<12u>, dim3(a.n_bins, a.splits), dim3(threads), lds, s, a); break;
        case 20: hipLaunchKernelGGL(k_bin_accumulate<20u>, dim3(a.n_bins, a.splits), dim3(threads), lds, s, a); break;
        case 28: hipLaunchKernelGGL(k_bin_accumulate<28u>, dim3(a.n_bins, a.splits), dim3(threads), lds, s, a); break;
        default: return 1;
    }
    return 0;
}

int binned_kernel_attributes() {
    // both kernels need more dynamic LDS than the 64 KiB default window
    hipError_t e = hipSuccess;
#define SAR_ATTR_LEAN(DD, RR, UU, HH) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_iterate_lean<DD, RR, UU, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SAR_FOR_EACH_LEAN(SAR_ATTR_LEAN)
#undef SAR_ATTR_LEAN
#define SAR_ATTR(RR) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bin_accumulate<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024)
    SAR_ATTR(12u);
    SAR_ATTR(20u);
    SAR_ATTR(28u);
#undef SAR_ATTR
    return (int)e;
}

void launch_fold_resolve(const FoldArgs& a, hipStream_t s) {
    const uint32_t grid = (a.npix + FOLD_PIX - 1) / FOLD_PIX;
    hipLaunchKernelGGL(k_fold_resolve, dim3(grid), dim3(256), 0, s, a);
}

void launch_reset(uint32_t* count, unsigned long long* key, double* steps, uint32_t npix, uint32_t* scalars,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_reset, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, count, key, steps, npix, scalars);
}
void launch_zbuf_out(const unsigned long long* key, float* out, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_zbuf_out, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, key, out, npix);
}
void launch_zbuf_in(const float* z, unsigned long long* key, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_zbuf_in, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, z, key, npix);
}
void launch_merge(uint32_t* count, unsigned long long* key, double* steps, const uint32_t* ocount,
                  const unsigned long long* okey, const double* osteps, uint32_t npix, uint32_t* scalars,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_merge, dim3(grid_for(npix, 256, 2048)), dim3(256), 0, s, count, key, steps, ocount, okey,
                       osteps, npix, scalars);
}
void launch_colorize_gas(const uint32_t* count, const double* steps, const uint32_t* scalars, const double* lut,
                         uint32_t lut_len, const PaletteParams& pal, double b_offset, double b_factor,
                         int transparent, uint32_t npix, void* out, hipStream_t s) {
    hipLaunchKernelGGL(k_colorize_gas, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, s, count, steps, scalars, lut,
                       lut_len, pal, b_offset, b_factor, transparent, npix, (ushort4*)out);
}
void launch_colorize_depth(const unsigned long long* key, uint32_t* scalars, uint32_t npix, void* out,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_zrange_init, dim3(1), dim3(1), 0, s, scalars);
    hipLaunchKernelGGL(k_zrange, dim3(grid_for(npix, 256, 1024)), dim3(256), 0, s, key, npix, scalars);
    hipLaunchKernelGGL(k_colorize_depth, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, s, key, scalars, npix,
                       (ushort4*)out);
}
int launch_convert(const void* rgba16, int format, void* out, uint32_t npix, hipStream_t s) {
    const dim3 grid(grid_for((npix + 3u) / 4u, 256, 8192)), block(256);
    switch (format) {
        case SAR_FMT_RGB16: hipLaunchKernelGGL(k_convert<SAR_FMT_RGB16>, grid, block, 0, s, (const ushort4*)rgba16, out, npix); break;
        case SAR_FMT_RGBA8: hipLaunchKernelGGL(k_convert<SAR_FMT_RGBA8>, grid, block, 0, s, (const ushort4*)rgba16, out, npix); break;
        case SAR_FMT_RGB8: hipLaunchKernelGGL(k_convert<SAR_FMT_RGB8>, grid, block, 0, s, (const ushort4*)rgba16, out, npix); break;
        default: return 1;
    }
    return 0;
}

uint32_t launch_extent(const MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* out, hipStream_t s) {
    const uint32_t blocks = (n_jobs + 255u) / 256u;
    hipLaunchKernelGGL(k_extent, dim3(blocks), dim3(256), 0, s, p, starts, n_jobs, iters, out);
    return blocks;
}

void launch_starts_soa(const double* aos, double* soa, uint32_t m, hipStream_t s) {
    hipLaunchKernelGGL(k_starts_soa, dim3((m + 255u) / 256u), dim3(256), 0, s, aos, soa, m);
}

void launch_warmup(const MapParams& p, const double* starts, uint32_t n_jobs, uint64_t iters, double* warm, uint32_t* joblist,
                   uint32_t* active, unsigned long long* nan_count, hipStream_t s) {
    hipLaunchKernelGGL(k_warmup, dim3((n_jobs + 255u) / 256u), dim3(256), 0, s, p, starts, n_jobs, iters, warm, joblist, active,
                       nan_count);
}

void launch_exch_export(const unsigned long long* key, uint32_t rank, void* out, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_export, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, key, rank, (long long*)out,
                       npix);
}
void launch_exch_select(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t rank,
                        const void* reduced, void* out, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_select, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, count, key, steps, rank,
                       (const long long*)reduced, (int*)out, npix);
}
void launch_exch_import(uint32_t* count, unsigned long long* key, double* steps, const void* reduced,
                        const void* sum, uint32_t npix, uint32_t* scalars, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_import, dim3(grid_for(npix, 256, 2048)), dim3(256), 0, s, count, key, steps,
                       (const long long*)reduced, (const int*)sum, npix, scalars);
}

}  // namespace sar
