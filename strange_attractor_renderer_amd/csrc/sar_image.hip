// sar_image.hip — gfx950 (MI355X) streaming kernels around the Runtime buffers: reset, zbuf in/out, merge, colorize
// (gas and depth), the export format conversion and the pack / select / import kernels of the multi-GPU exchange.
#include "sar_device.hpp"
#include "sar_launch.hpp"

namespace sar {

// ---------------------------------------------------------------------------------------------------
// state management
// ---------------------------------------------------------------------------------------------------
// (+ the depth hints a launch has written since they were last cleared: one launch instead of a kernel and a fill per frame of a sweep)
__global__ void k_reset(uint32_t* count, unsigned long long* key, double* steps, uint32_t npix,
                        uint32_t* scalars, uint32_t* hints, uint32_t hint_words, uint32_t hint_fill) {
    const unsigned long long init = ((unsigned long long)f32_sortable(-1.0f) << 32) | 0xFFFFFFFFull;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        count[p] = 0u;   // :687
        steps[p] = 0.;   // :690
        key[p] = init;   // zbuf = -1.0, :693
    }
    const uint4 fill = make_uint4(hint_fill, hint_fill, hint_fill, hint_fill);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < hint_words / 4u; q += gridDim.x * blockDim.x) ((uint4*)hints)[q] = fill;
    if (blockIdx.x == 0 && threadIdx.x < (hint_words & 3u)) hints[(hint_words & ~3u) + threadIdx.x] = hint_fill;
    if (blockIdx.x == 0 && threadIdx.x < SC_COUNT) scalars[threadIdx.x] = 0u;  // max = 0, :694
}

// F resets in one launch (the frames of a batch of a sweep: sixteen launches of k_reset one behind the other under another lane's
// iterate kernel cost a tenth of the sweep): blockIdx.y is the frame, the table comes by value
__global__ void k_reset_batch(const ResetBatch t, uint32_t npix) {
    const ResetBatch::Frame f = t.f[blockIdx.y];
    const unsigned long long init = ((unsigned long long)f32_sortable(-1.0f) << 32) | 0xFFFFFFFFull;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        f.count[p] = 0u;
        f.steps[p] = 0.;
        f.key[p] = init;
    }
    const uint4 fill = make_uint4(f.hint_fill, f.hint_fill, f.hint_fill, f.hint_fill);
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < f.hint_words / 4u; q += gridDim.x * blockDim.x) ((uint4*)f.hints)[q] = fill;
    if (blockIdx.x == 0 && threadIdx.x < (f.hint_words & 3u)) f.hints[(f.hint_words & ~3u) + threadIdx.x] = f.hint_fill;
    if (blockIdx.x == 0 && threadIdx.x < SC_COUNT) f.scalars[threadIdx.x] = 0u;
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

__device__ __forceinline__ void colorize_gas_body(const uint32_t* count, const double* steps, const uint32_t* scalars, const double* lut,
                                                  uint32_t lut_len, const PaletteParams& pal, double b_offset, double b_factor, int transparent,
                                                  uint32_t npix, ushort4* out, int plain_palette, double* s_pal) {
    for (uint32_t k = threadIdx.x; k < (pal.len + 1) * 3; k += blockDim.x) s_pal[k] = pal.rgb[k / 3][k % 3];
    __syncthreads();
    const uint32_t rmax = scalars[SC_WRAP] ? 0xFFFFFFFFu : scalars[SC_MAX];
    const double ln_base = ln_u32(rmax + 1u, lut, lut_len);  // ln(max + 1), :860
    const double count_f64 = (double)pal.len;
    // A pixel nobody visited — four fifths of a frame, and whole waves of them — has factor = ln(1) / ln(max + 1) = +0, -0 (the
    // wrapped max: ln(0) = -inf) or NaN (an empty frame: 0 / 0), and a colour r >= 0 that is finite whenever the palette is
    // (plain_palette: every entry finite, >= 0 and far from overflow — checked on the host) and steps is not NaN: r * factor is
    // then the factor itself, whatever r — the pixel needs no palette, no square root, no division. The same expressions on the
    // same values: the same bits.
    const double factor0 = ln_u32(1u, lut, lut_len) / ln_base;
    ushort4 o0;
    o0.x = o0.y = o0.z = as_u16((factor0 + b_offset) * b_factor * 65535.);
    o0.w = transparent ? as_u16(factor0 * 65535.) : (uint16_t)65535;
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
        // Palette::interpolate (:442-472)
        double v = steps[p];
        const uint32_t cnt = count[p];
        if (plain_palette && cnt == 0u && v == v) {
            out[p] = o0;
            continue;
        }
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
        const double factor = ln_u32(cnt + 1u, lut, lut_len) / ln_base;
        ushort4 o;
        o.x = as_u16((r * factor + b_offset) * b_factor * 65535.);
        o.y = as_u16((g * factor + b_offset) * b_factor * 65535.);
        o.z = as_u16((b * factor + b_offset) * b_factor * 65535.);
        o.w = transparent ? as_u16(factor * 65535.) : (uint16_t)65535;
        out[p] = o;
    }
}

__global__ void __launch_bounds__(256) k_colorize_gas(const uint32_t* count, const double* steps,
                                                      const uint32_t* scalars, const double* lut,
                                                      uint32_t lut_len, const PaletteParams pal,
                                                      double b_offset, double b_factor, int transparent,
                                                      uint32_t npix, ushort4* out, int plain_palette) {
    __shared__ double s_pal[(SAR_PALETTE_MAX + 1) * 3];
    colorize_gas_body(count, steps, scalars, lut, lut_len, pal, b_offset, b_factor, transparent, npix, out, plain_palette, s_pal);
}
// F frames of one palette in one launch (blockIdx.y: the frame)
__global__ void __launch_bounds__(256) k_colorize_gas_batch(const ColorizeBatch t, const double* lut, uint32_t lut_len, const PaletteParams pal,
                                                            double b_offset, double b_factor, int transparent, uint32_t npix, int plain_palette) {
    __shared__ double s_pal[(SAR_PALETTE_MAX + 1) * 3];
    const ColorizeBatch::Frame f = t.f[blockIdx.y];
    colorize_gas_body(f.count, f.steps, f.scalars, lut, lut_len, pal, b_offset, b_factor, transparent, npix, (ushort4*)f.out, plain_palette, s_pal);
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
// multi-GPU exchange, sliced form: every rank OWNS one slice of S consecutive pixels. All-to-all of the slices
// (16 B/px on the wire, each pair of GPUs over its own xGMI link), then the owner folds the G partial slices with
// Runtime::merge (:708-738) in rank order, colorizes its slice, and only RGBA16 (8 B/px) travels to the root.
// Block layout (pack output and merge input alike): [G blocks][count u32 x S | sortable(z) u32 x S | steps f64 x S].
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_exch_pack(const uint32_t* __restrict__ count, const unsigned long long* __restrict__ key,
                                                   const double* __restrict__ steps, uint32_t npix, uint32_t S, uint32_t G,
                                                   unsigned char* __restrict__ out) {
    const uint32_t total = S * G;  // host guarantees S * G < 2^32
    const uint32_t unset = f32_sortable(-1.0f);
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const uint32_t d = p / S, o = p - d * S;
        unsigned char* blk = out + (size_t)d * S * 16u;
        const bool in = p < npix;  // slices are consecutive pixel ranges: pixel index == p
        ((uint32_t*)blk)[o] = in ? count[p] : 0u;
        ((uint32_t*)(blk + (size_t)S * 4u))[o] = in ? (uint32_t)(key[p] >> 32) : unset;
        ((double*)(blk + (size_t)S * 8u))[o] = in ? steps[p] : 0.;
    }
}

// seeds of the depth range (:877-882) and, on every rank but the accumulator of the fold (rank 0), max <- 0: the
// reference's merge never looks at other.max (:716-724), only at the merged counts it walks over
__global__ void k_exch_scalars_init(uint32_t* scalars, int keep_max) {
    scalars[SC_ZMAX] = f32_sortable(0.0f);
    scalars[SC_ZMIN] = f32_sortable(3.40282346638528859811704183484516925e+38f);
    if (!keep_max) {
        scalars[SC_MAX] = 0u;
        scalars[SC_WRAP] = 0u;
    }
}

__global__ void __launch_bounds__(256) k_exch_merge_slices(uint32_t* __restrict__ count, unsigned long long* __restrict__ key,
                                                           double* __restrict__ steps, uint32_t first, uint32_t n, uint32_t S,
                                                           uint32_t G, const unsigned char* __restrict__ in, uint32_t* scalars) {
    const uint32_t unset = f32_sortable(-1.0f);
    uint32_t local_max = 0;
    uint32_t zmx = f32_sortable(0.0f), zmn = f32_sortable(3.40282346638528859811704183484516925e+38f);
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
        // the accumulator is rank 0's partial (`current`, :1070); ranks 1.. are merged into it in order (:1072-1076)
        uint32_t c = ((const uint32_t*)in)[o];
        uint32_t z = ((const uint32_t*)(in + (size_t)S * 4u))[o];
        double st = ((const double*)(in + (size_t)S * 8u))[o];
        for (uint32_t r = 1; r < G; ++r) {
            const unsigned char* blk = in + (size_t)r * S * 16u;
            c += ((const uint32_t*)blk)[o];                      // wrapping, :719
            local_max = c > local_max ? c : local_max;           // the running max sees every intermediate sum, :721-723
            const uint32_t oz = ((const uint32_t*)(blk + (size_t)S * 4u))[o];
            if (oz > z) {                                        // strict: the earlier rank wins ties, :728
                z = oz;
                st = ((const double*)(blk + (size_t)S * 8u))[o];
            }
        }
        const uint32_t px = first + o;
        count[px] = c;
        key[px] = ((unsigned long long)z << 32) | 0xFFFFFFFFull;
        steps[px] = st;
        if (z != unset) {
            zmx = z > zmx ? z : zmx;
            zmn = z < zmn ? z : zmn;
        }
    }
    __shared__ uint32_t s_tmp[4];
    const uint32_t m = block_max_u32(local_max, s_tmp);
    zmx = block_max_u32(zmx, s_tmp);
    zmn = block_min_u32(zmn, s_tmp);
    if (threadIdx.x == 0) {
        if (m) raise_scalar(&scalars[SC_MAX], m);
        atomicMax(&scalars[SC_ZMAX], zmx);
        atomicMin(&scalars[SC_ZMIN], zmn);
    }
}

// ---- sparse form: only the 64-pixel granules that differ from the reset state travel (a fifth of a frame) -----------------
// One WAVE per granule, lane = pixel. A granule is touched when any of its pixels has a count or a depth.
__device__ __forceinline__ bool exch_granule_touched(const uint32_t* __restrict__ count, const unsigned long long* __restrict__ key,
                                                     uint32_t npix, uint32_t p, uint32_t& c, uint32_t& z) {
    const uint32_t unset = f32_sortable(-1.0f);
    const bool in = p < npix;
    c = in ? count[p] : 0u;
    z = in ? (uint32_t)(key[p] >> 32) : unset;
    return wave_ballot(c != 0u || z != unset) != 0ull;
}
__device__ __forceinline__ void exch_write_record(unsigned char* rec, uint32_t lane, uint32_t c, uint32_t z, double st) {
    ((uint32_t*)rec)[lane] = c;
    ((uint32_t*)(rec + (size_t)kExchSeg * 4u))[lane] = z;
    ((double*)(rec + (size_t)kExchSeg * 8u))[lane] = st;
}
__global__ void __launch_bounds__(256) k_exch_flags(const uint32_t* __restrict__ count, const unsigned long long* __restrict__ key, uint32_t npix,
                                                    uint32_t nseg, unsigned char* __restrict__ flags) {
    const uint32_t seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= nseg) return;
    uint32_t c, z;
    const bool touched = exch_granule_touched(count, key, npix, seg * kExchSeg + lane, c, z);
    if (lane == 0u) flags[seg] = touched ? 1 : 0;
}
// one process per GPU: THE PLAN of a sparse exchange, from every rank's granule flags (flags_all[world][nseg], all-gathered) —
//   send_slot[g]      where my record of granule g goes in the send buffer: owner by owner (an owner's granules are consecutive),
//                     granule by granule; -1 = not sent
//   recv_slot[r][s]   where rank r's record of granule s of MY slice arrives, source by source; -1 = none
//   counts[0..world) records I send to each owner, [world..2 world) records I receive from each source (the split sizes of the
//   all-to-all: the only numbers that go to the host), [2 world] granules touched on all ranks together (dense or sparse)
// 32 workgroups scan my row, 32 the column of my slice, the others count: every rank runs the same arithmetic on the same flags.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t* s_wave /* [17] */) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); ++w) {
            const uint32_t t = s_wave[w];
            s_wave[w] = run;
            run += t;
        }
    }
    __syncthreads();
    return s_wave[wave] + inc - v;
}

// A scan is cut over `parts` workgroups: part p owns a run of consecutive 64-element groups. It first COUNTS the flags before its
// run (all its waves stride over them: at most n bytes out of the L2 — cheaper than a second launch or a look-back chain), then
// every wave takes a sub-run, one element per lane (coalesced): the flags of a group are one ballot, an element's slot the count
// before its group plus the flagged lanes below it. Slices are whole 2048-pixel blocks, so an owner's / a source's range of
// granules starts on a multiple of 32: each HALF of a group belongs to one of them.
template <typename FlagAt>
__device__ __forceinline__ void exch_plan_scan(uint32_t n, uint32_t sps, FlagAt flag, int32_t* __restrict__ slot, uint32_t* __restrict__ group_counts,
                                               uint32_t n_groups_out, uint32_t part, uint32_t parts, uint32_t* s_wave, uint32_t* s_cnt) {
    const uint32_t waves = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t groups = (n + 63u) / 64u, per_part = (groups + parts - 1u) / parts;
    const uint32_t p0 = part * per_part < groups ? part * per_part : groups, p1 = p0 + per_part < groups ? p0 + per_part : groups;
    const uint32_t per = (p1 - p0 + waves - 1u) / waves;
    const uint32_t g0 = p0 + wave * per < p1 ? p0 + wave * per : p1, g1 = g0 + per < p1 ? g0 + per : p1;
    for (uint32_t k = threadIdx.x; k < n_groups_out; k += blockDim.x) s_cnt[k] = 0u;
    uint32_t before = 0, c = 0;
    for (uint32_t g = wave; g < p0; g += waves) before += (uint32_t)__popcll(wave_ballot(g * 64u + lane < n && flag(g * 64u + lane)));
    for (uint32_t g = g0; g < g1; ++g) {
        const uint32_t i = g * 64u + lane;
        c += (uint32_t)__popcll(wave_ballot(i < n && flag(i)));
    }
    if (lane == 0u) { s_wave[wave] = c; s_wave[17u + wave] = before; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < waves; ++w) run += s_wave[17u + w];   // everything before this part
        for (uint32_t w = 0; w < waves; ++w) {
            const uint32_t t = s_wave[w];
            s_wave[w] = run;
            run += t;
        }
    }
    __syncthreads();
    uint32_t at = s_wave[wave];
    for (uint32_t g = g0; g < g1; ++g) {
        const uint32_t i = g * 64u + lane;
        const bool f = i < n && flag(i);
        const unsigned long long mask = wave_ballot(f);
        if (i < n) slot[i] = f ? (int32_t)(at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))) : -1;
        at += (uint32_t)__popcll(mask);
        if (lane == 0u) {
            const uint32_t lo = (uint32_t)__popcll(mask & 0xFFFFFFFFull), hi = (uint32_t)__popcll(mask >> 32);
            if (lo) atomicAdd(&s_cnt[(g * 64u) / sps], lo);
            if (hi) atomicAdd(&s_cnt[(g * 64u + 32u) / sps], hi);
        }
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_groups_out; k += blockDim.x)
        if (s_cnt[k]) atomicAdd(&group_counts[k], s_cnt[k]);
}

constexpr uint32_t kExchPlanParts = 32;  // workgroups per scan
__global__ void __launch_bounds__(1024) k_exch_plan(const unsigned char* __restrict__ flags_all, uint32_t world, uint32_t rank, uint32_t nseg, uint32_t sps,
                                                    int32_t* __restrict__ send_slot, int32_t* __restrict__ recv_slot, uint32_t* __restrict__ counts) {
    __shared__ uint32_t s_wave[34];
    __shared__ uint32_t s_cnt[kMaxExchRanks];
    if (blockIdx.x < kExchPlanParts) {
        const unsigned char* mine = flags_all + (size_t)rank * nseg;
        exch_plan_scan(nseg, sps, [&](uint32_t g) { return mine[g] != 0; }, send_slot, counts, world, blockIdx.x, kExchPlanParts, s_wave, s_cnt);
    } else if (blockIdx.x < 2u * kExchPlanParts) {
        const uint32_t g0 = rank * sps;
        exch_plan_scan(world * sps, sps, [&](uint32_t e) {
            const uint32_t r = e / sps, g = g0 + (e - r * sps);
            return g < nseg && flags_all[(size_t)r * nseg + g] != 0;
        }, recv_slot, counts + world, world, blockIdx.x - kExchPlanParts, kExchPlanParts, s_wave, s_cnt);
    } else {
        const size_t total = (size_t)world * nseg;
        const uint32_t part = blockIdx.x - 2u * kExchPlanParts, parts = gridDim.x - 2u * kExchPlanParts;
        uint32_t c = 0;
        for (size_t i = (size_t)part * blockDim.x + threadIdx.x; i < total; i += (size_t)parts * blockDim.x) c += flags_all[i] ? 1u : 0u;
        c = block_exclusive_sum(c, s_wave) + c;   // (the last thread holds the block's sum)
        if (threadIdx.x == blockDim.x - 1u && c) atomicAdd(&counts[2u * world], c);
    }
}
// the records of this rank's touched granules, compacted in the order send_slot gives, for a variable-size all-to-all
__global__ void __launch_bounds__(256) k_exch_pack_sparse(const uint32_t* __restrict__ count, const unsigned long long* __restrict__ key,
                                                          const double* __restrict__ steps, uint32_t npix, uint32_t nseg,
                                                          const int32_t* __restrict__ send_slot, unsigned char* __restrict__ out) {
    const uint32_t seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= nseg) return;
    const int32_t slot = send_slot[seg];
    if (slot < 0) return;
    const uint32_t p = seg * kExchSeg + lane;
    const bool in = p < npix;
    exch_write_record(out + (size_t)slot * kExchRecordBytes, lane, in ? count[p] : 0u, in ? (uint32_t)(key[p] >> 32) : f32_sortable(-1.0f),
                      in ? steps[p] : 0.);
}
// the multi-device renderer: every wave looks at one granule of the image, finds its owner and — if the granule is touched —
// stores its record into the owner's buffer
__global__ void __launch_bounds__(256) k_exch_push(const ExchPushArgs a) {
    const uint32_t seg = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (seg >= a.G * a.sps) return;
    const uint32_t o = seg / a.sps, s = seg - o * a.sps;
    uint32_t c = 0u, z = 0u;
    const uint32_t p = seg * kExchSeg + lane;
    const bool flag = seg < a.nseg && exch_granule_touched(a.count, a.key, a.npix, p, c, z);
    const uint32_t rec = a.src * a.sps + s;
    if (lane == 0u) {
        a.slot[o][rec] = flag ? (int32_t)rec : -1;
        if (flag && o != a.src) atomicAdd(a.bytes, (unsigned long long)kExchRecordBytes);
    }
    if (!flag) return;
    exch_write_record(a.recv[o] + (size_t)rec * kExchRecordBytes, lane, c, z, p < a.npix ? a.steps[p] : 0.);
}
// The owner folds what arrived for its slice with Runtime::merge in rank order (:708-738, :1068-1076). A rank without a record
// for a granule holds the reset state there — count 0, zbuf -1.0: adding it changes no sum and it never wins a depth test — so
// skipping it is exactly the dense fold (k_exch_merge_slices); a granule nobody sent is already in the reset state here.
__global__ void __launch_bounds__(256) k_exch_merge_sparse(uint32_t* __restrict__ count, unsigned long long* __restrict__ key,
                                                           double* __restrict__ steps, uint32_t first, uint32_t n, uint32_t sps, uint32_t G,
                                                           const unsigned char* __restrict__ records, const int32_t* __restrict__ slot,
                                                           uint32_t* scalars) {
    const uint32_t unset = f32_sortable(-1.0f);
    uint32_t local_max = 0;
    uint32_t zmx = f32_sortable(0.0f), zmn = f32_sortable(3.40282346638528859811704183484516925e+38f);
    for (uint32_t o = blockIdx.x * blockDim.x + threadIdx.x; o < n; o += gridDim.x * blockDim.x) {
        const uint32_t s = o / kExchSeg, w = o - s * kExchSeg;
        uint32_t c = 0u, z = unset;
        double st = 0.;
        bool any = false;
        for (uint32_t r = 0; r < G; ++r) {
            const int32_t at = slot[r * sps + s];
            if (at >= 0) {
                const unsigned char* rec = records + (size_t)at * kExchRecordBytes;
                const uint32_t oc = ((const uint32_t*)rec)[w];
                const uint32_t oz = ((const uint32_t*)(rec + (size_t)kExchSeg * 4u))[w];
                any = true;
                c += oc;                                             // rank 0: the accumulator (`current`, :1070); then wrapping adds, :719
                if (r == 0u || oz > z) {                             // strict: the earlier rank wins ties, :728
                    z = oz;
                    st = ((const double*)(rec + (size_t)kExchSeg * 8u))[w];
                }
            }
            // the running max sees every intermediate sum of the fold (:721-723) — an absent rank's is the sum before it
            if (r != 0u) local_max = c > local_max ? c : local_max;
        }
        if (!any) continue;
        const uint32_t px = first + o;
        count[px] = c;
        key[px] = ((unsigned long long)z << 32) | 0xFFFFFFFFull;
        steps[px] = st;
        if (z != unset) {
            zmx = z > zmx ? z : zmx;
            zmn = z < zmn ? z : zmn;
        }
    }
    __shared__ uint32_t s_tmp[4];
    const uint32_t m = block_max_u32(local_max, s_tmp);
    zmx = block_max_u32(zmx, s_tmp);
    zmn = block_min_u32(zmn, s_tmp);
    if (threadIdx.x == 0) {
        if (m) raise_scalar(&scalars[SC_MAX], m);
        atomicMax(&scalars[SC_ZMAX], zmx);
        atomicMin(&scalars[SC_ZMIN], zmn);
    }
}

// {max, wrap flag, sortable(zmax), ~sortable(zmin)} as int64: one all-reduce MAX makes them global
__global__ void k_exch_scalars_export(const uint32_t* scalars, long long* out4) {
    out4[0] = scalars[SC_MAX];
    out4[1] = scalars[SC_WRAP];
    out4[2] = scalars[SC_ZMAX];
    out4[3] = (long long)(~scalars[SC_ZMIN]);
    __threadfence_system();  // out4 may be host memory that other devices read (the native renderer's board)
}
// the native multi-device renderer: every device's quad sits on a board in page-locked host memory (written by
// k_exch_scalars_export, one slot per device); each device reduces the G quads itself — no host round trip
__global__ void k_exch_scalars_reduce(uint32_t* scalars, const volatile long long* board, uint32_t G) {
    long long m[4] = {0, 0, 0, 0};
    for (uint32_t d = 0; d < G; ++d)
        for (int k = 0; k < 4; ++k) {
            const long long v = board[4 * d + k];
            m[k] = v > m[k] ? v : m[k];
        }
    scalars[SC_MAX] = (uint32_t)m[0];
    scalars[SC_WRAP] = (uint32_t)m[1];
    scalars[SC_ZMAX] = (uint32_t)m[2];
    scalars[SC_ZMIN] = ~(uint32_t)m[3];
}
__global__ void k_exch_scalars_import(uint32_t* scalars, const long long* in4) {
    scalars[SC_MAX] = (uint32_t)in4[0];
    scalars[SC_WRAP] = (uint32_t)in4[1];
    scalars[SC_ZMAX] = (uint32_t)in4[2];
    scalars[SC_ZMIN] = ~(uint32_t)in4[3];
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

void launch_reset(uint32_t* count, unsigned long long* key, double* steps, uint32_t npix, uint32_t* scalars, void* hints,
                  uint32_t hint_words, uint32_t hint_fill, hipStream_t s) {
    hipLaunchKernelGGL(k_reset, dim3(grid_for(npix, 256, 4096)), dim3(256), 0, s, count, key, steps, npix, scalars, (uint32_t*)hints,
                       hints ? hint_words : 0u, hint_fill);
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

static int plain_palette(const PaletteParams& pal) {
    int plain = 1;
    for (uint32_t k = 0; k <= pal.len && k <= SAR_PALETTE_MAX; ++k)
        for (int ch = 0; ch < 3; ++ch) {
            const double v = pal.rgb[k][ch];
            if (!(v >= 0.) || v > 1e300 || __builtin_signbit(v)) plain = 0;
        }
    return plain;
}
void launch_colorize_gas(const uint32_t* count, const double* steps, const uint32_t* scalars, const double* lut,
                         uint32_t lut_len, const PaletteParams& pal, double b_offset, double b_factor,
                         int transparent, uint32_t npix, void* out, hipStream_t s) {
    // (k_colorize_gas's short way for unvisited pixels needs colours that are finite whatever the blend: every entry a finite,
    // non-negative number — no -0.0, whose square root keeps its sign — far from overflow)
    const int plain = plain_palette(pal);
    hipLaunchKernelGGL(k_colorize_gas, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, s, count, steps, scalars, lut,
                       lut_len, pal, b_offset, b_factor, transparent, npix, (ushort4*)out, plain);
}

void launch_colorize_gas_batch(const ColorizeBatch& t, uint32_t n_frames, const double* lut, uint32_t lut_len, const PaletteParams& pal, double b_offset,
                               double b_factor, int transparent, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_colorize_gas_batch, dim3(grid_for(npix, 256, 2048), n_frames), dim3(256), 0, s, t, lut, lut_len, pal, b_offset, b_factor,
                       transparent, npix, plain_palette(pal));
}
void launch_reset_batch(const ResetBatch& t, uint32_t n_frames, uint32_t npix, hipStream_t s) {
    hipLaunchKernelGGL(k_reset_batch, dim3(grid_for(npix, 256, 1024), n_frames), dim3(256), 0, s, t, npix);
}

void launch_colorize_depth(const unsigned long long* key, uint32_t* scalars, uint32_t npix, void* out,
                           hipStream_t s) {
    hipLaunchKernelGGL(k_zrange_init, dim3(1), dim3(1), 0, s, scalars);
    hipLaunchKernelGGL(k_zrange, dim3(grid_for(npix, 256, 1024)), dim3(256), 0, s, key, npix, scalars);
    hipLaunchKernelGGL(k_colorize_depth, dim3(grid_for(npix, 256, 8192)), dim3(256), 0, s, key, scalars, npix,
                       (ushort4*)out);
}

// a pixel range with the depth range the scalars already hold (sliced multi-GPU colorize: the range is global)
void launch_colorize_depth_range(const unsigned long long* key, const uint32_t* scalars, uint32_t n, void* out, hipStream_t s) {
    hipLaunchKernelGGL(k_colorize_depth, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, key, scalars, n, (ushort4*)out);
}

void launch_exch_pack(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t npix, uint32_t S,
                      uint32_t G, void* out, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_pack, dim3(grid_for(S * G, 256, 8192)), dim3(256), 0, s, count, key, steps, npix, S, G,
                       (unsigned char*)out);
}

void launch_exch_merge_slices(uint32_t* count, unsigned long long* key, double* steps, uint32_t first, uint32_t n, uint32_t S,
                              uint32_t G, const void* in, uint32_t* scalars, bool keep_max, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_scalars_init, dim3(1), dim3(1), 0, s, scalars, keep_max ? 1 : 0);
    if (n)
        hipLaunchKernelGGL(k_exch_merge_slices, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, count, key, steps, first, n, S, G,
                           (const unsigned char*)in, scalars);
}

void launch_exch_flags(const uint32_t* count, const unsigned long long* key, uint32_t npix, void* flags, hipStream_t s) {
    const uint32_t nseg = (npix + kExchSeg - 1u) / kExchSeg;
    hipLaunchKernelGGL(k_exch_flags, dim3((nseg + 3u) / 4u), dim3(256), 0, s, count, key, npix, nseg, (unsigned char*)flags);
}
void launch_exch_plan(const void* flags_all, uint32_t world, uint32_t rank, uint32_t nseg, uint32_t sps, int32_t* send_slot, int32_t* recv_slot,
                      uint32_t* counts /* [2 * world + 1], zeroed */, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_plan, dim3(2u * kExchPlanParts + 32u), dim3(1024), 0, s, (const unsigned char*)flags_all, world, rank, nseg, sps, send_slot, recv_slot, counts);
}
void launch_exch_pack_sparse(const uint32_t* count, const unsigned long long* key, const double* steps, uint32_t npix, const int32_t* send_slot,
                             void* out, hipStream_t s) {
    const uint32_t nseg = (npix + kExchSeg - 1u) / kExchSeg;
    hipLaunchKernelGGL(k_exch_pack_sparse, dim3((nseg + 3u) / 4u), dim3(256), 0, s, count, key, steps, npix, nseg, send_slot, (unsigned char*)out);
}
void launch_exch_push(const ExchPushArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_push, dim3((a.G * a.sps + 3u) / 4u), dim3(256), 0, s, a);
}
void launch_exch_merge_sparse(uint32_t* count, unsigned long long* key, double* steps, uint32_t first, uint32_t n, uint32_t sps, uint32_t G,
                              const void* records, const int32_t* slot, uint32_t* scalars, bool keep_max, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_scalars_init, dim3(1), dim3(1), 0, s, scalars, keep_max ? 1 : 0);
    if (n)
        hipLaunchKernelGGL(k_exch_merge_sparse, dim3(grid_for(n, 256, 4096)), dim3(256), 0, s, count, key, steps, first, n, sps, G,
                           (const unsigned char*)records, slot, scalars);
}

void launch_exch_scalars_export(const uint32_t* scalars, void* out4, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_scalars_export, dim3(1), dim3(1), 0, s, scalars, (long long*)out4);
}
void launch_exch_scalars_reduce(uint32_t* scalars, const void* board, uint32_t G, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_scalars_reduce, dim3(1), dim3(1), 0, s, scalars, (const volatile long long*)board, G);
}
void launch_exch_scalars_import(uint32_t* scalars, const void* in4, hipStream_t s) {
    hipLaunchKernelGGL(k_exch_scalars_import, dim3(1), dim3(1), 0, s, scalars, (const long long*)in4);
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
