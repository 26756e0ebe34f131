// sar_device.hpp — device-side helpers shared by the kernel files (sar_iterate.hip, sar_accumulate.hip, sar_image.hip).
//
// Bit-exactness contract: every floating-point operation of the map, the projection and the colour transform is the
// reference's operation, in the reference's order, with separate multiply and add (all kernel files are compiled with
// -ffp-contract=off and the build checks that the iterate / warm-up / extent kernels contain no v_fma_f64). The chaotic
// map amplifies a 1-ulp deviation exponentially, so "close" does not exist here: either the op sequence is identical
// or the images differ.
#pragma once

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
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);  // negative: ~b, else b | sign — three instructions
}
__device__ __forceinline__ float sortable_f32(uint32_t s) {
    const uint32_t b = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
    return __uint_as_float(b);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// The narrow depth hints are 16-bit fixed point: q(z) = clamp(floor((z - z0) * s), 0, 65535). Every step is monotone
// non-decreasing in z, so q(a) < q(b) implies a < b: a visit whose q is below the stored q of an already-sent visit
// cannot win the depth test. Half the footprint of the 32-bit hints (the depth itself as f32), at the price of passing
// every visit within 1/s of the best depth: the host picks the type by the view's footprint (sar_runtime.cpp).
// (z0, s) spread the 65536 levels over the depths this view really produces — the range k_warmup saw during the last
// iterations of the warm-up, 12.5 % wider on either side; depths outside it clamp, which only lets more visits through.
// With no range known: z in [-1, 3) at 2^-14 (round 2's fixed quantiser; 4096^2: 5 % of the visits pass stage 1).
struct HintQuant {
    float z0, s;
};
__device__ __forceinline__ HintQuant hint_quant(const uint32_t* range) {
    HintQuant h = {-1.0f, 16384.0f};
    if (range) {  // {~sortable(min z), sortable(max z)}, both raised with atomicMax from 0
        const uint32_t nlo = range[0], hi = range[1];
        if (nlo && hi) {
            const float lo = sortable_f32(~nlo), hif = sortable_f32(hi);
            const float span = hif - lo;
            if (span > 0.0f) {
                h.z0 = lo - 0.125f * span;
                h.s = 65535.0f / (1.25f * span);
            }
        }
    }
    return h;
}
__device__ __forceinline__ uint32_t depth_q16(float zf, const HintQuant& h) {
    const float s = (zf - h.z0) * h.s;
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

__device__ __forceinline__ uint32_t xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4): id 20, offset 0, size 4 -> simm16 = (3<<11)|(0<<6)|20
    return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
}


// PoolStager (sar_iterate.hip): a staged chunk has its final form {prev, n, R x u16}; kPoolSpare spare buffers per wave
#ifndef SAR_POOL_SPARE
#define SAR_POOL_SPARE 16u  // a test build shrinks it (SAR_EXTRA_FLAGS=-DSAR_POOL_SPARE=2u) to force the many-fillers rounds
#endif
constexpr uint32_t kPoolChunkBytes(uint32_t R) { return 8u + 2u * R; }
constexpr uint32_t kChunkQuads(uint32_t R) { return (8u + 2u * R) / 16u; }  // 16-byte quads of data per chunk: R = 12, 20, 28, 60
// lanes that move one chunk (one quad each): 2, 4 or 8
constexpr uint32_t kChunkLanes(uint32_t R) { return kChunkQuads(R) <= 2u ? 2u : (kChunkQuads(R) <= 4u ? 4u : 8u); }
// spare buffers of the pool stager: as many as one cooperative copy-out of the wave moves (64 lanes), at most SAR_POOL_SPARE
constexpr uint32_t kPoolSpareOf(uint32_t R) { return SAR_POOL_SPARE < 64u / kChunkLanes(R) ? SAR_POOL_SPARE : 64u / kChunkLanes(R); }
constexpr uint32_t kPoolWaveLds(uint32_t bins, uint32_t R) {
    // buffers | ctl words | ring, rounded up to a multiple of 16 bytes
    return ((bins + kPoolSpareOf(R)) * kPoolChunkBytes(R) + bins * 4u + kPoolSpareOf(R) * 4u + 15u) & ~15u;
}
// Chunks never straddle a 64-byte sector of the arena: the 48-byte chunk (R = 20) is laid out on a 64-byte stride.
// The accumulate kernel's chunk reads are isolated, and an isolated read moves whole sectors (measured, tools/ubench/
// gather_runs.hip: 3.4 TB/s for 64-byte pieces) — a straddling chunk would cost two.
constexpr uint32_t kChunkStride(uint32_t R) { return R == 12u ? 2u : (R == 60u ? 8u : 4u); }


// a * b for operands below 2^24 as ONE full-rate instruction. (Through the __umul24 builtin the optimiser knows the
// operand ranges, turns the product back into a 32-bit multiply and picks quarter-rate v_mul_lo_u32 / v_mad_u64_u32.)
__device__ __forceinline__ uint32_t mul24(uint32_t a, uint32_t wave_uniform_b) {
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(wave_uniform_b), "v"(a));  // src0 may be scalar, src1 is a VGPR
    return r;
}

// (mask & a) | (~mask & b) as ONE instruction (the compiler splits it into v_and + v_and_or when ~mask is loop-invariant).
__device__ __forceinline__ uint32_t bfi(uint32_t wave_uniform_mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(wave_uniform_mask), "v"(a), "v"(b));
    return r;
}

// Lane mask of an unsigned compare, straight from the compare instruction. (wave_ballot of a bool that was combined
// from several conditions costs v_cndmask + v_cmp_ne on top of them.)
__device__ __forceinline__ unsigned long long lanes_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
__device__ __forceinline__ unsigned long long lanes_ge(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 35); }

// Lane mask of a predicate. (HIP's __ballot goes through an integer: v_cndmask + v_cmp_ne per call.)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }


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
    // Rust `as u32` is the hardware conversion: v_cvt_u32_f64 truncates, saturates and turns NaN into 0 (a C cast would
    // be undefined for NaN / out-of-range values, so the compiler is not asked)
    uint32_t i, j;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(i) : "v"(fi));
    asm("v_cvt_u32_f64 %0, %1" : "=v"(j) : "v"(fj));
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


// The argument block of one frame of a BATCHED launch (BatchFrame, sar_internal.hpp), read from the table in device memory
// through the CONSTANT address space: the address is wave-uniform and nothing writes the table while a launch runs, so these
// are scalar loads into SGPRs — what the by-value kernel arguments of the single-frame kernels are.
template <typename T>
__device__ __forceinline__ T load_frame_args(const T* p) {
    static_assert(sizeof(T) % 8 == 0 && alignof(T) == 8, "argument blocks are 8-byte aligned");
    typedef const __attribute__((address_space(4))) unsigned long long* Src;  // (dword-aligned or better: a scalar load needs it)
    union U { T v; unsigned long long w[sizeof(T) / 8]; __device__ U() {} } u;
    Src src = (Src)(uintptr_t)p;
#pragma unroll
    for (uint32_t k = 0; k < sizeof(T) / 8; ++k) u.w[k] = src[k];
    return u.v;
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
// launch wrappers (called from sar_runtime.cpp)
// ---------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint32_t n, uint32_t block, uint32_t cap) {
    uint32_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    return g ? g : 1;
}


}  // namespace sar
