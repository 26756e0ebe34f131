/*
 * sar_oracle.c — CPU oracle (plain C restatement of the reference arithmetic). TEST INFRASTRUCTURE.
 * See sar_oracle.h for the usage rules and the parity status ("bit-level parity unpinned";
 * statistically pinned against the reference's PNG; KATs frozen in tests/golden/).
 *
 * Build: gcc -O2 -ffp-contract=off (MANDATORY: Rust/LLVM never fuses a*b+c; a single fused op
 * changes a chaotic trajectory completely). No -ffast-math.
 *
 * Every function cites the reference lines (Icelk/strange-attractor-renderer, src/lib.rs) it follows.
 */
#include "sar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Rust `as` casts (saturating, NaN -> 0) ------------------------------------------------ */
static inline uint32_t as_u32(double v) {
    if (!(v == v)) return 0u;
    if (v <= 0.0) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
static inline uint16_t as_u16(double v) {
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 65535.0) return 65535;
    return (uint16_t)v;
}
static inline uint16_t as_u16_f32(float v) {
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 65535.0f) return 65535;
    return (uint16_t)v;
}

/* ---- a1: next_point, src/lib.rs:583-621 ------------------------------------------------------ */
/* sum_coefficients (:588-600): sum starts at 0., adds m[i]*c[i] strictly left to right. */
static inline double sum10(const double m[10], const double c[10]) {
    double sum = 0.;
    for (int i = 0; i < 10; ++i) {
        double prod = m[i] * c[i];
        sum = sum + prod;
    }
    return sum;
}

void sar_oracle_next_point(const sar_config* cfg, const double p[3], double out[3]) {
    const double x = p[0], y = p[1], z = p[2];
    /* monomials in the reference's order (:602-613) */
    double m[10];
    m[0] = 1.;
    m[1] = x;
    m[2] = x * x;
    m[3] = x * y;
    m[4] = x * z;
    m[5] = y;
    m[6] = y * y;
    m[7] = y * z;
    m[8] = z;
    m[9] = z * z;
    out[0] = sum10(m, cfg->coeff_x);
    out[1] = sum10(m, cfg->coeff_y);
    out[2] = sum10(m, cfg->coeff_z);
}

void sar_oracle_iterate(const sar_config* cfg, const double p0[3], uint64_t n, double out[3]) {
    double p[3] = {p0[0], p0[1], p0[2]};
    for (uint64_t k = 0; k < n; ++k) {
        double q[3];
        sar_oracle_next_point(cfg, p, q);
        p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
    }
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
}

/* ---- a3: rotation matrix, src/lib.rs:176-196 (release build: axis used as given) ------------- */
void sar_oracle_rotation_matrix(const sar_config* cfg, double m[9]) {
    const double x = cfg->rotation_axis[0], y = cfg->rotation_axis[1], z = cfg->rotation_axis[2];
    const double c = cos(cfg->rotation_angle);
    const double c1 = 1. - c;
    const double s = sin(cfg->rotation_angle);
    /* entries exactly as :190-192; `x * y * c1` parses as (x*y)*c1 */
    m[0] = c + x * x * c1;      m[1] = x * y * c1 - z * s;  m[2] = x * z * c1 + y * s;
    m[3] = y * x * c1 + z * s;  m[4] = c + y * y * c1;      m[5] = y * z * c1 - x * s;
    m[6] = z * x * c1 - y * s;  m[7] = z * y * c1 + x * s;  m[8] = c + z * z * c1;
}

/* Matrix3x3::mul_right, src/lib.rs:205-216: (m0*x + m1*y) + m2*z per row */
static inline void mul_right(const double m[9], const double v[3], double out[3]) {
    out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    out[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    out[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}

/* Vec3::magnitude, src/lib.rs:129-131 */
static inline double magnitude(const double v[3]) {
    return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
}

/* Extent of the attractor (the reference's TODO first pass, src/lib.rs:326-333). */
void sar_oracle_extent(const sar_config* cfg, const double* starts_xyz, uint32_t jobs, uint64_t iters_per_job,
                       double out[12]) {
    double m[9];
    sar_oracle_rotation_matrix(cfg, m);
    for (int k = 0; k < 6; ++k) {
        out[2 * k] = INFINITY;
        out[2 * k + 1] = -INFINITY;
    }
    for (uint32_t job = 0; job < jobs; ++job) {
        double p[3] = {starts_xyz[3 * job], starts_xyz[3 * job + 1], starts_xyz[3 * job + 2]}, q[3], ss[3];
        for (int w = 0; w < 1000; ++w) {  /* :750-752 */
            sar_oracle_next_point(cfg, p, q);
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
        }
        for (uint64_t t = 0; t < iters_per_job; ++t) {
            sar_oracle_next_point(cfg, p, q);
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            mul_right(m, p, ss);  /* :773 */
            for (int c = 0; c < 3; ++c) {
                if (ss[c] < out[2 * c]) out[2 * c] = ss[c];
                if (ss[c] > out[2 * c + 1]) out[2 * c + 1] = ss[c];
                if (p[c] < out[6 + 2 * c]) out[6 + 2 * c] = p[c];
                if (p[c] > out[6 + 2 * c + 1]) out[6 + 2 * c + 1] = p[c];
            }
        }
    }
}

/* ---- a8 / a8': colour transforms, src/lib.rs:507-516, 520-558 ------------------------------- */
double sar_oracle_color_transform(const sar_config* cfg, const double delta[3], const double ss[3]) {
    if (cfg->color_transform == SAR_CT_ADJUSTED_VELOCITY) {
        /* (delta.magnitude() + offset) * factor, :514 */
        return (magnitude(delta) + cfg->ct_offset) * cfg->ct_factor;
    }
    /* poisson_saturne: literal cos/sin of 45.5 degrees (:529-536), independent of config.angle */
    const double COS = 0.7009092642998509;  /* == the 64-digit literal at :530 (same double) */
    const double SIN = 0.7132504491541816;  /* == the literal at :536 */
    const double x2 = (ss[0] + cfg->center_camera[0]) * COS + (ss[2] + cfg->center_camera[1]) * SIN;
    double part;
    if (x2 < -0.0839
        || 10.55 * x2 + ss[1] < 0.46 - 1.0941
        || 1.0426 * x2 + ss[1] < 0.179 - 0.1576
        || 0.5139 * x2 - ss[1] > -0.04 - 0.04092) {
        part = 0.;
    } else {
        part = 1.;
    }
    const double color = (part + magnitude(delta)) / 2.;  /* :556 */
    return (color - 0.1) / 0.9;                            /* :557 */
}

/* ---- a9 / a10: Runtime, src/lib.rs:631-739 ---------------------------------------------------- */
sar_oracle_runtime* sar_oracle_runtime_new(uint32_t width, uint32_t height) {
    sar_oracle_runtime* rt = (sar_oracle_runtime*)calloc(1, sizeof(*rt));
    if (!rt) return NULL;
    const size_t n = (size_t)width * height;
    rt->width = width;
    rt->height = height;
    rt->count = (uint32_t*)malloc(n * sizeof(uint32_t) + 8);
    rt->steps = (double*)malloc(n * sizeof(double) + 8);
    rt->zbuf = (float*)malloc(n * sizeof(float) + 8);
    if (!rt->count || !rt->steps || !rt->zbuf) {
        sar_oracle_runtime_free(rt);
        return NULL;
    }
    sar_oracle_runtime_reset(rt);
    return rt;
}

void sar_oracle_runtime_free(sar_oracle_runtime* rt) {
    if (!rt) return;
    free(rt->count);
    free(rt->steps);
    free(rt->zbuf);
    free(rt);
}

/* reset, :682-699 */
void sar_oracle_runtime_reset(sar_oracle_runtime* rt) {
    const size_t n = (size_t)rt->width * rt->height;
    for (size_t k = 0; k < n; ++k) {
        rt->count[k] = 0u;
        rt->steps[k] = 0.;
        rt->zbuf[k] = -1.f;
    }
    rt->max = 0u;
}

/* merge, :708-738 — x outer / y inner like the reference (result is order independent) */
int sar_oracle_runtime_merge(sar_oracle_runtime* dst, const sar_oracle_runtime* src) {
    if (dst->width != src->width || dst->height != src->height) return SAR_ERR_DIM_MISMATCH;
    const uint32_t w = dst->width, h = dst->height;
    for (uint32_t x = 0; x < w; ++x) {
        for (uint32_t y = 0; y < h; ++y) {
            const size_t k = (size_t)y * w + x;
            const uint32_t merged = dst->count[k] + src->count[k]; /* wrapping (release) */
            dst->count[k] = merged;
            if (merged > dst->max) dst->max = merged;
            if (src->zbuf[k] > dst->zbuf[k]) { /* strict: dst wins ties */
                dst->steps[k] = src->steps[k];
                dst->zbuf[k] = src->zbuf[k];
            }
        }
    }
    return SAR_OK;
}

/* ---- a4-a7: render, src/lib.rs:747-838 --------------------------------------------------------- */
void sar_oracle_render(const sar_config* cfg, sar_oracle_runtime* rt, const double p0[3],
                       uint64_t iterations) {
    double cur[3] = {p0[0], p0[1], p0[2]};
    double nxt[3];
    /* warm-up, :750-752 */
    for (int k = 0; k < 1000; ++k) {
        sar_oracle_next_point(cfg, cur, nxt);
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];
    }
    /* hoisted constants, :755-764 */
    double m[9];
    sar_oracle_rotation_matrix(cfg, m);
    const double sin_v = sin(cfg->angle);
    const double cos_v = cos(cfg->angle);
    const double ccx = cfg->center_camera[0], ccy = cfg->center_camera[1], ccz = cfg->center_camera[2];
    const double width = (double)cfg->width;
    const double height = (double)cfg->height;
    const double width_scaled = width * cfg->scale;
    const double scale_adjusted_mid = 0.5 / cfg->scale;
    const uint32_t W = rt->width;

    double prev[3] = {cur[0], cur[1], cur[2]};

    for (uint64_t it = 0; it < iterations; ++it) {
        sar_oracle_next_point(cfg, cur, nxt); /* :770 */
        cur[0] = nxt[0]; cur[1] = nxt[1]; cur[2] = nxt[2];

        double ss[3];
        mul_right(m, cur, ss); /* :773 */

        /* :776-779 — note center_camera.y pairs with screen_space.z */
        const double x2 = (ss[0] + ccx) * cos_v + (ss[2] + ccy) * sin_v;
        const double z2 = (ss[0] + ccx) * sin_v - (ss[2] + ccy) * cos_v;
        const double fi = (scale_adjusted_mid - x2) * width_scaled;   /* :783 */
        const double fj = height / 2. - (ss[1] + ccz) * width_scaled; /* :786 */

        /* bounds test, :789 — a NaN passes (all comparisons false) */
        if (fi >= width || fj >= height || fi < 0. || fj < 0.) {
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2]; /* :793 */
            continue;
        }
        const uint32_t i = as_u32(fi); /* :800 */
        const uint32_t j = as_u32(fj); /* :802 */
        const size_t idx = (size_t)j * W + i;

        const uint32_t c = rt->count[idx] + 1u; /* :811, wrapping */
        rt->count[idx] = c;
        if (c > rt->max) rt->max = c; /* :813-815 */

        const float zf = (float)z2;  /* `z2 as f32`, round to nearest */
        if (zf > rt->zbuf[idx]) {    /* :821, strict */
            const double delta[3] = {cur[0] - prev[0], cur[1] - prev[1], cur[2] - prev[2]};
            rt->steps[idx] = sar_oracle_color_transform(cfg, delta, ss); /* :826-830 */
            rt->zbuf[idx] = zf;                                          /* :832 */
        }
        prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2]; /* :836 */
    }
}

void sar_oracle_render_jobs(const sar_config* cfg, sar_oracle_runtime* rt, const double* starts_xyz,
                            uint32_t jobs, uint64_t iters_per_job) {
    for (uint32_t k = 0; k < jobs; ++k) {
        sar_oracle_render(cfg, rt, starts_xyz + 3 * (size_t)k, iters_per_job);
    }
}

/* ---- a12: Palette::interpolate, src/lib.rs:442-472 -------------------------------------------- */
void sar_oracle_palette(const sar_config* cfg, double value, double rgb[3]) {
    const uint32_t len = cfg->palette_len; /* list has len+1 entries, last duplicated (:416-418) */
    const double count_f64 = (double)len;
    if (value < 0.) value = 0.;
    else if (value >= 1.) value = 0.999999; /* 0.999_999 */
    value = value * count_f64;
    uint32_t n = as_u32(floor(value));
    const double t = fmod(value, 1.);
    const double t1 = 1.0 - t;
    if (n >= len) n = len - 1; /* unreachable for non-NaN input; keeps the oracle memory safe */
    const uint32_t n1 = (n + 1 < len) ? n + 1 : len - 1; /* entry `len` is a copy of entry len-1 */
    for (int ch = 0; ch < 3; ++ch) {
        const double c1 = cfg->palette_rgb[n][ch];
        const double c2 = cfg->palette_rgb[n1][ch];
        rgb[ch] = sqrt(c2 * t + c1 * t1); /* :468-470 */
    }
}

/* ---- a13 / a14: colorize, src/lib.rs:841-904 -------------------------------------------------- */
/* src/bin/main.rs:52-57; image 0.25 `FromPrimitive<u16> for u8`: ((c + 128) / 257), channels copied in order,
 * alpha dropped (not pre-multiplied) by to_rgb16 / to_rgb8. */
static uint8_t oracle_u16_to_u8(uint16_t c) { return (uint8_t)(((uint32_t)c + 128u) / 257u); }

void sar_oracle_convert(int format, uint64_t npix, const uint16_t* rgba16, void* out) {
    uint16_t* o16 = (uint16_t*)out;
    uint8_t* o8 = (uint8_t*)out;
    for (uint64_t p = 0; p < npix; ++p) {
        const uint16_t* px = rgba16 + 4 * p;
        switch (format) {
            case SAR_FMT_RGBA16:
                for (int c = 0; c < 4; ++c) o16[4 * p + c] = px[c];
                break;
            case SAR_FMT_RGB16:
                for (int c = 0; c < 3; ++c) o16[3 * p + c] = px[c];
                break;
            case SAR_FMT_RGBA8:
                for (int c = 0; c < 4; ++c) o8[4 * p + c] = oracle_u16_to_u8(px[c]);
                break;
            case SAR_FMT_RGB8:
                for (int c = 0; c < 3; ++c) o8[3 * p + c] = oracle_u16_to_u8(px[c]);
                break;
            default:
                break;
        }
    }
}

void sar_oracle_colorize(const sar_config* cfg, const sar_oracle_runtime* rt, uint16_t* rgba) {
    const size_t n = (size_t)rt->width * rt->height;
    const double u16_max = 65535.;
    if (cfg->render_kind == SAR_RENDER_GAS) {
        const double off = cfg->brightness_offset, fac = cfg->brightness_factor;
        const double ln_base = log((double)(uint32_t)(rt->max + 1u));
        for (size_t k = 0; k < n; ++k) {
            double rgb[3];
            sar_oracle_palette(cfg, rt->steps[k], rgb);
            /* f64::log(self, base) == self.ln() / base.ln(), :860 */
            const double factor = log((double)(uint32_t)(rt->count[k] + 1u)) / ln_base;
            rgba[4 * k + 0] = as_u16((rgb[0] * factor + off) * fac * u16_max);
            rgba[4 * k + 1] = as_u16((rgb[1] * factor + off) * fac * u16_max);
            rgba[4 * k + 2] = as_u16((rgb[2] * factor + off) * fac * u16_max);
            rgba[4 * k + 3] = cfg->transparent ? as_u16(factor * u16_max) : 65535;
        }
    } else {
        /* fold seeds (0.0, f32::MAX), :877-882 */
        float zmax = 0.0f, zmin = 3.40282346638528859811704183484516925e+38f;
        for (size_t k = 0; k < n; ++k) {
            const float p = rt->zbuf[k];
            if (p != -1.0f) {
                zmax = fmaxf(zmax, p);
                zmin = fminf(zmin, p);
            }
        }
        const float diff = zmax - zmin;
        for (size_t k = 0; k < n; ++k) {
            float z = rt->zbuf[k];
            if (z == -1.0f) z = 0.0f;
            else z = (z - zmin) / diff; /* f32 arithmetic, :893 */
            const uint16_t v = as_u16_f32(z * 65535.0f);
            rgba[4 * k + 0] = v;
            rgba[4 * k + 1] = v;
            rgba[4 * k + 2] = v;
            rgba[4 * k + 3] = 65535;
        }
    }
}

/* ---- start-point stream (this project's definition; the reference uses OS entropy, :656) ------ */
/* SplitMix64 (Vigna's splitmix64.c) seeds xoshiro256++ 1.0 (Blackman & Vigna, xoshiro256plusplus.c): the algorithm rand 0.9
 * documents for SmallRng::seed_from_u64 on 64-bit targets (Cargo.toml:16; unpinned, no lockfile: SURVEY 8c). Jobs are drawn
 * in BLOCKS of 4096: block b uses the generator after b applications of the published jump() (2^128 steps each), job k of
 * the stream takes draws 3i..3i+2 (x, y, z) of block k / 4096 with i = k % 4096. The raw entries below exist for the tests
 * that hold the three pieces to their published vectors (tests/test_oracle_kat.py). */
static inline uint64_t rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }

static uint64_t splitmix64_next(uint64_t* state) {
    uint64_t z = (*state += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static uint64_t xoshiro256pp_next(uint64_t s[4]) {
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static void xoshiro256_jump(uint64_t s[4]) {
    static const uint64_t JUMP[] = {0x180ec6d33cfd0abaULL, 0xd5a61266f0c9392cULL, 0xa9582618e03fc9aaULL, 0x39abdc4529b1661cULL};
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (unsigned i = 0; i < sizeof JUMP / sizeof *JUMP; i++)
        for (int b = 0; b < 64; b++) {
            if (JUMP[i] & (UINT64_C(1) << b)) {
                s0 ^= s[0];
                s1 ^= s[1];
                s2 ^= s[2];
                s3 ^= s[3];
            }
            (void)xoshiro256pp_next(s);
        }
    s[0] = s0;
    s[1] = s1;
    s[2] = s2;
    s[3] = s3;
}

void sar_oracle_splitmix64(uint64_t seed, uint32_t n, uint64_t* out) {
    for (uint32_t k = 0; k < n; ++k) out[k] = splitmix64_next(&seed);
}
void sar_oracle_xoshiro256pp(uint64_t state[4], uint32_t n, uint64_t* out) {
    for (uint32_t k = 0; k < n; ++k) out[k] = xoshiro256pp_next(state);
}
void sar_oracle_xoshiro256_jump(uint64_t state[4]) { xoshiro256_jump(state); }
double sar_oracle_unit_f64(uint64_t raw) { return (double)(raw >> 11) * 0x1.0p-53; } /* [0,1): rand's StandardUniform for f64 */

void sar_oracle_start_points(uint64_t seed, uint64_t first_job, uint32_t n_jobs, double* xyz) {
    uint64_t block_state[4], s[4] = {0, 0, 0, 0};
    uint64_t sm = seed;
    for (int k = 0; k < 4; ++k) block_state[k] = splitmix64_next(&sm);
    uint64_t block = 0;     /* the block block_state belongs to */
    uint64_t drawn = 4096;  /* jobs already drawn from s; 4096 = s is not set up */
    uint64_t s_block = ~0ULL;
    for (uint64_t job = first_job; job < first_job + n_jobs; ++job) {
        const uint64_t b = job / 4096u, i = job % 4096u;
        while (block < b) {
            xoshiro256_jump(block_state);
            ++block;
        }
        if (s_block != b || drawn > i) {
            for (int k = 0; k < 4; ++k) s[k] = block_state[k];
            s_block = b;
            drawn = 0;
        }
        for (; drawn < i; ++drawn) {
            (void)xoshiro256pp_next(s);
            (void)xoshiro256pp_next(s);
            (void)xoshiro256pp_next(s);
        }
        for (int c = 0; c < 3; ++c) /* `random::<Vec3>() * 0.1`, :748: x then y then z (:161-166) */
            xyz[3 * (job - first_job) + c] = sar_oracle_unit_f64(xoshiro256pp_next(s)) * 0.1;
        ++drawn;
    }
}

uint64_t sar_oracle_fnv1a64(const void* data, uint64_t nbytes) {
    const unsigned char* p = (const unsigned char*)data;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t k = 0; k < nbytes; ++k) {
        h ^= p[k];
        h *= 0x100000001b3ULL;
    }
    return h;
}
