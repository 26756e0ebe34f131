// sar_host.cpp — host-only part of the C ABI: Config presets, validation, setup math, the
// start-point stream and status strings. No device code here; see sar_runtime.hip for the path.
//
// Built with -ffp-contract=off: sar_rotation_matrix must produce the same doubles as the reference's
// EulerAxisRotation::to_rotation_matrix (src/lib.rs:176-196), which Rust/LLVM never contracts.
#include "sar_internal.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>

namespace sar {

thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

int abi_caught() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return SAR_ERR_OOM;
    } catch (const std::exception& e) {
        set_error("internal error: %s", e.what());
        return SAR_ERR_INVALID;
    } catch (...) {
        set_error("internal error (unknown exception)");
        return SAR_ERR_INVALID;
    }
}

// Config::new defaults, src/lib.rs:289-307; Colors::default :480-492; BrighnessConstants::default :397-404
static void config_defaults(sar_config* c) {
    std::memset(c, 0, sizeof(*c));
    c->iterations = 10000000ull;
    c->width = 1920;
    c->height = 1080;
    c->render_kind = SAR_RENDER_GAS;
    c->transparent = 1;
    c->angle = 0.0;
    c->silent = 1;
    c->attractor_kind = SAR_ATTRACTOR_SPROTT2;
    static const double pal[6][3] = {
        {1.0, 1.0, 0.5}, {0.5, 1.0, 0.5}, {1.0, 0.5, 0.5},
        {0.5, 1.0, 1.0}, {0.5, 0.5, 1.0}, {1.0, 0.5, 1.0},
    };
    c->palette_len = 6;
    for (int k = 0; k < 6; ++k)
        for (int ch = 0; ch < 3; ++ch) c->palette_rgb[k][ch] = pal[k][ch];
    c->brightness_offset = -0.15;
    c->brightness_factor = 5. / 3.;
    c->seed = 0;
    c->jobs_total = 1;
}

// xoshiro256++ seeded through SplitMix64 (definition in include/sar.h, sar_start_points): SplitMix64 as published by
// Steele, Lea & Flood / Vigna (splitmix64.c), xoshiro256++ 1.0 as published by Blackman & Vigna (xoshiro256plusplus.c) —
// what rand 0.9 documents for `SmallRng::seed_from_u64` on 64-bit targets. tests/test_oracle_kat.py holds both to their
// published vectors and proves the jump polynomial (T^(2^128) over GF(2)).
void Rng::seed(uint64_t seed) {
    uint64_t sm = seed;
    for (int k = 0; k < 4; ++k) {
        sm += 0x9e3779b97f4a7c15ull;
        uint64_t z = sm;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        s[k] = z ^ (z >> 31);
    }
    for (int k = 0; k < 4; ++k) base[k] = s[k];
    in_block = 0;
}

static inline uint64_t rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }

uint64_t Rng::next_u64() {
    const uint64_t r = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return r;
}

// xoshiro256's jump(): equivalent to 2^128 calls of next_u64 (the polynomial is the published JUMP constant)
void Rng::jump(uint64_t st[4]) {
    static const uint64_t kJump[4] = {0x180ec6d33cfd0abaull, 0xd5a61266f0c9392cull, 0xa9582618e03fc9aaull, 0x39abdc4529b1661cull};
    uint64_t acc[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w)
        for (int b = 0; b < 64; ++b) {
            if (kJump[w] & (1ull << b))
                for (int k = 0; k < 4; ++k) acc[k] ^= st[k];
            const uint64_t t = st[1] << 17;  // one step of the linear engine (next_u64 without its output)
            st[2] ^= st[0];
            st[3] ^= st[1];
            st[1] ^= st[2];
            st[0] ^= st[3];
            st[2] ^= t;
            st[3] = rotl64(st[3], 45);
        }
    for (int k = 0; k < 4; ++k) st[k] = acc[k];
}

// `rng.random::<Vec3>() * 0.1` (src/lib.rs:748, :161-166): x, y, z drawn in that order
void Rng::start_point(double out[3]) {
    if (in_block == kStartBlockJobs) {  // the next block: the generator 2^128 steps on from this block's start
        jump(base);
        for (int k = 0; k < 4; ++k) s[k] = base[k];
        in_block = 0;
    }
    for (int k = 0; k < 3; ++k) {
        const double u = static_cast<double>(next_u64() >> 11) * 0x1.0p-53;
        out[k] = u * 0.1;
    }
    ++in_block;
}

void Rng::skip_points(uint64_t n_jobs) {
    while (n_jobs) {
        if (in_block == kStartBlockJobs) {
            jump(base);
            for (int k = 0; k < 4; ++k) s[k] = base[k];
            in_block = 0;
        }
        const uint64_t room = kStartBlockJobs - in_block;
        if (n_jobs >= room) {  // the rest of this block: nothing to draw, the next block starts from jump(base)
            in_block = kStartBlockJobs;
            n_jobs -= room;
        } else {
            for (uint64_t d = 0; d < 3 * n_jobs; ++d) (void)next_u64();
            in_block += static_cast<uint32_t>(n_jobs);
            n_jobs = 0;
        }
    }
}

void rotation_matrix(const sar_config& cfg, double m[9]) {
    const double x = cfg.rotation_axis[0], y = cfg.rotation_axis[1], z = cfg.rotation_axis[2];
    const double c = std::cos(cfg.rotation_angle);
    const double c1 = 1. - c;
    const double s = std::sin(cfg.rotation_angle);
    m[0] = c + x * x * c1;     m[1] = x * y * c1 - z * s; m[2] = x * z * c1 + y * s;
    m[3] = y * x * c1 + z * s; m[4] = c + y * y * c1;     m[5] = y * z * c1 - x * s;
    m[6] = z * x * c1 - y * s; m[7] = z * y * c1 + x * s; m[8] = c + z * z * c1;
}

int validate(const sar_config* cfg) {
    if (!cfg) { set_error("config is NULL"); return SAR_ERR_INVALID; }
    if (cfg->width == 0 || cfg->height == 0) { set_error("zero image dimension"); return SAR_ERR_INVALID; }
    if (static_cast<uint64_t>(cfg->width) * cfg->height > 0x7fffffffull) {
        set_error("width*height exceeds 2^31-1 pixels"); return SAR_ERR_RANGE;
    }
    if (cfg->width >= (1u << 24) || cfg->height >= (1u << 24)) {  // pixel indices are formed with 24-bit multiplies
        set_error("an image dimension exceeds 2^24-1"); return SAR_ERR_RANGE;
    }
    if (cfg->render_kind != SAR_RENDER_GAS && cfg->render_kind != SAR_RENDER_DEPTH) {
        set_error("unknown render_kind %d", cfg->render_kind); return SAR_ERR_INVALID;
    }
    if (cfg->attractor_kind != SAR_ATTRACTOR_SPROTT2) {
        set_error("unknown attractor_kind %d", cfg->attractor_kind); return SAR_ERR_INVALID;
    }
    if (cfg->color_transform != SAR_CT_POISSON_SATURNE && cfg->color_transform != SAR_CT_ADJUSTED_VELOCITY) {
        set_error("unknown color_transform %d", cfg->color_transform); return SAR_ERR_INVALID;
    }
    if (cfg->palette_len == 0 || cfg->palette_len > SAR_PALETTE_MAX) { // Palette::new panics on empty, :413-418
        set_error("palette_len %u outside 1..%d", cfg->palette_len, SAR_PALETTE_MAX); return SAR_ERR_INVALID;
    }
    return SAR_OK;
}

}  // namespace sar

extern "C" {

int sar_abi_version(void) { return SAR_ABI_VERSION; }

#ifndef SAR_BUILD_ID
#define SAR_BUILD_ID "unidentified"
#endif
// (the marker lets build.py read the id of a library file without loading it)
const char* sar_build_id(void) {
    static const char marked[] = "SAR_BUILD_ID=" SAR_BUILD_ID;
    return marked + sizeof("SAR_BUILD_ID=") - 1;
}

int sar_checksum_fnv1a64(const void* data_host, size_t nbytes, uint64_t* out) try {
    if ((!data_host && nbytes) || !out) return SAR_ERR_INVALID;
    const unsigned char* p = static_cast<const unsigned char*>(data_host);
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t k = 0; k < nbytes; ++k) {
        h ^= p[k];
        h *= 0x100000001b3ULL;
    }
    *out = h;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

const char* sar_status_string(int status) {
    switch (status) {
        case SAR_OK: return "ok";
        case SAR_ERR_INVALID: return "invalid argument";
        case SAR_ERR_DIM_MISMATCH: return "runtime dimensions differ";
        case SAR_ERR_NO_DEVICE: return "no HIP device";
        case SAR_ERR_HIP: return "HIP call failed";
        case SAR_ERR_OOM: return "out of memory";
        case SAR_ERR_RANGE: return "size out of range";
        case SAR_ERR_IO: return "image file could not be written";
        default: return "unknown status";
    }
}

const char* sar_last_error(void) { return sar::g_last_error; }

int sar_config_poisson_saturne(sar_config* out) try {
    if (!out) return SAR_ERR_INVALID;
    sar::config_defaults(out);
    // values: src/lib.rs:311-350
    static const double x[10] = {0.021, 1.182, -1.183, 0.128, -1.12, -0.641, -1.152, -0.834, -0.97, 0.722};
    static const double y[10] = {0.243038, -0.825, -1.2, -0.835443, -0.835443, -0.364557, 0.458, 0.622785,
                                 -0.394937, -1.032911};
    static const double z[10] = {-0.455696, 0.673, 0.915, -0.258228, -0.495, -0.264, -0.432, -0.416, -0.877, -0.3};
    for (int k = 0; k < 10; ++k) { out->coeff_x[k] = x[k]; out->coeff_y[k] = y[k]; out->coeff_z[k] = z[k]; }
    out->center_camera[0] = -0.005;
    out->center_camera[1] = 0.262;
    out->center_camera[2] = -0.366 + 0.12;
    out->rotation_axis[0] = 0.304289493528802;
    out->rotation_axis[1] = 0.760492682863655;
    out->rotation_axis[2] = 0.573636455813981;
    out->rotation_angle = 1.78268191887446;
    out->scale = 1.;
    out->color_transform = SAR_CT_POISSON_SATURNE;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_config_solar_sail(sar_config* out) try {
    if (!out) return SAR_ERR_INVALID;
    sar::config_defaults(out);
    // values: src/lib.rs:356-385
    static const double x[10] = {0.744304, -0.546835, 0.121519, -0.653165, 0.399, 0.379, 0.44, 1.014, -0.805063, 0.377};
    static const double y[10] = {-0.683, 0.531646, -0.04557, -1.2, -0.546835, 0.091139, 0.744304, -0.273418,
                                 -0.349367, -0.531646};
    static const double z[10] = {0.712, 0.744304, -0.577215, 0.966, 0.04557, 1.063291, 0.01519, -0.425316, 0.212658,
                                 -0.01519};
    for (int k = 0; k < 10; ++k) { out->coeff_x[k] = x[k]; out->coeff_y[k] = y[k]; out->coeff_z[k] = z[k]; }
    out->center_camera[0] = 0.28;
    out->center_camera[1] = -0.12;
    out->center_camera[2] = 0.22;
    out->rotation_axis[0] = 0.02466;
    out->rotation_axis[1] = 0.4618;
    out->rotation_axis[2] = -0.54789;
    out->rotation_angle = 2.2195;
    out->scale = 1.7;
    out->color_transform = SAR_CT_ADJUSTED_VELOCITY;
    out->ct_factor = -0.2;
    out->ct_offset = 0.8;
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_config_validate(const sar_config* cfg) { return sar::validate(cfg); }

int sar_rotation_matrix(const sar_config* cfg, double m_out[9]) try {
    if (!cfg || !m_out) return SAR_ERR_INVALID;
    sar::rotation_matrix(*cfg, m_out);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_start_points(uint64_t seed, uint64_t first_job, uint32_t n_jobs, double* xyz_out_host) try {
    if (!xyz_out_host && n_jobs) return SAR_ERR_INVALID;
    // reaching job k takes k / 4096 jumps of ~1 us each: 2^36 jobs (seconds) is far beyond any frame list and still bounded
    if (first_job > (1ull << 36)) { sar::set_error("sar_start_points: first_job beyond 2^36"); return SAR_ERR_RANGE; }
    sar::Rng rng;
    rng.seed(seed);
    rng.skip_points(first_job);
    for (uint32_t k = 0; k < n_jobs; ++k) rng.start_point(xyz_out_host + 3 * static_cast<size_t>(k));
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
