// sar_export.cpp — image export of the CLI's write_image_matches (reference src/bin/main.rs:40-100): format choice,
// PNG / BMP / PAM encoders. Host only. The reference delegates all of this to the `image` crate (0.25, unpinned, not
// vendored): what is restated here is the crate's published behaviour; the parity bar is "the file decodes to the
// same samples"  and tests/test_export.py decodes every file it writes.
#include <zlib.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "sar_internal.hpp"

namespace {

using sar::set_error;

struct Layout {
    uint32_t channels, bytes_per_sample;
};
bool layout_of(int format, Layout& l) {
    switch (format) {
        case SAR_FMT_RGBA16: l = {4, 2}; return true;
        case SAR_FMT_RGB16: l = {3, 2}; return true;
        case SAR_FMT_RGBA8: l = {4, 1}; return true;
        case SAR_FMT_RGB8: l = {3, 1}; return true;
        default: return false;
    }
}

struct File {
    FILE* f = nullptr;
    explicit File(const char* path) : f(path ? std::fopen(path, "wb") : nullptr) {}
    ~File() { if (f) std::fclose(f); }
    bool put(const void* p, size_t n) { return std::fwrite(p, 1, n, f) == n; }
};
int io_error(const char* path) {
    set_error("cannot write '%s'", path ? path : "(null)");
    return SAR_ERR_IO;
}

void be32(unsigned char* p, uint32_t v) {
    p[0] = static_cast<unsigned char>(v >> 24);
    p[1] = static_cast<unsigned char>(v >> 16);
    p[2] = static_cast<unsigned char>(v >> 8);
    p[3] = static_cast<unsigned char>(v);
}
void le32(unsigned char* p, uint32_t v) {
    p[0] = static_cast<unsigned char>(v);
    p[1] = static_cast<unsigned char>(v >> 8);
    p[2] = static_cast<unsigned char>(v >> 16);
    p[3] = static_cast<unsigned char>(v >> 24);
}
void le16(unsigned char* p, uint32_t v) {
    p[0] = static_cast<unsigned char>(v);
    p[1] = static_cast<unsigned char>(v >> 8);
}

bool png_chunk(File& out, const char type[4], const unsigned char* data, size_t n) {
    unsigned char head[8];
    be32(head, static_cast<uint32_t>(n));
    std::memcpy(head + 4, type, 4);
    uLong crc = crc32(0L, head + 4, 4);
    if (n) crc = crc32(crc, data, static_cast<uInt>(n));
    unsigned char tail[4];
    be32(tail, static_cast<uint32_t>(crc));
    return out.put(head, 8) && (n == 0 || out.put(data, n)) && out.put(tail, 4);
}

// PNG filter types 0..4 applied to one row (bpp = bytes per complete pixel); returns the sum of absolute values of
// the filtered bytes read as signed — the "minimum sum of absolute differences" heuristic of the PNG specification,
// which is what an adaptive encoder minimises per row.
uint64_t png_filter_row(int type, const unsigned char* cur, const unsigned char* up, size_t n, size_t bpp, unsigned char* dst) {
    uint64_t score = 0;
    for (size_t i = 0; i < n; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0;
        const int b = up ? up[i] : 0;
        const int c = (up && i >= bpp) ? up[i - bpp] : 0;
        int pred = 0;
        switch (type) {
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c;
                const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: break;
        }
        const unsigned char v = static_cast<unsigned char>(cur[i] - pred);
        dst[i] = v;
        score += v < 128 ? v : 256 - v;
    }
    return score;
}

}  // namespace

extern "C" {

int sar_image_format(int transparent, int eight_bit) try {
    // src/bin/main.rs:52-57
    if (transparent) return eight_bit ? SAR_FMT_RGBA8 : SAR_FMT_RGBA16;
    return eight_bit ? SAR_FMT_RGB8 : SAR_FMT_RGB16;
} catch (...) { return sar::abi_caught(); }

size_t sar_image_bytes(int format, uint32_t width, uint32_t height) {
    Layout l;
    if (!layout_of(format, l)) return 0;
    return static_cast<size_t>(width) * height * l.channels * l.bytes_per_sample;
}

int sar_write_png(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) try {
    Layout l;
    if (!path || !pixels || !layout_of(format, l) || width == 0 || height == 0) {
        set_error("sar_write_png: bad argument");
        return SAR_ERR_INVALID;
    }
    const size_t bpp = static_cast<size_t>(l.channels) * l.bytes_per_sample;
    const size_t row = static_cast<size_t>(width) * bpp;
    const unsigned char* src = static_cast<const unsigned char*>(pixels);

    // The zlib stream is produced in bands of kBandRows scanlines that are filtered and deflated independently — raw
    // deflate, every band but the last closed with a sync flush (byte aligned), adler32 combined afterwards — so that
    // the bands can be encoded by several threads (a 1920x1080 RGB16 frame takes 0.3 s on one thread, 40x the time
    // the GPU needs to render it). The band size is fixed: the file does not depend on the number of threads.
    constexpr uint32_t kBandRows = 64;
    const uint32_t bands = (height + kBandRows - 1) / kBandRows;
    struct Band {
        std::vector<unsigned char> z;
        uLong adler = 1;
        uLong raw_len = 0;
        bool ok = false;
    };
    std::vector<Band> out(bands);
    // scanline in PNG byte order (16-bit samples are big-endian in the file)
    auto load_row = [&](uint32_t y, unsigned char* dst) {
        const unsigned char* in = src + static_cast<size_t>(y) * row;
        if (l.bytes_per_sample == 2) {
            const uint16_t* s16 = reinterpret_cast<const uint16_t*>(in);
            for (size_t k = 0; k < row / 2; ++k) {
                dst[2 * k] = static_cast<unsigned char>(s16[k] >> 8);
                dst[2 * k + 1] = static_cast<unsigned char>(s16[k]);
            }
        } else {
            std::memcpy(dst, in, row);
        }
    };
    auto encode_band = [&](uint32_t b) {
        Band& o = out[b];
        const uint32_t y0 = b * kBandRows, y1 = (y0 + kBandRows < height) ? y0 + kBandRows : height;
        std::vector<unsigned char> prev(row), cur(row), best(row), trial(row), filtered;
        filtered.reserve((row + 1) * (y1 - y0));
        if (y0) load_row(y0 - 1, prev.data());
        for (uint32_t y = y0; y < y1; ++y) {
            load_row(y, cur.data());
            // png::FilterType::Adaptive (:89): per row, the filter with the smallest sum of absolute differences
            int best_type = 0;
            uint64_t best_score = png_filter_row(0, cur.data(), y ? prev.data() : nullptr, row, bpp, best.data());
            for (int t = 1; t <= 4; ++t) {
                const uint64_t sc = png_filter_row(t, cur.data(), y ? prev.data() : nullptr, row, bpp, trial.data());
                if (sc < best_score) {
                    best_score = sc;
                    best_type = t;
                    best.swap(trial);
                }
            }
            filtered.push_back(static_cast<unsigned char>(best_type));
            filtered.insert(filtered.end(), best.begin(), best.end());
            prev.swap(cur);
        }
        o.raw_len = static_cast<uLong>(filtered.size());
        o.adler = adler32(adler32(0L, Z_NULL, 0), filtered.data(), static_cast<uInt>(filtered.size()));
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        // png::CompressionType::Default (:88) = zlib's default level; raw deflate (the zlib wrapper is written once)
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
        o.z.resize(deflateBound(&zs, o.raw_len) + 16);
        zs.next_in = filtered.data();
        zs.avail_in = static_cast<uInt>(filtered.size());
        zs.next_out = o.z.data();
        zs.avail_out = static_cast<uInt>(o.z.size());
        const bool last = b + 1 == bands;
        const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
        o.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0);
        o.z.resize(o.z.size() - zs.avail_out);
        deflateEnd(&zs);
    };
    {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned nthreads = hw ? hw : 1;
        if (nthreads > 16) nthreads = 16;
        if (nthreads > bands) nthreads = bands;
        std::atomic<uint32_t> next{0};
        auto worker = [&]() {
            for (uint32_t b = next.fetch_add(1); b < bands; b = next.fetch_add(1)) encode_band(b);
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    }
    std::vector<unsigned char> idat;
    idat.push_back(0x78);  // zlib header: deflate, 32 KiB window; FLEVEL = default; FCHECK makes it a multiple of 31
    idat.push_back(0x9C);
    uLong adler = adler32(0L, Z_NULL, 0);
    for (uint32_t b = 0; b < bands; ++b) {
        if (!out[b].ok) {
            set_error("deflate failed in band %u", b);
            return SAR_ERR_OOM;
        }
        idat.insert(idat.end(), out[b].z.begin(), out[b].z.end());
        adler = adler32_combine(adler, out[b].adler, static_cast<z_off_t>(out[b].raw_len));
    }
    unsigned char ad[4];
    be32(ad, static_cast<uint32_t>(adler));
    idat.insert(idat.end(), ad, ad + 4);

    File fout(path);
    if (!fout.f) return io_error(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    unsigned char ihdr[13];
    be32(ihdr, width);
    be32(ihdr + 4, height);
    ihdr[8] = static_cast<unsigned char>(8 * l.bytes_per_sample);  // bit depth
    ihdr[9] = l.channels == 4 ? 6 : 2;                              // colour type: RGBA / RGB
    ihdr[10] = ihdr[11] = ihdr[12] = 0;                             // deflate, adaptive filtering, no interlace
    bool ok = fout.put(sig, 8) && png_chunk(fout, "IHDR", ihdr, 13);
    // one IDAT chunk per 1 GiB at most (a chunk length is 31 bits)
    for (size_t off = 0; ok && off < idat.size(); off += (1u << 30)) {
        const size_t n = idat.size() - off < (1u << 30) ? idat.size() - off : (1u << 30);
        ok = png_chunk(fout, "IDAT", idat.data() + off, n);
    }
    ok = ok && png_chunk(fout, "IEND", nullptr, 0);
    return ok ? SAR_OK : io_error(path);
} catch (...) { return sar::abi_caught(); }

int sar_write_bmp(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) try {
    if (!path || !pixels || width == 0 || height == 0 || (format != SAR_FMT_RGB8 && format != SAR_FMT_RGBA8)) {
        set_error("sar_write_bmp: 8-bit RGB or RGBA only (the CLI requires --8bit with --bmp, main.rs:256-258)");
        return SAR_ERR_INVALID;
    }
    const bool alpha = format == SAR_FMT_RGBA8;
    const uint32_t bpp = alpha ? 4 : 3;
    const uint32_t row = (width * bpp + 3u) & ~3u;  // rows are padded to 4 bytes
    const uint32_t dib = alpha ? 108u : 40u;        // BITMAPV4HEADER carries the channel masks, BITMAPINFOHEADER does not
    const uint64_t image_bytes = static_cast<uint64_t>(row) * height;
    if (14ull + dib + image_bytes > 0xFFFFFFFFull) {
        set_error("image too large for BMP");
        return SAR_ERR_RANGE;
    }
    std::vector<unsigned char> head(14 + dib, 0);
    head[0] = 'B';
    head[1] = 'M';
    le32(&head[2], static_cast<uint32_t>(14 + dib + image_bytes));
    le32(&head[10], 14 + dib);
    unsigned char* d = &head[14];
    le32(d, dib);
    le32(d + 4, width);
    le32(d + 8, height);  // positive height: bottom-up
    le16(d + 12, 1);
    le16(d + 14, 8 * bpp);
    le32(d + 16, alpha ? 3u : 0u);  // BI_BITFIELDS / BI_RGB
    le32(d + 20, static_cast<uint32_t>(image_bytes));
    le32(d + 24, 2835);  // 72 dpi
    le32(d + 28, 2835);
    if (alpha) {
        le32(d + 40, 0x00FF0000u);  // red, green, blue, alpha masks of a B,G,R,A byte order
        le32(d + 44, 0x0000FF00u);
        le32(d + 48, 0x000000FFu);
        le32(d + 52, 0xFF000000u);
        le32(d + 56, 0x73524742u);  // "sRGB"
    }
    File out(path);
    if (!out.f) return io_error(path);
    if (!out.put(head.data(), head.size())) return io_error(path);
    std::vector<unsigned char> line(row, 0);
    const unsigned char* src = static_cast<const unsigned char*>(pixels);
    for (uint32_t y = height; y-- > 0;) {
        const unsigned char* in = src + static_cast<size_t>(y) * width * bpp;
        for (uint32_t x = 0; x < width; ++x) {
            line[x * bpp] = in[x * bpp + 2];
            line[x * bpp + 1] = in[x * bpp + 1];
            line[x * bpp + 2] = in[x * bpp];
            if (alpha) line[x * bpp + 3] = in[x * bpp + 3];
        }
        if (!out.put(line.data(), row)) return io_error(path);
    }
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

int sar_write_pam(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) try {
    if (!path || !pixels || width == 0 || height == 0 || (format != SAR_FMT_RGB8 && format != SAR_FMT_RGBA8)) {
        set_error("sar_write_pam: 8-bit RGB or RGBA only (the CLI requires --8bit with --pam, main.rs:256-258)");
        return SAR_ERR_INVALID;
    }
    const bool alpha = format == SAR_FMT_RGBA8;
    char head[128];
    // PnmSubtype::ArbitraryMap (:68)
    const int n = std::snprintf(head, sizeof(head), "P7\nWIDTH %u\nHEIGHT %u\nDEPTH %u\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n", width,
                                height, alpha ? 4u : 3u, alpha ? "RGB_ALPHA" : "RGB");
    File out(path);
    if (!out.f) return io_error(path);
    if (!out.put(head, static_cast<size_t>(n)) || !out.put(pixels, sar_image_bytes(format, width, height))) return io_error(path);
    return SAR_OK;
} catch (...) { return sar::abi_caught(); }

}  // extern "C"
