// sar_export.cpp — image export of the CLI's write_image_matches (reference src/bin/main.rs:40-100): format choice,
// PNG / BMP / PAM encoders. Host only. The reference delegates all of this to the `image` crate (0.25, unpinned, not
// vendored): what is restated here is the crate's published behaviour; the parity bar is "the file decodes to the
// same samples"  and tests/test_export.py decodes every file it writes.
#include <zlib.h>

#include <cstdio>
#include <cstring>
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

int sar_image_format(int transparent, int eight_bit) {
    // src/bin/main.rs:52-57
    if (transparent) return eight_bit ? SAR_FMT_RGBA8 : SAR_FMT_RGBA16;
    return eight_bit ? SAR_FMT_RGB8 : SAR_FMT_RGB16;
}

size_t sar_image_bytes(int format, uint32_t width, uint32_t height) {
    Layout l;
    if (!layout_of(format, l)) return 0;
    return static_cast<size_t>(width) * height * l.channels * l.bytes_per_sample;
}

int sar_write_png(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) {
    Layout l;
    if (!path || !pixels || !layout_of(format, l) || width == 0 || height == 0) {
        set_error("sar_write_png: bad argument");
        return SAR_ERR_INVALID;
    }
    const size_t bpp = static_cast<size_t>(l.channels) * l.bytes_per_sample;
    const size_t row = static_cast<size_t>(width) * bpp;
    // scanlines with PNG byte order (16-bit samples are big-endian in the file), each behind its filter byte
    std::vector<unsigned char> prev(row), cur(row), best(row), trial(row);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (deflateInit(&zs, Z_DEFAULT_COMPRESSION) != Z_OK) {  // png::CompressionType::Default (:88)
        set_error("deflateInit failed");
        return SAR_ERR_OOM;
    }
    std::vector<unsigned char> idat;
    std::vector<unsigned char> zbuf(1u << 16);
    auto pump = [&](int flush) {
        int rc;
        do {
            zs.next_out = zbuf.data();
            zs.avail_out = static_cast<uInt>(zbuf.size());
            rc = deflate(&zs, flush);
            idat.insert(idat.end(), zbuf.data(), zbuf.data() + (zbuf.size() - zs.avail_out));
        } while (zs.avail_out == 0 && rc != Z_STREAM_END);
    };
    const unsigned char* src = static_cast<const unsigned char*>(pixels);
    for (uint32_t y = 0; y < height; ++y) {
        const unsigned char* in = src + static_cast<size_t>(y) * row;
        if (l.bytes_per_sample == 2) {
            const uint16_t* s16 = reinterpret_cast<const uint16_t*>(in);
            for (size_t k = 0; k < row / 2; ++k) {
                cur[2 * k] = static_cast<unsigned char>(s16[k] >> 8);
                cur[2 * k + 1] = static_cast<unsigned char>(s16[k]);
            }
        } else {
            std::memcpy(cur.data(), in, row);
        }
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
        unsigned char ft = static_cast<unsigned char>(best_type);
        zs.next_in = &ft;
        zs.avail_in = 1;
        pump(Z_NO_FLUSH);
        zs.next_in = best.data();
        zs.avail_in = static_cast<uInt>(row);
        pump(Z_NO_FLUSH);
        prev.swap(cur);
    }
    zs.next_in = nullptr;
    zs.avail_in = 0;
    pump(Z_FINISH);
    deflateEnd(&zs);

    File out(path);
    if (!out.f) return io_error(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    unsigned char ihdr[13];
    be32(ihdr, width);
    be32(ihdr + 4, height);
    ihdr[8] = static_cast<unsigned char>(8 * l.bytes_per_sample);  // bit depth
    ihdr[9] = l.channels == 4 ? 6 : 2;                              // colour type: RGBA / RGB
    ihdr[10] = ihdr[11] = ihdr[12] = 0;                             // deflate, adaptive filtering, no interlace
    bool ok = out.put(sig, 8) && png_chunk(out, "IHDR", ihdr, 13);
    // one IDAT chunk per 1 GiB at most (a chunk length is 31 bits)
    for (size_t off = 0; ok && off < idat.size(); off += (1u << 30)) {
        const size_t n = idat.size() - off < (1u << 30) ? idat.size() - off : (1u << 30);
        ok = png_chunk(out, "IDAT", idat.data() + off, n);
    }
    ok = ok && png_chunk(out, "IEND", nullptr, 0);
    return ok ? SAR_OK : io_error(path);
}

int sar_write_bmp(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) {
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
}

int sar_write_pam(const char* path, int format, uint32_t width, uint32_t height, const void* pixels) {
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
}

}  // extern "C"
