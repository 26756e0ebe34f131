// sar.hpp — header-only C++17 mirror of the reference crate's surface over the C ABI (sar.h).
//
// Names follow Icelk/strange-attractor-renderer (src/lib.rs): Config (:265), Runtime (:631), render (:747),
// colorize (:841), ParallelRenderer (:908), render_parallel (:1051). Errors become exceptions here (the
// reference panics); nothing throws across the C boundary itself.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "sar.h"

namespace sar {

struct Error : std::runtime_error {
    int status;
    Error(int s, const char* where)
        : std::runtime_error(std::string(where) + ": " + sar_status_string(s) + " — " + sar_last_error()), status(s) {}
};
inline void check(int status, const char* where) {
    if (status != SAR_OK) throw Error(status, where);
}

enum class RenderKind : int32_t { Gas = SAR_RENDER_GAS, Depth = SAR_RENDER_DEPTH };  // :233-239

// 16-bit RGBA image, row-major (FinalImage, :625)
struct FinalImage {
    uint32_t width = 0, height = 0;
    std::vector<uint16_t> rgba;
};

struct Config : sar_config {
    static Config poisson_saturne() {  // :310
        Config c;
        check(sar_config_poisson_saturne(&c), "Config::poisson_saturne");
        return c;
    }
    static Config solar_sail() {  // :355
        Config c;
        check(sar_config_solar_sail(&c), "Config::solar_sail");
        return c;
    }
    void validate() const { check(sar_config_validate(this), "Config::validate"); }
};

class Runtime {  // :631
public:
    explicit Runtime(const Config& config, int device = 0) { check(sar_runtime_new(&config, device, &rt_), "Runtime::new"); }
    ~Runtime() { sar_runtime_free(rt_); }
    Runtime(const Runtime&) = delete;
    Runtime& operator=(const Runtime&) = delete;
    Runtime(Runtime&& o) noexcept : rt_(std::exchange(o.rt_, nullptr)) {}

    void reset() { check(sar_runtime_reset(rt_), "Runtime::reset"); }                      // :682
    void merge(const Runtime& other) { check(sar_runtime_merge(rt_, other.rt_), "Runtime::merge"); }  // :708
    void seed(uint64_t s) { check(sar_runtime_seed(rt_, s), "Runtime::seed"); }
    uint32_t max() { uint32_t m = 0; check(sar_runtime_max(rt_, &m), "Runtime::max"); return m; }
    std::vector<uint32_t> count() {
        uint32_t w = 0, h = 0;
        check(sar_runtime_dims(rt_, &w, &h), "Runtime::dims");
        std::vector<uint32_t> out(static_cast<size_t>(w) * h);
        check(sar_runtime_count(rt_, out.data()), "Runtime::count");
        return out;
    }
    sar_runtime* handle() const { return rt_; }

private:
    sar_runtime* rt_ = nullptr;
};

// render(&config, &mut runtime): one trajectory of config.iterations (:747)
inline void render(const Config& config, Runtime& runtime) { check(sar_render(&config, runtime.handle()), "render"); }
// config.jobs_total trajectories with the sequential semantics of calling render that many times
inline void render_jobs(const Config& config, Runtime& runtime, const double* starts_xyz = nullptr) {
    check(sar_render_jobs(&config, runtime.handle(), starts_xyz), "render_jobs");
}
// F frames of a sweep (the CLI's frame loop, src/bin/main.rs:493-517) through ONE set of launches: frame i is
// render_jobs(configs[i], *runtimes[i], starts[i]) — bit for bit; the frames share the chip instead of following each other
inline void render_jobs_batch(const std::vector<const Config*>& configs, const std::vector<Runtime*>& runtimes,
                              const std::vector<const double*>& starts_xyz = {}) {
    if (configs.size() != runtimes.size() || (!starts_xyz.empty() && starts_xyz.size() != configs.size()))
        throw Error(SAR_ERR_INVALID, "render_jobs_batch: configs, runtimes and starts must have the same length");
    std::vector<const sar_config*> c(configs.begin(), configs.end());
    std::vector<sar_runtime*> r;
    for (Runtime* rt : runtimes) r.push_back(rt->handle());
    check(sar_render_jobs_batch(static_cast<uint32_t>(c.size()), c.data(), r.data(), starts_xyz.empty() ? nullptr : starts_xyz.data()),
          "render_jobs_batch");
}
// colorize(&config, &runtime) -> FinalImage (:841)
inline FinalImage colorize(const Config& config, Runtime& runtime) {
    FinalImage img;
    img.width = config.width;
    img.height = config.height;
    img.rgba.resize(static_cast<size_t>(config.width) * config.height * 4);
    check(sar_colorize(&config, runtime.handle(), img.rgba.data()), "colorize");
    return img;
}

// write_image_matches (src/bin/main.rs:40-100): colorize, convert by (transparent, 8bit) on the device, encode by
// (pam, bmp) — both need 8bit (:256-258) — and replace the extension of `name`. Returns the path written.
inline std::string write_image_matches(const Config& config, Runtime& runtime, const std::string& name, bool eight_bit = false,
                                       bool pam = false, bool bmp = false) {
    if ((pam || bmp) && !eight_bit) throw Error(SAR_ERR_INVALID, "write_image_matches: --pam/--bmp require --8bit");
    const int format = sar_image_format(config.transparent, eight_bit ? 1 : 0);
    std::vector<unsigned char> pixels(sar_image_bytes(format, config.width, config.height));
    check(sar_colorize_format(&config, runtime.handle(), format, pixels.data()), "colorize_format");
    const size_t dot = name.find_last_of('.'), slash = name.find_last_of('/');
    const std::string stem = (dot != std::string::npos && (slash == std::string::npos || dot > slash)) ? name.substr(0, dot) : name;
    const std::string path = stem + (pam ? ".pam" : (bmp ? ".bmp" : ".png"));
    check((pam ? sar_write_pam : (bmp ? sar_write_bmp : sar_write_png))(path.c_str(), format, config.width, config.height, pixels.data()),
          "write_image");
    return path;
}

class ParallelRenderer {  // :908
public:
    explicit ParallelRenderer(int device = 0, uint32_t units = 0, uint64_t seed = 0) {
        check(sar_renderer_new(device, units, seed, &r_), "ParallelRenderer::new");
    }
    // every GPU of the node behind one renderer (jobs sharded over the devices, partial buffers merged over xGMI)
    explicit ParallelRenderer(const std::vector<int>& devices, uint32_t units = 0, uint64_t seed = 0) {
        check(sar_renderer_new_multi(devices.data(), static_cast<uint32_t>(devices.size()), units, seed, &r_), "ParallelRenderer::new_multi");
    }
    uint32_t num_devices() const { uint32_t n = 0; check(sar_renderer_num_devices(r_, &n), "num_devices"); return n; }
    ~ParallelRenderer() { shutdown(); }
    ParallelRenderer(const ParallelRenderer&) = delete;
    ParallelRenderer& operator=(const ParallelRenderer&) = delete;
    uint32_t num_threads() const { uint32_t n = 0; check(sar_renderer_num_units(r_, &n), "num_threads"); return n; }
    void shutdown() { sar_renderer_shutdown(r_); r_ = nullptr; }  // :1020
    sar_renderer* handle() const { return r_; }

private:
    sar_renderer* r_ = nullptr;
};

// render_parallel(&mut renderer, config, jobs_per_thread) -> FinalImage (:1051)
inline FinalImage render_parallel(ParallelRenderer& renderer, const Config& config, uint32_t jobs_per_thread) {
    FinalImage img;
    img.width = config.width;
    img.height = config.height;
    img.rgba.resize(static_cast<size_t>(config.width) * config.height * 4);
    check(sar_render_parallel(renderer.handle(), &config, jobs_per_thread, img.rgba.data()), "render_parallel");
    return img;
}

}  // namespace sar
