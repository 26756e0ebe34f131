"""CPU: the C-ABI library loads, exports everything include/sar.h declares, agrees with the ctypes mirror on
the struct layout, and its host-side logic (presets, setup math, start-point stream, validation, error
behaviour) matches the oracle / the reference's data. No compute calls: there is no GPU here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sar.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sar_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sar):
    from strange_attractor_renderer_amd import _abi
    lib = sar.load_library()
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sar.h but not exported by libsar_hip.so"
    assert sorted(_abi.PROTOTYPES) == names, "ctypes prototypes and header declarations differ"
    assert lib.sar_abi_version() == int(re.search(r"#define\s+SAR_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))


def test_struct_layout_matches_c(sar):
    from strange_attractor_renderer_amd._abi import SarConfig, SarTiming
    fields = ["iterations", "width", "render_kind", "angle", "coeff_x", "coeff_z", "palette_len", "palette_rgb",
              "brightness_offset", "center_camera", "rotation_axis", "rotation_angle", "scale", "color_transform",
              "ct_offset", "seed", "jobs_total"]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "sar.h"\nint main(void){\n'
    prog += 'printf("%zu %zu\\n", sizeof(sar_config), sizeof(sar_timing));\n'
    for f in fields:
        prog += f'printf("%zu\\n", offsetof(sar_config, {f}));\n'
    prog += "return 0;}\n"
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe],
                       check=True)  # the header must be plain C
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == C.sizeof(SarConfig) and int(out[1]) == C.sizeof(SarTiming)
    for f, off in zip(fields, out[2:]):
        assert getattr(SarConfig, f).offset == int(off), f


def test_presets_are_the_reference_values(sar, oracle):
    assert bytes(sar.Config.poisson_saturne().c) == bytes(oracle.poisson_saturne())
    assert bytes(sar.Config.solar_sail().c) == bytes(oracle.solar_sail())
    c = sar.Config.poisson_saturne()
    # Config::new defaults (src/lib.rs:289-307)
    assert (c.iterations, c.width, c.height, c.transparent, c.angle, c.silent) == (10_000_000, 1920, 1080, 1, 0.0, 1)
    assert c.render_kind == sar.SAR_RENDER_GAS and c.palette_len == 6
    assert c.brightness_offset == -0.15 and c.brightness_factor == 5.0 / 3.0
    assert c.center_camera[2] == -0.366 + 0.12 and c.scale == 1.0
    s = sar.Config.solar_sail()
    assert s.scale == 1.7 and s.color_transform == sar.SAR_CT_ADJUSTED_VELOCITY
    assert (s.ct_offset, s.ct_factor) == (0.8, -0.2)
    # `Config { iterations: 100_000_000, ..Config::poisson_saturne() }` (the reference's only doc-test, :9-15)
    d = sar.Config.poisson_saturne(iterations=100_000_000)
    assert d.iterations == 100_000_000 and d.width == 1920
    d.validate()


def test_setup_math_matches_oracle_bit_for_bit(sar, oracle):
    for mk_s, mk_o in ((sar.Config.poisson_saturne, oracle.poisson_saturne), (sar.Config.solar_sail, oracle.solar_sail)):
        a = mk_s().rotation_matrix()
        b = oracle.rotation_matrix(mk_o())
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    # the start-point stream: blocks of 4096 jobs, block b = the generator jumped b times (also across block boundaries,
    # from inside a block, and far into the stream)
    for seed, first, n in ((0, 0, 7), (1, 5, 2), (2**63 + 12345, 1000, 33), (7, 0, 3 * 4096 + 5), (7, 4090, 12), (7, 4096, 1),
                           (7, 8191, 4100), (2**64 - 1, 1_048_576 - 3, 10), (3, 5 * 131072, 4096)):
        assert np.array_equal(sar.start_points(seed, first, n), oracle.start_points(seed, first, n)), (seed, first, n)


def test_validation_and_error_reporting(sar):
    lib = sar.load_library()
    with pytest.raises(sar.SarError) as e:   # a start point 2^40 jobs into the stream: refused, not computed for minutes
        sar.start_points(1, 1 << 40, 1)
    assert e.value.status == 6
    for bad in (dict(width=0), dict(height=0), dict(render_kind=7), dict(color_transform=5), dict(palette_len=0),
                dict(palette_len=16), dict(attractor_kind=3), dict(width=65536, height=65536)):
        with pytest.raises(sar.SarError) as e:
            sar.Config.poisson_saturne(**bad).validate()
        assert e.value.status in (1, 6)
    assert lib.sar_config_validate(None) == 1
    assert lib.sar_config_poisson_saturne(None) == 1
    assert lib.sar_status_string(2).decode() == "runtime dimensions differ"
    assert lib.sar_status_string(12345).decode() == "unknown status"
    assert lib.sar_runtime_free(None) == 0 and lib.sar_renderer_shutdown(None) == 0
    assert lib.sar_runtime_reset(None) == 1 and lib.sar_runtime_merge(None, None) == 1


def test_no_device_is_an_error_not_a_fallback(sar):
    if sar.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(sar.SarError) as e:
        sar.Runtime(sar.Config.poisson_saturne(width=8, height=8))
    assert e.value.status == 3  # SAR_ERR_NO_DEVICE
    with pytest.raises(sar.SarError) as e:
        sar.ParallelRenderer()
    assert e.value.status == 3


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may import, link or load it."""
    pkg = os.path.join(ROOT, "strange_attractor_renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_lib" not in text and "libsar_oracle" not in text and "sar_oracle" not in text, f
    out = subprocess.run(["ldd", os.path.join(pkg, "libsar_hip.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_missing_library_fails_loudly(sar):
    from strange_attractor_renderer_amd import _abi
    with pytest.raises(_abi.SarLibraryMissing):
        _abi.load_library("/nonexistent/libsar_hip.so")


def test_cpp_header_mirror_compiles_links_and_runs(sar):
    """include/sar.hpp (the C++ mirror of the crate's names) builds against the library; host-only calls work and a
    missing device surfaces as sar::Error, not a crash."""
    prog = r'''
#include <cstdio>
#include "sar.hpp"
int main() {
    auto c = sar::Config::poisson_saturne();
    c.iterations = 1000; c.width = 16; c.height = 16; c.validate();
    auto s = sar::Config::solar_sail();
    if (c.coeff_x[1] != 1.182 || s.scale != 1.7) return 2;
    int n = 0; sar_device_count(&n);
    try { sar::Runtime rt(c); sar::render(c, rt); auto img = sar::colorize(c, rt); if (img.rgba.size() != 16*16*4) return 3; }
    catch (const sar::Error& e) { if (n > 0 || e.status != SAR_ERR_NO_DEVICE) return 4; }
    std::puts("ok");
    return 0;
}
'''
    pkg = os.path.join(ROOT, "strange_attractor_renderer_amd")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(prog)
        exe = os.path.join(d, "t")
        subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
                        "-L", pkg, "-l:libsar_hip.so", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
        out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)


def test_rust_binding_source_declares_every_abi_function():
    """bindings/rust is source-only (no Rust toolchain here); at least keep it in step with the header."""
    rs = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (sar_[a-z0-9_]+)\s*\(", rs))
    assert declared == set(declared_functions())
    # field order of the #[repr(C)] mirror == the ctypes mirror == the C header
    from strange_attractor_renderer_amd._abi import SarConfig, SarTiming
    body = rs[rs.index("pub struct SarConfig {"):]
    body = body[:body.index("}")]
    assert re.findall(r"pub (\w+):", body) == [f for f, _ in SarConfig._fields_]
    body = rs[rs.index("pub struct SarTiming {"):]
    body = body[:body.index("}")]
    assert re.findall(r"pub (\w+):", body) == [f for f, _ in SarTiming._fields_]


def test_graft_entry_build_runs():
    """The driver's build check: __graft_entry__.build() compiles (or finds up to date) both libraries and loads the
    product — without a GPU."""
    import importlib
    g = importlib.import_module("__graft_entry__")
    g.build()


def test_rust_safe_layer_calls_only_what_the_sys_crate_declares():
    """bindings/rust-safe cannot be compiled here either: at least every `sys::sar_*` call must name a function of the
    sys crate with the declared number of arguments, and every constant it uses must exist there."""
    sys_rs = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    safe_rs = open(os.path.join(ROOT, "bindings", "rust-safe", "src", "lib.rs")).read()
    decl = {}
    for m in re.finditer(r"pub fn (sar_[a-z0-9_]+)\s*\(([^)]*)\)", sys_rs, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        decl[m.group(1)] = len(args)
    calls = re.findall(r"sys::(sar_[a-z0-9_]+)\s*\(", safe_rs)
    assert calls, "the safe layer calls nothing?"
    for name in set(calls):
        assert name in decl, f"{name} is not declared in the sys crate"
    # argument counts: parse each call's parenthesised argument list (no nested commas except inside inner calls)
    for m in re.finditer(r"sys::(sar_[a-z0-9_]+)\s*\(", safe_rs):
        i, depth, n, seen = m.end(), 1, 0, False
        while depth:
            c = safe_rs[i]
            if c in "([": depth += 1
            elif c in ")]": depth -= 1
            elif c == "," and depth == 1: n += 1
            elif not c.isspace() and depth >= 1: seen = True
            i += 1
        assert (n + 1 if seen else 0) == decl[m.group(1)], f"{m.group(1)}: {n + 1} arguments passed, {decl[m.group(1)]} declared"
    for const in set(re.findall(r"sys::(SAR_[A-Z0-9_]+)", safe_rs)):
        assert re.search(rf"pub const {const}\b", sys_rs), f"{const} missing in the sys crate"
    # to_abi fills every field of sar_config the header declares (padding aside): a field added to the ABI and forgotten
    # here would reach the library as zero
    from strange_attractor_renderer_amd._abi import SarConfig, SarParallelTiming
    body = safe_rs[safe_rs.index("fn to_abi<"):]
    body = body[:body.index("\n}\n")]
    filled = set(re.findall(r"\bs\.(\w+)(?:\[\w+\])?\s*=", body))
    assert filled == {f for f, _ in SarConfig._fields_ if not f.startswith("_pad")}, filled ^ {f for f, _ in SarConfig._fields_ if not f.startswith("_pad")}
    tbody = sys_rs[sys_rs.index("pub struct SarParallelTiming {"):]
    tbody = tbody[:tbody.index("}")]
    assert re.findall(r"pub (\w+):", tbody) == [f for f, _ in SarParallelTiming._fields_]
    # a runtime handed out by the renderer is tied to the renderer's lifetime (it is freed with it)
    assert "pub fn runtime(&mut self) -> BorrowedRuntime<'_>" in safe_rs and "PhantomData<&'a mut GpuRenderer>" in safe_rs
    # the reference items the layer stands in for are all there
    for item in ("pub struct GpuRuntime", "pub struct GpuRenderer", "pub fn render<", "pub fn colorize<", "pub fn render_parallel<",
                 "impl Drop for GpuRuntime", "impl Drop for GpuRenderer", "pub fn check(", "pub fn new_multi("):
        assert item in safe_rs, item


@pytest.mark.parametrize("size", [(2048, 2048), (1800, 2000), (1920, 1080), (2560, 2560), (3072, 3072), (3840, 2160), (4096, 4096),
                                  (8192, 8192), (700, 500), (64, 48), (1, 1), (2048, 3), (8192, 6000), (9000, 9000)])
@pytest.mark.parametrize("bin_shift,interleave", [(0, 0), (12, 2), (14, 1), (14, 2), (15, 2), (16, 1), (16, 2)])
def test_bin_map_is_a_bijection_onto_bins_and_records(sar, size, bin_shift, interleave):
    """The pixel -> (bin, 16-bit record) map of the LDS-binned path (host arithmetic behind sar_bin_geometry, the same
    BinGeometry the launches use): every pixel gets a bin below the bin count and a record below the bin's size, the
    inverse formula of k_bin_accumulate / k_fold_resolve gives the pixel back, no two pixels share (bin, record), and with
    interleaved bins every bin gets the same number of 2048-pixel segments (+-1). Images without a binned geometry
    (too many bins) say so."""
    w, h = size
    g = sar.bin_geometry(w, h, bin_shift, interleave)
    npix = w * h
    if not g["ok"]:
        assert (npix + (1 << g["bin_shift"]) - 1) >> g["bin_shift"] > 1024 or g["bins"] > 1024
        return
    assert 12 <= g["bin_shift"] <= 16 and 1 <= g["bins"] <= 1024
    step = max(1, npix // 3_000_000)  # every pixel of the small shapes, a dense sample (plus the edges) of the 64-Mpx ones
    idx = np.unique(np.concatenate([np.arange(0, npix, step, dtype=np.int64), np.arange(max(0, npix - 70000), npix, dtype=np.int64),
                                    np.arange(0, min(npix, 70000), dtype=np.int64)]))
    low, seg, hi, bits = g["low_mask"], g["seg_shift"], g["hi_shift"], g["bin_bits"]
    b = (idx >> seg) & ((1 << bits) - 1)
    rec = (idx & low) | ((idx >> hi) & ~np.int64(low) & 0xFFFFFFFF)
    assert b.max() < g["bins"] and rec.max() < (1 << g["bin_shift"]) and rec.max() < 65536
    back = (rec & low) | (b << seg) | ((rec & ~np.int64(low)) << hi)
    np.testing.assert_array_equal(back, idx)
    if step == 1:
        key = b * 65536 + rec
        assert np.unique(key).size == idx.size
        if g["interleaved"] and npix >= 2048 * g["bins"]:
            segs = np.bincount(((np.arange(0, npix, 2048) >> seg) & ((1 << bits) - 1)).astype(np.int64), minlength=g["bins"])
            assert segs.max() - segs.min() <= 1
    if bin_shift == 0 and interleave == 0 and npix <= 16 << 20:
        assert g["interleaved"], "every shape up to 4096^2 gets interleaved bins by default"


def test_library_is_tied_to_its_sources(sar, tmp_path):
    """The binary carries the id of the sources it was built from (sar_build_id == build.source_id of the tree); a tree whose
    kernel file was touched has another id, and the loader refuses the binary until build() has run (__graft_entry__.build
    rebuilds on ids, not on mtimes). Variants live under build/variants, never next to the product."""
    import glob
    import shutil
    from strange_attractor_renderer_amd import _abi, build
    lib = sar.load_library()
    assert lib.sar_build_id().decode() == build.source_id(extra_flags=[]) == build.library_id(_abi.LIB_PATH)
    _abi.verify_library(lib)
    # a copy of the tree with one kernel file touched: another id, and the loaded binary is refused against it
    inc = tmp_path / "include"
    pkg = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "include"), inc)
    shutil.copytree(build.CSRC, pkg / "csrc")
    touched = pkg / "csrc" / "sar_iterate.hip"
    touched.write_text(touched.read_text() + "\n// touched\n")
    assert build.source_id(str(pkg / "csrc"), extra_flags=[]) != build.source_id(extra_flags=[])
    with pytest.raises(_abi.SarLibraryStale):
        _abi.verify_library(lib, csrc=str(pkg / "csrc"))
    # a header change counts as well (include/sar.h is hashed through csrc/../../include)
    before = build.source_id(str(pkg / "csrc"), extra_flags=[])
    (inc / "sar.h").write_text((inc / "sar.h").read_text() + "\n/* touched */\n")
    assert build.source_id(str(pkg / "csrc"), extra_flags=[]) != before
    assert build.source_id(extra_flags=["-DSAR_POOL_SPARE=2u"]) != build.source_id(extra_flags=[])   # and so do the flags
    assert glob.glob(os.path.join(os.path.dirname(_abi.LIB_PATH), "libsar_hip_*.so")) == [], "a variant build sits next to the product"


def test_test_hooks_are_not_in_the_product(sar):
    """include/sar.h is the product's whole ABI: sar_runtime_set_test_option (include/sar_test_hooks.h) is exported by the hooks
    build of the test-suite alone — the same object files plus one — and sar_runtime_set_option knows five stable options."""
    from strange_attractor_renderer_amd import _abi, build
    product = C.CDLL(_abi.LIB_PATH)
    hooks = C.CDLL(_abi.HOOKS_PATH)
    assert not hasattr(product, "sar_runtime_set_test_option") and hasattr(hooks, "sar_runtime_set_test_option")
    assert build.library_id(_abi.LIB_PATH) == build.library_id(_abi.HOOKS_PATH) == build.source_id(extra_flags=[])
    header = open(HEADER).read()
    assert "sar_runtime_set_test_option(" not in re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    doc = header[header.index("/* Options by name"):header.index("int sar_runtime_set_option(")]
    assert sorted(re.findall(r'^ \*   "([a-z_]+)"', doc, flags=re.M)) == sorted(_abi.STABLE_OPTIONS)
    with tempfile.TemporaryDirectory() as d:   # the hooks header is plain C as well
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "sar_test_hooks.h"\nint main(void) { int (*f)(sar_runtime*, const char*, uint64_t) = sar_runtime_set_test_option; return f ? 0 : 1; }\n')
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(d, "t.o")],
                       check=True)


def test_no_exception_unwinds_across_the_abi(sar):
    """every entry returns a status. Every int-returning entry point of csrc/ is a function-try-block whose handler
    turns an exception into a status (sar::abi_caught): held here on the source, and end to end through the hooks build's
    "debug_throw" — std::bad_alloc -> SAR_ERR_OOM, std::exception / anything else -> SAR_ERR_INVALID, text in sar_last_error()."""
    from strange_attractor_renderer_amd import _abi
    csrc = os.path.join(ROOT, "strange_attractor_renderer_amd", "csrc")
    sources = {f: open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith(".cpp")}
    entries = guarded = 0
    for f, text in sources.items():
        for m in re.finditer(r"^int (sar_[a-z0-9_]+)\([^;{]*?\)\s*(try\s*)?\{([^\n]*)$", text, flags=re.M | re.S):
            entries += 1
            one_liner = m.group(3).rstrip().endswith("}")           # `{ return a constant or sar::validate(cfg); }`
            assert m.group(2) or one_liner, f"{f}: {m.group(1)} is not a function-try-block"
            guarded += bool(m.group(2))
    assert entries >= 70 and guarded >= entries - 2, (entries, guarded)
    handlers = sum(len(re.findall(r"\} catch \(\.\.\.\) \{ return sar::abi_caught\(\); \}", text)) for text in sources.values())
    assert handlers == guarded
    hooks = C.CDLL(_abi.HOOKS_PATH)
    fn = hooks.sar_runtime_set_test_option
    fn.argtypes, fn.restype = [C.c_void_p, C.c_char_p, C.c_uint64], C.c_int
    hooks.sar_last_error.restype = C.c_char_p
    assert fn(None, b"debug_throw", 0) == 0
    assert fn(None, b"debug_throw", 1) == _abi.SAR_ERR_OOM and b"out of host memory" in hooks.sar_last_error()
    assert fn(None, b"debug_throw", 2) == _abi.SAR_ERR_INVALID and b"thrown on request" in hooks.sar_last_error()
    assert fn(None, b"debug_throw", 3) == _abi.SAR_ERR_INVALID and b"unknown exception" in hooks.sar_last_error()


def test_checksum_is_the_oracles_fnv1a64_and_bench_extras_read_the_goldens(sar, oracle):
    """bench.py's `parity` rests on two things that need no GPU: sar_checksum_fnv1a64 (the product's) equals the oracle's FNV-1a,
    and tools/bench_extras.py finds the committed checksums of the full-size frames and compares field by field."""
    sys_path = os.path.join(ROOT, "tools")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import bench_extras as X
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 4096, 100_003):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        assert X.fnv1a64(sar, a) == f"{oracle.fnv1a64(a):016x}"
    g = X.golden_case("c4_full_1e10")
    assert g and g["jobs"] == 1048576 and len(g["count_fnv"]) == 16 and X.golden_case("no such case") is None
    # a frame that is not the golden's: "differs", and the differing fields are named
    z = np.zeros((4, 4), np.uint32)
    p = X.frame_parity(sar, "c4_full_1e10", z, z.astype(np.float32), z.astype(np.float64), np.zeros((4, 4, 4), np.uint16), 0)
    assert p["result"] == "differs" and "count_fnv" in p["differing_fields"] and p["against"].endswith("[c4_full_1e10]")
    assert X.frame_parity(sar, "nope", z, z, z, z, 0)["result"] == "no golden"
    buf = C.create_string_buffer(64)
    if sar.device_count() == 0:
        assert sar.load_library().sar_device_pci_bus_id(0, buf, 64) != 0


def test_batch_frames_answers_one_where_frames_cannot_share_launches():
    """sar_runtime_batch_frames makes the test sar_render_jobs_batch makes (host arithmetic; rt == NULL: a runtime yet to be made):
    frames beyond 4 Mpx (bins of 65 536 pixels), of several launch chunks or on the one-atomic-per-visit path render one after
    the other — the caller builds ONE runtime per lane for them, not a batch of runtimes."""
    import strange_attractor_renderer_amd as S
    c5 = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, jobs_total=65536)
    assert S.batch_frames(c5) == 16                                     # no launch has reported its survivors yet: two frames per XCD
    c2 = S.Config.poisson_saturne(iterations=1_000_000_000, width=2048, height=2048, jobs_total=131072)
    assert S.batch_frames(c2) in (8, 16)
    assert S.batch_frames(c2.replace(width=4096, height=4096)) == 1     # bins of 65 536 pixels: packed counters, no batched kernel
    assert S.batch_frames(c2.replace(width=9000, height=9000)) == 1     # 81 Mpx: one global atomic per visit
    assert S.batch_frames(c2.replace(jobs_total=16, iterations=16 * (1 << 33))) == 1   # jobs of several segments
    assert S.batch_frames(c2.replace(iterations=400_000_000_000, jobs_total=1 << 22)) == 1   # several launch chunks (the 24 GiB scratch cap)


def test_the_library_exports_no_c_named_variable(sar):
    """Two copies of the library live in one process in this test-suite (the product and the hooks build, loaded RTLD_GLOBAL): a
    variable with a C name and external linkage — what an unnamed namespace inside `extern "C"` produces — is shared between them,
    constructed and destroyed twice (a double free at exit once it owns memory; found that way in round 6). Every unmangled name
    the product defines is a `sar_` function or the HIP compiler's per-file id."""
    from strange_attractor_renderer_amd import _abi
    for path in (_abi.LIB_PATH, _abi.HOOKS_PATH):
        out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
        odd = []
        for line in out.splitlines():
            kind, name = line.split()[-2:]
            if name.startswith("_Z") or name.startswith("__hip_cuid_"):
                continue
            if not (name.startswith("sar_") and kind == "T"):
                odd.append(line)
        assert not odd, (path, odd)
    both = subprocess.run([sys.executable, "-c",
                           "import ctypes as C; from strange_attractor_renderer_amd import _abi; "
                           "a = C.CDLL(_abi.HOOKS_PATH, mode=C.RTLD_GLOBAL); b = C.CDLL(_abi.LIB_PATH, mode=C.RTLD_GLOBAL); "
                           "assert a.sar_abi_version() == b.sar_abi_version()"],
                          cwd=ROOT, capture_output=True, text=True)
    assert both.returncode == 0, both.stderr


def test_host_reserve_prepares_and_releases_blocks_without_a_device():
    """sar_host_reserve only maps and zeroes memory on a helper thread (no HIP call): it works, can be replaced and dropped —
    also while the helper is still at it — on a machine without a GPU."""
    import strange_attractor_renderer_amd as S
    S.host_reserve(8 << 20, 6)
    S.host_reserve(24 << 20, 3)          # replaces the first announcement (its blocks are unmapped)
    S.host_reserve(0, 0)
    S.host_reserve(1 << 20, 5)           # below 4 MiB: sar_host_alloc does not map such blocks, nothing to prepare
    S.host_reserve(0, 0)
    with pytest.raises(S.SarError):
        S.host_reserve(8 << 20, 5000)
    import threading                     # announcements from several threads follow each other (helpers counted right: no hang)

    def announce(k):
        for _ in range(10):
            S.host_reserve((8 + 2 * k) << 20, 3)
    threads = [threading.Thread(target=announce, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    S.host_reserve(0, 0)
