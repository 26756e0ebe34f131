"""ctypes mirror of include/sar.h (the C ABI of the HIP library).

Plumbing only: struct layouts, prototypes and the loader for ``libsar_hip.so``. The library is the
product; there is NO CPU fallback — loading fails loudly when the shared object is missing.
"""
from __future__ import annotations

import ctypes as C
import os

SAR_OK = 0
SAR_ERR_INVALID = 1
SAR_ERR_DIM_MISMATCH = 2
SAR_ERR_NO_DEVICE = 3
SAR_ERR_HIP = 4
SAR_ERR_OOM = 5
SAR_ERR_RANGE = 6
SAR_ERR_IO = 7

SAR_RENDER_GAS = 0
SAR_RENDER_DEPTH = 1
SAR_ATTRACTOR_SPROTT2 = 0
SAR_CT_POISSON_SATURNE = 0
SAR_CT_ADJUSTED_VELOCITY = 1
SAR_PALETTE_MAX = 15
SAR_FMT_RGBA16 = 0
SAR_FMT_RGB16 = 1
SAR_FMT_RGBA8 = 2
SAR_FMT_RGB8 = 3


class SarConfig(C.Structure):
    """``struct sar_config`` — POD mirror of the reference's Config/View/Colors (src/lib.rs:253-308)."""

    _fields_ = [
        ("iterations", C.c_uint64),
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("render_kind", C.c_int32),
        ("transparent", C.c_int32),
        ("angle", C.c_double),
        ("silent", C.c_int32),
        ("attractor_kind", C.c_int32),
        ("coeff_x", C.c_double * 10),
        ("coeff_y", C.c_double * 10),
        ("coeff_z", C.c_double * 10),
        ("palette_len", C.c_uint32),
        ("_pad0", C.c_uint32),
        ("palette_rgb", (C.c_double * 3) * SAR_PALETTE_MAX),
        ("brightness_offset", C.c_double),
        ("brightness_factor", C.c_double),
        ("center_camera", C.c_double * 3),
        ("rotation_axis", C.c_double * 3),
        ("rotation_angle", C.c_double),
        ("scale", C.c_double),
        ("color_transform", C.c_int32),
        ("_pad1", C.c_int32),
        ("ct_offset", C.c_double),
        ("ct_factor", C.c_double),
        ("seed", C.c_uint64),
        ("jobs_total", C.c_uint32),
        ("_pad2", C.c_uint32),
    ]


class SarParallelTiming(C.Structure):
    _fields_ = [
        ("total_ms", C.c_float),
        ("render_ms", C.c_float),
        ("exchange_ms", C.c_float),
        ("colorize_ms", C.c_float),
        ("n_devices", C.c_uint32),
        ("peer_access_failures", C.c_uint32),
        ("exchange_bytes_per_device", C.c_uint64),
        ("host_ms_before_exchange", C.c_float),
        ("host_ms_enqueue", C.c_float),
        ("draw_ahead_ms", C.c_float),
        ("_pad", C.c_float),
    ]


class SarExchangeLayout(C.Structure):
    _fields_ = [("world", C.c_uint32), ("rank", C.c_uint32), ("slice_pixels", C.c_uint32), ("first_px", C.c_uint32), ("n_px", C.c_uint32),
                ("granules", C.c_uint32), ("block_bytes", C.c_uint64)]


class SarTiming(C.Structure):
    _fields_ = [
        ("iterate_ms", C.c_float),
        ("resolve_ms", C.c_float),
        ("colorize_ms", C.c_float),
        ("merge_ms", C.c_float),
        ("iterate_launches", C.c_uint32),
        ("warmup_ms", C.c_float),
        ("iterations_counted", C.c_uint64),
        ("depth_atomics", C.c_uint64),
        ("depth_candidates", C.c_uint64),
    ]


_P = C.POINTER
_cfg_p = _P(SarConfig)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/sar.h declares
PROTOTYPES = {
    "sar_abi_version": (C.c_int, []),
    "sar_build_id": (C.c_char_p, []),
    "sar_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "sar_checksum_fnv1a64": (C.c_int, [_vp, C.c_size_t, _P(C.c_uint64)]),
    "sar_status_string": (C.c_char_p, [C.c_int]),
    "sar_last_error": (C.c_char_p, []),
    "sar_device_count": (C.c_int, [_P(C.c_int)]),
    "sar_config_poisson_saturne": (C.c_int, [_cfg_p]),
    "sar_config_solar_sail": (C.c_int, [_cfg_p]),
    "sar_config_validate": (C.c_int, [_cfg_p]),
    "sar_rotation_matrix": (C.c_int, [_cfg_p, _P(C.c_double)]),
    "sar_start_points": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint32, _P(C.c_double)]),
    "sar_runtime_new": (C.c_int, [_cfg_p, C.c_int, _P(_vp)]),
    "sar_runtime_new_group": (C.c_int, [_cfg_p, C.c_int, C.c_uint32, _P(_vp)]),
    "sar_runtime_reset_batch": (C.c_int, [C.c_uint32, _P(_vp)]),
    "sar_colorize_device_batch": (C.c_int, [C.c_uint32, _P(_cfg_p), _P(_vp), _P(_vp)]),
    "sar_runtime_free": (C.c_int, [_vp]),
    "sar_runtime_reset": (C.c_int, [_vp]),
    "sar_runtime_set_width_height": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "sar_runtime_seed": (C.c_int, [_vp, C.c_uint64]),
    "sar_runtime_merge": (C.c_int, [_vp, _vp]),
    "sar_runtime_synchronize": (C.c_int, [_vp]),
    "sar_runtime_dims": (C.c_int, [_vp, _P(C.c_uint32), _P(C.c_uint32)]),
    "sar_runtime_set_stream": (C.c_int, [_vp, _vp]),
    "sar_runtime_get_stream": (C.c_int, [_vp, _P(_vp)]),
    "sar_runtime_get_copy_stream": (C.c_int, [_vp, _P(_vp)]),
    "sar_runtime_set_copy_stream": (C.c_int, [_vp, _vp]),
    "sar_render": (C.c_int, [_cfg_p, _vp]),
    "sar_render_jobs": (C.c_int, [_cfg_p, _vp, _P(C.c_double)]),
    "sar_render_job_range": (C.c_int, [_cfg_p, _vp, C.c_uint32, C.c_uint64, _P(C.c_double)]),
    "sar_render_job_range_device": (C.c_int, [_cfg_p, _vp, C.c_uint32, C.c_uint64, _vp]),
    "sar_render_jobs_batch": (C.c_int, [C.c_uint32, _P(_cfg_p), _P(_vp), _P(_P(C.c_double))]),
    "sar_runtime_batch_frames": (C.c_int, [_cfg_p, _vp, _P(C.c_uint32)]),
    "sar_runtime_prefetch_device": (C.c_int, [_cfg_p, _vp, C.c_uint32, C.c_uint64, _vp]),
    "sar_runtime_describe_last_launch": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "sar_colorize": (C.c_int, [_cfg_p, _vp, _P(C.c_uint16)]),
    "sar_colorize_device": (C.c_int, [_cfg_p, _vp, _vp]),
    "sar_runtime_extent": (C.c_int, [_cfg_p, _vp, C.c_uint32, C.c_uint64, _P(C.c_double), _P(C.c_double)]),
    "sar_image_format": (C.c_int, [C.c_int, C.c_int]),
    "sar_image_bytes": (C.c_size_t, [C.c_int, C.c_uint32, C.c_uint32]),
    "sar_image_convert_device": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "sar_colorize_format": (C.c_int, [_cfg_p, _vp, C.c_int, _vp]),
    "sar_colorize_format_async": (C.c_int, [_cfg_p, _vp, C.c_int, _vp, _P(C.c_uint64)]),
    "sar_runtime_wait_image": (C.c_int, [_vp, C.c_uint64]),
    "sar_runtime_read_image_async": (C.c_int, [_vp, _vp, _P(C.c_uint64)]),
    "sar_runtime_image_done": (C.c_int, [_vp, C.c_uint64, _P(C.c_int)]),
    "sar_host_alloc": (C.c_int, [C.c_size_t, _P(C.c_void_p)]),
    "sar_host_free": (C.c_int, [_vp]),
    "sar_host_reserve": (C.c_int, [C.c_size_t, C.c_uint32]),
    "sar_write_png": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "sar_write_bmp": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "sar_write_pam": (C.c_int, [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "sar_runtime_count": (C.c_int, [_vp, _P(C.c_uint32)]),
    "sar_runtime_steps": (C.c_int, [_vp, _P(C.c_double)]),
    "sar_runtime_zbuf": (C.c_int, [_vp, _P(C.c_float)]),
    "sar_runtime_max": (C.c_int, [_vp, _P(C.c_uint32)]),
    "sar_runtime_load": (C.c_int, [_vp, _P(C.c_uint32), _P(C.c_double), _P(C.c_float), C.c_uint32]),
    "sar_exchange_slice_pixels": (C.c_int, [C.c_uint32, C.c_uint32, _P(C.c_uint32)]),
    "sar_exchange_new": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _P(_vp), _P(SarExchangeLayout)]),
    "sar_exchange_free": (C.c_int, [_vp]),
    "sar_exchange_flags": (C.c_int, [_vp, _vp]),
    "sar_exchange_pack": (C.c_int, [_vp, _vp, C.c_double, _vp, _P(C.c_uint64), _P(C.c_uint64), _P(C.c_int)]),
    "sar_exchange_merge": (C.c_int, [_vp, _vp, _vp]),
    "sar_exchange_finish": (C.c_int, [_vp, _vp]),
    "sar_exchange_rooted": (C.c_int, [_vp, C.c_uint32, _vp, _vp]),
    "sar_colorize_range_device": (C.c_int, [_cfg_p, _vp, C.c_uint32, C.c_uint32, _vp]),
    "sar_renderer_new": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, _P(_vp)]),
    "sar_renderer_new_multi": (C.c_int, [_P(C.c_int), C.c_uint32, C.c_uint32, C.c_uint64, _P(_vp)]),
    "sar_renderer_num_devices": (C.c_int, [_vp, _P(C.c_uint32)]),
    "sar_renderer_last_timing": (C.c_int, [_vp, _P(SarParallelTiming)]),
    "sar_renderer_set_exchange": (C.c_int, [_vp, C.c_uint32]),
    "sar_renderer_num_units": (C.c_int, [_vp, _P(C.c_uint32)]),
    "sar_renderer_shutdown": (C.c_int, [_vp]),
    "sar_render_parallel": (C.c_int, [_vp, _cfg_p, C.c_uint32, _P(C.c_uint16)]),
    "sar_renderer_runtime": (C.c_int, [_vp, _P(_vp)]),
    "sar_runtime_enable_timing": (C.c_int, [_vp, C.c_int]),
    "sar_runtime_last_timing": (C.c_int, [_vp, _P(SarTiming)]),
    "sar_runtime_set_option": (C.c_int, [_vp, C.c_char_p, C.c_uint64]),
    "sar_bin_geometry": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P(C.c_uint32)]),
}

# the hooks build only (include/sar_test_hooks.h): attached when the loaded library exports it
OPTIONAL_PROTOTYPES = {
    "sar_runtime_set_test_option": (C.c_int, [_vp, C.c_char_p, C.c_uint64]),
    "sar_runtime_debug_spans": (C.c_int, [_vp, C.c_uint32, _P(C.c_float), C.c_uint32, _P(C.c_uint32)]),
}
STABLE_OPTIONS = ("block_threads", "checkpoint_stride", "hint_bits", "split_waves", "timing_accumulate")

LIB_NAME = "libsar_hip.so"
_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, LIB_NAME)
HOOKS_PATH = os.path.join(os.path.dirname(_PKG_DIR), "tests", "hooks", "libsar_hip_hooks.so")   # product objects + the test hooks

_lib = None
_default_path = LIB_PATH


def use_hooks_build():
    """From now on load_library() loads the hooks build (the product's object files + sar_runtime_set_test_option): what the
    test-suite and the A/B tools do before their first call. The product is never replaced on disk."""
    global _default_path, _lib
    if not os.path.exists(HOOKS_PATH):
        raise SarLibraryMissing(f"{HOOKS_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    if _default_path != HOOKS_PATH:
        _default_path, _lib = HOOKS_PATH, None


class SarLibraryMissing(RuntimeError):
    pass


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen the in-tree HIP library and attach prototypes. Raises if it is missing — by design
    there is no fallback path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SAR_LIBRARY") or _default_path  # SAR_LIBRARY: A/B timing of another build of the same ABI
    if not os.path.exists(p):
        raise SarLibraryMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in OPTIONAL_PROTOTYPES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if p in (LIB_PATH, HOOKS_PATH):
        verify_library(lib)  # a variant named explicitly (path / SAR_LIBRARY) is the caller's business
    if path is None:
        _lib = lib
    return lib


class SarLibraryStale(RuntimeError):
    pass


def verify_library(lib, csrc: str | None = None):
    """The product library must have been built from the sources that lie next to it: its embedded id (sar_build_id) against
    build.source_id() of the tree. Raises SarLibraryStale on a mismatch; silent where there is no source tree (an installed copy)."""
    from . import build
    if not os.path.isdir(csrc or build.CSRC):
        return
    want = build.source_id(csrc, extra_flags=[])
    got = lib.sar_build_id().decode()
    if got != want:
        raise SarLibraryStale(f"{LIB_NAME} was built from other sources (its id {got}, the tree's {want}): run "
                              "`python -c 'import __graft_entry__ as g; g.build()'` — or name a variant through SAR_LIBRARY")
