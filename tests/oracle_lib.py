"""ctypes loader for the CPU oracle (oracle/libsar_oracle.so) — test infrastructure.

Also holds an independent Python statement of the two reference presets (data from
reference src/lib.rs:310-387, defaults :289-307, :397-404, :480-492) so that oracle tests do not
depend on the product library, and product presets can be cross-checked against them.
"""
from __future__ import annotations

import ctypes as C  # noqa
import os
import subprocess

import numpy as np

from strange_attractor_renderer_amd._abi import (
    SAR_CT_ADJUSTED_VELOCITY,
    SAR_CT_POISSON_SATURNE,
    SAR_RENDER_DEPTH,
    SAR_RENDER_GAS,
    SarConfig,
)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libsar_oracle.so")


class OracleRuntime(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max", C.c_uint32),
        ("_pad", C.c_uint32),
        ("count", C.POINTER(C.c_uint32)),
        ("steps", C.POINTER(C.c_double)),
        ("zbuf", C.POINTER(C.c_float)),
    ]


_lib = None


def build_oracle():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    cfgp = C.POINTER(SarConfig)
    dp = C.POINTER(C.c_double)
    rtp = C.POINTER(OracleRuntime)
    L.sar_oracle_next_point.argtypes = [cfgp, dp, dp]
    L.sar_oracle_next_point.restype = None
    L.sar_oracle_rotation_matrix.argtypes = [cfgp, dp]
    L.sar_oracle_rotation_matrix.restype = None
    L.sar_oracle_color_transform.argtypes = [cfgp, dp, dp]
    L.sar_oracle_color_transform.restype = C.c_double
    L.sar_oracle_runtime_new.argtypes = [C.c_uint32, C.c_uint32]
    L.sar_oracle_runtime_new.restype = rtp
    L.sar_oracle_runtime_free.argtypes = [rtp]
    L.sar_oracle_runtime_free.restype = None
    L.sar_oracle_runtime_reset.argtypes = [rtp]
    L.sar_oracle_runtime_reset.restype = None
    L.sar_oracle_runtime_merge.argtypes = [rtp, rtp]
    L.sar_oracle_runtime_merge.restype = C.c_int
    L.sar_oracle_render.argtypes = [cfgp, rtp, dp, C.c_uint64]
    L.sar_oracle_render.restype = None
    L.sar_oracle_render_jobs.argtypes = [cfgp, rtp, dp, C.c_uint32, C.c_uint64]
    L.sar_oracle_render_jobs.restype = None
    L.sar_oracle_render_jobs_mt.argtypes = [cfgp, rtp, dp, C.c_uint32, C.c_uint64, C.c_uint32]
    L.sar_oracle_render_jobs_mt.restype = C.c_int
    L.sar_oracle_iterate.argtypes = [cfgp, dp, C.c_uint64, dp]
    L.sar_oracle_iterate.restype = None
    L.sar_oracle_palette.argtypes = [cfgp, C.c_double, dp]
    L.sar_oracle_palette.restype = None
    L.sar_oracle_colorize.argtypes = [cfgp, rtp, C.POINTER(C.c_uint16)]
    L.sar_oracle_colorize.restype = None
    L.sar_oracle_extent.argtypes = [cfgp, dp, C.c_uint32, C.c_uint64, dp]
    L.sar_oracle_extent.restype = None
    L.sar_oracle_convert.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_uint16), C.c_void_p]
    L.sar_oracle_convert.restype = None
    L.sar_oracle_start_points.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, dp]
    L.sar_oracle_start_points.restype = None
    u64p = C.POINTER(C.c_uint64)
    L.sar_oracle_splitmix64.argtypes = [C.c_uint64, C.c_uint32, u64p]
    L.sar_oracle_splitmix64.restype = None
    L.sar_oracle_xoshiro256pp.argtypes = [u64p, C.c_uint32, u64p]
    L.sar_oracle_xoshiro256pp.restype = None
    L.sar_oracle_xoshiro256_jump.argtypes = [u64p]
    L.sar_oracle_xoshiro256_jump.restype = None
    L.sar_oracle_unit_f64.argtypes = [C.c_uint64]
    L.sar_oracle_unit_f64.restype = C.c_double
    L.sar_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
    L.sar_oracle_fnv1a64.restype = C.c_uint64
    L.sar_oracle_render_parallel.argtypes = [
        cfgp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint16), C.POINTER(C.c_uint64), rtp]
    L.sar_oracle_render_parallel.restype = C.c_double
    _lib = L
    return L


# ---- presets (data) -------------------------------------------------------------------------
_DEFAULT_PALETTE = [
    (1.0, 1.0, 0.5), (0.5, 1.0, 0.5), (1.0, 0.5, 0.5),
    (0.5, 1.0, 1.0), (0.5, 0.5, 1.0), (1.0, 0.5, 1.0),
]


def _base_config() -> SarConfig:
    c = SarConfig()
    c.iterations = 10_000_000
    c.width, c.height = 1920, 1080
    c.render_kind = SAR_RENDER_GAS
    c.transparent = 1
    c.angle = 0.0
    c.silent = 1
    c.attractor_kind = 0
    c.palette_len = len(_DEFAULT_PALETTE)
    for k, rgb in enumerate(_DEFAULT_PALETTE):
        for ch in range(3):
            c.palette_rgb[k][ch] = rgb[ch]
    c.brightness_offset = -0.15
    c.brightness_factor = 5.0 / 3.0
    c.seed = 0
    c.jobs_total = 1
    return c


def poisson_saturne() -> SarConfig:
    c = _base_config()
    xs = [0.021, 1.182, -1.183, 0.128, -1.12, -0.641, -1.152, -0.834, -0.97, 0.722]
    ys = [0.243038, -0.825, -1.2, -0.835443, -0.835443, -0.364557, 0.458, 0.622785, -0.394937, -1.032911]
    zs = [-0.455696, 0.673, 0.915, -0.258228, -0.495, -0.264, -0.432, -0.416, -0.877, -0.3]
    for k in range(10):
        c.coeff_x[k], c.coeff_y[k], c.coeff_z[k] = xs[k], ys[k], zs[k]
    c.center_camera[0], c.center_camera[1], c.center_camera[2] = -0.005, 0.262, -0.366 + 0.12
    c.rotation_axis[0] = 0.304289493528802
    c.rotation_axis[1] = 0.760492682863655
    c.rotation_axis[2] = 0.573636455813981
    c.rotation_angle = 1.78268191887446
    c.scale = 1.0
    c.color_transform = SAR_CT_POISSON_SATURNE
    c.ct_offset = 0.0
    c.ct_factor = 0.0
    return c


def solar_sail() -> SarConfig:
    c = _base_config()
    xs = [0.744304, -0.546835, 0.121519, -0.653165, 0.399, 0.379, 0.44, 1.014, -0.805063, 0.377]
    ys = [-0.683, 0.531646, -0.04557, -1.2, -0.546835, 0.091139, 0.744304, -0.273418, -0.349367, -0.531646]
    zs = [0.712, 0.744304, -0.577215, 0.966, 0.04557, 1.063291, 0.01519, -0.425316, 0.212658, -0.01519]
    for k in range(10):
        c.coeff_x[k], c.coeff_y[k], c.coeff_z[k] = xs[k], ys[k], zs[k]
    c.center_camera[0], c.center_camera[1], c.center_camera[2] = 0.28, -0.12, 0.22
    c.rotation_axis[0], c.rotation_axis[1], c.rotation_axis[2] = 0.02466, 0.4618, -0.54789
    c.rotation_angle = 2.2195
    c.scale = 1.7
    c.color_transform = SAR_CT_ADJUSTED_VELOCITY
    c.ct_offset = 0.8
    c.ct_factor = -0.2
    return c


def copy_config(c: SarConfig) -> SarConfig:
    out = SarConfig()
    C.memmove(C.byref(out), C.byref(c), C.sizeof(SarConfig))
    return out


# ---- numpy-friendly wrappers ------------------------------------------------------------------
class Runtime:
    """Owns a sar_oracle_runtime; exposes numpy views of its buffers."""

    def __init__(self, width: int, height: int):
        self._p = lib().sar_oracle_runtime_new(width, height)
        if not self._p:
            raise MemoryError
        self.width, self.height = width, height

    def __del__(self):
        try:
            if self._p:
                lib().sar_oracle_runtime_free(self._p)
                self._p = None
        except Exception:
            pass

    @property
    def ptr(self):
        return self._p

    @property
    def count(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._p.contents.count, shape=(self.height, self.width))

    @property
    def steps(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._p.contents.steps, shape=(self.height, self.width))

    @property
    def zbuf(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._p.contents.zbuf, shape=(self.height, self.width))

    @property
    def max(self) -> int:
        return int(self._p.contents.max)

    def set_max(self, v: int):
        self._p.contents.max = v

    def reset(self):
        lib().sar_oracle_runtime_reset(self._p)


def _dptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def start_points(seed: int, first_job: int, n_jobs: int) -> np.ndarray:
    out = np.empty((n_jobs, 3), dtype=np.float64)
    lib().sar_oracle_start_points(seed, first_job, n_jobs, _dptr(out))
    return out


# the start-point stream's pieces (tests/test_oracle_kat.py holds them to their published vectors)
def splitmix64(seed: int, n: int) -> list:
    out = (C.c_uint64 * n)()
    lib().sar_oracle_splitmix64(seed, n, out)
    return list(out)


def xoshiro256pp(state, n: int):
    """n outputs from `state` (four u64); returns (outputs, state afterwards)."""
    st = (C.c_uint64 * 4)(*state)
    out = (C.c_uint64 * n)()
    lib().sar_oracle_xoshiro256pp(st, n, out)
    return list(out), list(st)


def xoshiro256_jump(state) -> list:
    st = (C.c_uint64 * 4)(*state)
    lib().sar_oracle_xoshiro256_jump(st)
    return list(st)


def unit_f64(raw: int) -> float:
    return lib().sar_oracle_unit_f64(raw)


def render(cfg: SarConfig, rt: Runtime, p0, iterations: int):
    p = np.ascontiguousarray(p0, dtype=np.float64)
    lib().sar_oracle_render(C.byref(cfg), rt.ptr, _dptr(p), iterations)


def render_jobs(cfg: SarConfig, rt: Runtime, starts: np.ndarray, iters_per_job: int):
    s = np.ascontiguousarray(starts, dtype=np.float64)
    lib().sar_oracle_render_jobs(C.byref(cfg), rt.ptr, _dptr(s), s.shape[0], iters_per_job)


def host_threads(cap: int = 64) -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(cap, n))


def render_jobs_mt(cfg: SarConfig, rt: Runtime, starts: np.ndarray, iters_per_job: int, threads: int = 0):
    """render_jobs on `threads` host threads, identical bits (contiguous job slices merged in slice order)."""
    s = np.ascontiguousarray(starts, dtype=np.float64)
    rc = lib().sar_oracle_render_jobs_mt(C.byref(cfg), rt.ptr, _dptr(s), s.shape[0], iters_per_job,
                                         threads or host_threads())
    if rc != 0:
        raise MemoryError("sar_oracle_render_jobs_mt")


def iterate(cfg: SarConfig, p0, n: int) -> np.ndarray:
    p = np.ascontiguousarray(p0, dtype=np.float64)
    out = np.empty(3, dtype=np.float64)
    lib().sar_oracle_iterate(C.byref(cfg), _dptr(p), n, _dptr(out))
    return out


def rotation_matrix(cfg: SarConfig) -> np.ndarray:
    m = np.empty(9, dtype=np.float64)
    lib().sar_oracle_rotation_matrix(C.byref(cfg), _dptr(m))
    return m.reshape(3, 3)


def colorize(cfg: SarConfig, rt: Runtime) -> np.ndarray:
    out = np.empty((rt.height, rt.width, 4), dtype=np.uint16)
    lib().sar_oracle_colorize(C.byref(cfg), rt.ptr, out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def extent(cfg: SarConfig, starts: np.ndarray, iters_per_job: int) -> np.ndarray:
    """[xmin,xmax,ymin,ymax,zmin,zmax] of the screen-space points, then of the raw points (src/lib.rs:326-333)."""
    st = np.ascontiguousarray(starts, dtype=np.float64)
    out = np.zeros(12)
    lib().sar_oracle_extent(C.byref(cfg), _dptr(st), st.shape[0], iters_per_job, _dptr(out))
    return out


def convert(fmt: int, rgba16: np.ndarray) -> np.ndarray:
    """write_image_matches' format conversion (src/bin/main.rs:52-57) of an (H, W, 4) uint16 image."""
    h, w = rgba16.shape[:2]
    ch, dt = {0: (4, np.uint16), 1: (3, np.uint16), 2: (4, np.uint8), 3: (3, np.uint8)}[fmt]
    src = np.ascontiguousarray(rgba16, dtype=np.uint16)
    out = np.zeros((h, w, ch), dtype=dt)
    lib().sar_oracle_convert(fmt, h * w, src.ctypes.data_as(C.POINTER(C.c_uint16)), out.ctypes.data_as(C.c_void_p))
    return out


def merge(dst: Runtime, src: Runtime) -> int:
    return lib().sar_oracle_runtime_merge(dst.ptr, src.ptr)


def fnv1a64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(lib().sar_oracle_fnv1a64(a.ctypes.data_as(C.c_void_p), a.nbytes))


def render_parallel(cfg: SarConfig, threads: int, jobs_per_thread: int, seed: int,
                    want_image: bool = True, merged: Runtime | None = None):
    """CPU baseline shaped like the reference's render_parallel. Returns (seconds, iters, rgba)."""
    img = np.empty((cfg.height, cfg.width, 4), dtype=np.uint16) if want_image else None
    done = C.c_uint64(0)
    secs = lib().sar_oracle_render_parallel(
        C.byref(cfg), threads, jobs_per_thread, seed,
        img.ctypes.data_as(C.POINTER(C.c_uint16)) if want_image else None,
        C.byref(done), merged.ptr if merged is not None else None)
    return secs, int(done.value), img


__all__ = [
    "lib", "build_oracle", "poisson_saturne", "solar_sail", "copy_config", "Runtime", "start_points", "splitmix64", "xoshiro256pp", "xoshiro256_jump", "unit_f64",
    "render", "render_jobs", "render_jobs_mt", "host_threads", "iterate", "rotation_matrix", "colorize", "merge", "fnv1a64",
    "render_parallel", "SAR_RENDER_GAS", "SAR_RENDER_DEPTH",
]
