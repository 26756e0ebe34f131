"""Host-side mirror of the reference crate's surface for the iterate/accumulate path, over the C ABI.

Names and argument meaning follow the reference (src/lib.rs):

    Config.poisson_saturne() / Config.solar_sail()      config presets           (:310, :355)
    Runtime(config)  .reset()  .merge(other)            Runtime                  (:631-739)
    render(config, runtime)                             one trajectory           (:747-838)
    colorize(config, runtime) -> HxWx4 uint16           FinalImage               (:841-904)
    ParallelRenderer(...) / .shutdown()                 thread pool -> GPU lanes (:908-1031)
    render_parallel(renderer, config, jobs_per_thread)  job split + merge        (:1051-1082)

plus what the reference hides: a seed, the job count, explicit start points and read-back accessors.
Everything here is plumbing (ctypes + numpy); all arithmetic runs in libsar_hip.so on the GPU.
There is no CPU fallback: without the library or without a HIP device these calls raise.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import (SAR_CT_ADJUSTED_VELOCITY, SAR_CT_POISSON_SATURNE, SAR_RENDER_DEPTH,  # noqa: F401
                   SAR_RENDER_GAS, SarConfig, SarParallelTiming, SarTiming)


class SarError(RuntimeError):
    def __init__(self, status: int, where: str):
        lib = _abi.load_library()
        msg = lib.sar_last_error().decode(errors="replace")
        name = lib.sar_status_string(status).decode()
        super().__init__(f"{where}: {name} ({status}) {msg}")
        self.status = status


def _check(status: int, where: str):
    if status != _abi.SAR_OK:
        raise SarError(status, where)


def _lib():
    return _abi.load_library()


class Config:
    """``Config<PolynomialSprott2Degree, _>`` as a thin wrapper around ``struct sar_config``.

    Field names are the reference's; nested View/Colors fields are flattened
    (``scale``, ``center_camera``, ``brightness_offset`` ...)."""

    def __init__(self, c: SarConfig | None = None):
        self.c = c if c is not None else SarConfig()

    @classmethod
    def poisson_saturne(cls, **overrides) -> "Config":
        cfg = cls()
        _check(_lib().sar_config_poisson_saturne(C.byref(cfg.c)), "sar_config_poisson_saturne")
        return cfg.replace(**overrides)

    @classmethod
    def solar_sail(cls, **overrides) -> "Config":
        cfg = cls()
        _check(_lib().sar_config_solar_sail(C.byref(cfg.c)), "sar_config_solar_sail")
        return cfg.replace(**overrides)

    def copy(self) -> "Config":
        out = SarConfig()
        C.memmove(C.byref(out), C.byref(self.c), C.sizeof(SarConfig))
        return Config(out)

    def replace(self, **kw) -> "Config":
        """``Config { iterations: ..., ..preset }`` update syntax (src/lib.rs:9-15)."""
        out = self.copy()
        for k, v in kw.items():
            if k == "render":
                k = "render_kind"
            if not hasattr(out.c, k):
                raise AttributeError(f"sar_config has no field {k!r}")
            cur = getattr(out.c, k)
            if isinstance(cur, C.Array):
                arr = np.asarray(v, dtype=np.float64)
                if k == "palette_rgb":
                    out.c.palette_len = arr.shape[0]
                    for i in range(arr.shape[0]):
                        for ch in range(3):
                            out.c.palette_rgb[i][ch] = float(arr[i, ch])
                else:
                    for i in range(len(cur)):
                        cur[i] = float(arr[i])
            else:
                setattr(out.c, k, v)
        return out

    def validate(self):
        _check(_lib().sar_config_validate(C.byref(self.c)), "sar_config_validate")

    def __getattr__(self, name):
        c = object.__getattribute__(self, "c")
        if hasattr(c, name):
            v = getattr(c, name)
            return np.ctypeslib.as_array(v).copy() if isinstance(v, C.Array) else v
        raise AttributeError(name)

    def rotation_matrix(self) -> np.ndarray:
        m = np.empty(9, dtype=np.float64)
        _check(_lib().sar_rotation_matrix(C.byref(self.c), m.ctypes.data_as(C.POINTER(C.c_double))),
               "sar_rotation_matrix")
        return m.reshape(3, 3)


def start_points(seed: int, first_job: int, n_jobs: int) -> np.ndarray:
    """The start-point stream the reference leaves to OS entropy: (n_jobs, 3) f64, already * 0.1."""
    out = np.empty((n_jobs, 3), dtype=np.float64)
    _check(_lib().sar_start_points(seed, first_job, n_jobs, out.ctypes.data_as(C.POINTER(C.c_double))),
           "sar_start_points")
    return out


def device_count() -> int:
    n = C.c_int(0)
    st = _lib().sar_device_count(C.byref(n))
    return int(n.value) if st == _abi.SAR_OK else 0


@dataclass
class Timing:
    iterate_ms: float
    resolve_ms: float
    colorize_ms: float
    merge_ms: float
    iterate_launches: int
    iterations_counted: int
    depth_atomics: int = 0
    warmup_ms: float = 0.0
    depth_candidates: int = 0


class Runtime:
    """``Runtime`` (src/lib.rs:631-646) living on one GPU."""

    def __init__(self, config: Config, device: int = 0, _borrowed=None):
        self._own = _borrowed is None
        if _borrowed is not None:
            self._h = _borrowed
        else:
            h = C.c_void_p()
            _check(_lib().sar_runtime_new(C.byref(config.c), device, C.byref(h)), "sar_runtime_new")
            self._h = h
            # A/B and whole-suite test hook of THIS harness (the library itself reads no environment): SAR_SPLIT=1|2 forces
            # the whole or the split iterate kernel on every runtime created through it
            for env, opt in (("SAR_SPLIT", "split_waves"),):
                if os.environ.get(env, "") in ("1", "2"):
                    _check(_lib().sar_runtime_set_option(self._h, opt.encode(), int(os.environ[env])), "sar_runtime_set_option")
        self.device = device

    @classmethod
    def group(cls, config: Config, n: int, device: int = 0) -> list:
        """n runtimes for the frames of one batch (sar_runtime_new_group): one stream, one read-back stream, their buffers carved
        from one device and one page-locked allocation. Each is an ordinary Runtime; close them in any order."""
        handles = (C.c_void_p * n)()
        _check(_lib().sar_runtime_new_group(C.byref(config.c), device, n, handles), "sar_runtime_new_group")
        out = []
        for h in handles:
            rt = cls(config, device, _borrowed=C.c_void_p(h))
            rt._own = True
            for env, opt in (("SAR_SPLIT", "split_waves"),):
                if os.environ.get(env, "") in ("1", "2"):
                    _check(_lib().sar_runtime_set_option(rt._h, opt.encode(), int(os.environ[env])), "sar_runtime_set_option")
            out.append(rt)
        return out

    def close(self):
        for ref in getattr(self, "_exchanges", []):       # contexts that borrow this runtime go first
            ex = ref()
            if ex is not None:
                ex.close()
        self._exchanges = []
        if getattr(self, "_h", None) and self._own:
            _lib().sar_runtime_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def dims(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(_lib().sar_runtime_dims(self._h, C.byref(w), C.byref(h)), "sar_runtime_dims")
        return int(w.value), int(h.value)

    def reset(self):
        _check(_lib().sar_runtime_reset(self._h), "sar_runtime_reset")

    def set_width_height(self, width: int, height: int):
        _check(_lib().sar_runtime_set_width_height(self._h, width, height), "sar_runtime_set_width_height")

    def seed(self, seed: int):
        _check(_lib().sar_runtime_seed(self._h, seed), "sar_runtime_seed")

    def merge(self, other: "Runtime"):
        _check(_lib().sar_runtime_merge(self._h, other._h), "sar_runtime_merge")

    def synchronize(self):
        _check(_lib().sar_runtime_synchronize(self._h), "sar_runtime_synchronize")

    # ---- read-back ------------------------------------------------------------------------------
    def count(self) -> np.ndarray:
        w, h = self.dims()
        out = np.empty((h, w), dtype=np.uint32)
        _check(_lib().sar_runtime_count(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32))), "sar_runtime_count")
        return out

    def steps(self) -> np.ndarray:
        w, h = self.dims()
        out = np.empty((h, w), dtype=np.float64)
        _check(_lib().sar_runtime_steps(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "sar_runtime_steps")
        return out

    def zbuf(self) -> np.ndarray:
        w, h = self.dims()
        out = np.empty((h, w), dtype=np.float32)
        _check(_lib().sar_runtime_zbuf(self._h, out.ctypes.data_as(C.POINTER(C.c_float))), "sar_runtime_zbuf")
        return out

    def max(self) -> int:
        m = C.c_uint32()
        _check(_lib().sar_runtime_max(self._h, C.byref(m)), "sar_runtime_max")
        return int(m.value)

    def load(self, count: np.ndarray, steps: np.ndarray, zbuf: np.ndarray, max_: int):
        count = np.ascontiguousarray(count, dtype=np.uint32)
        steps = np.ascontiguousarray(steps, dtype=np.float64)
        zbuf = np.ascontiguousarray(zbuf, dtype=np.float32)
        _check(_lib().sar_runtime_load(self._h, count.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       steps.ctypes.data_as(C.POINTER(C.c_double)),
                                       zbuf.ctypes.data_as(C.POINTER(C.c_float)), int(max_)), "sar_runtime_load")

    # ---- measurement / tuning -----------------------------------------------------------------------
    def enable_timing(self, on: bool = True):
        _check(_lib().sar_runtime_enable_timing(self._h, 1 if on else 0), "sar_runtime_enable_timing")

    def last_timing(self) -> Timing:
        t = SarTiming()
        _check(_lib().sar_runtime_last_timing(self._h, C.byref(t)), "sar_runtime_last_timing")
        return Timing(t.iterate_ms, t.resolve_ms, t.colorize_ms, t.merge_ms, t.iterate_launches,
                      t.iterations_counted, t.depth_atomics, t.warmup_ms, t.depth_candidates)

    def debug_spans(self, which: int = 0) -> list:
        """(hooks build) the individual HIP-event spans held since they were last read, in launch order: 0 iterate, 1 accumulate +
        fold, 2 warm-up (include/sar_test_hooks.h)."""
        hook = getattr(_lib(), "sar_runtime_debug_spans", None)
        if hook is None:
            raise RuntimeError("debug_spans is a test hook: load the hooks build (_abi.use_hooks_build())")
        n = C.c_uint32(0)
        _check(hook(self._h, which, None, 0, C.byref(n)), "sar_runtime_debug_spans")
        buf = (C.c_float * max(n.value, 1))()
        _check(hook(self._h, which, buf, n.value, C.byref(n)), "sar_runtime_debug_spans")
        return [float(buf[k]) for k in range(n.value)]

    def set_option(self, name: str, value: int):
        """A stable option (include/sar.h) — or, on the hooks build the test-suite loads, an A/B / test option
        (include/sar_test_hooks.h); the product library has no such entry point."""
        if name in _abi.STABLE_OPTIONS:
            _check(_lib().sar_runtime_set_option(self._h, name.encode(), int(value)), f"sar_runtime_set_option({name})")
            return
        hook = getattr(_lib(), "sar_runtime_set_test_option", None)
        if hook is None:
            raise RuntimeError(f"option {name!r} is a test hook: load the hooks build (strange_attractor_renderer_amd._abi.use_hooks_build(), "
                               "or SAR_LIBRARY=tests/hooks/libsar_hip_hooks.so)")
        _check(hook(self._h, name.encode(), int(value)), f"sar_runtime_set_test_option({name})")

    def set_tuning(self, block_threads: int = 0, checkpoint_stride: int = 0, variant: int = 0, **more):
        """Convenience over set_option. variant: bits 0-3 path (0 automatic, 1 one atomic per visit, 3 binned), bits 8+
        debug_chunk_jobs."""
        if (variant >> 4) & 0xF:
            raise ValueError(f"variant {variant:#x}: bits 4-7 (the removed measurement modes) must be zero")
        self.set_option("block_threads", block_threads)
        self.set_option("checkpoint_stride", checkpoint_stride)
        hooks = hasattr(_lib(), "sar_runtime_set_test_option")
        if hooks or variant & 0xF:
            self.set_option("path", variant & 0xF)          # (test hooks: only the hooks build has them; 0 is the product's behaviour)
        if hooks or variant >> 8:
            self.set_option("debug_chunk_jobs", variant >> 8)
        for k, v in more.items():
            self.set_option(k, v)

    def describe_last_launch(self) -> str:
        """Which kernels the last render call really launched (sar_runtime_describe_last_launch)."""
        buf = C.create_string_buffer(512)
        _check(_lib().sar_runtime_describe_last_launch(self._h, buf, 512), "sar_runtime_describe_last_launch")
        return buf.value.decode()

    def stream(self) -> int:
        s = C.c_void_p()
        _check(_lib().sar_runtime_get_stream(self._h, C.byref(s)), "sar_runtime_get_stream")
        return int(s.value or 0)

    def set_stream(self, stream_ptr: int):
        _check(_lib().sar_runtime_set_stream(self._h, C.c_void_p(stream_ptr)), "sar_runtime_set_stream")

    def copy_stream(self) -> int:
        s = C.c_void_p()
        _check(_lib().sar_runtime_get_copy_stream(self._h, C.byref(s)), "sar_runtime_get_copy_stream")
        return int(s.value or 0)

    def set_copy_stream(self, stream_ptr: int):
        _check(_lib().sar_runtime_set_copy_stream(self._h, C.c_void_p(stream_ptr)), "sar_runtime_set_copy_stream")

    def share_streams(self, leader: "Runtime"):
        """This runtime enqueues where `leader` does (launch stream and read-back stream): the frames of one batch."""
        self.set_stream(leader.stream())
        self.set_copy_stream(leader.copy_stream())


class Exchange:
    """The ONE exchange step before colorize of the one-process-per-GPU path (sar_exchange_*, include/sar.h): Runtime::merge
    (src/lib.rs:708-738) folded in rank order over image slices. The collectives are the caller's (distributed.py), on buffers the
    caller owns; all arguments are device pointers."""

    RECORD = 64 * 16   # bytes of one record of the sparse form (SAR_EXCHANGE_GRANULE pixels x 16 B)

    def __init__(self, runtime: Runtime, world: int, rank: int):
        h, lay = C.c_void_p(), _abi.SarExchangeLayout()
        _check(_lib().sar_exchange_new(runtime.handle, world, rank, C.byref(h), C.byref(lay)), "sar_exchange_new")
        self._h, self.runtime = h, runtime      # (the context borrows the runtime: Runtime.close closes its exchanges first)
        runtime._exchanges = getattr(runtime, "_exchanges", [])
        runtime._exchanges.append(weakref.ref(self))
        self.world, self.rank = world, rank
        self.slice_pixels, self.first, self.count = lay.slice_pixels, lay.first_px, lay.n_px
        self.granules, self.block_bytes = lay.granules, lay.block_bytes

    def close(self):
        if getattr(self, "_h", None):
            _lib().sar_exchange_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flags(self, flags_dev_ptr: int):
        _check(_lib().sar_exchange_flags(self._h, C.c_void_p(flags_dev_ptr)), "sar_exchange_flags")

    def pack(self, flags_all_dev_ptr: int | None, dense_above: float, send_dev_ptr: int):
        """Plans (from the gathered flags; None: dense) and packs. Returns (sparse, send_bytes[world], recv_bytes[world]) — the
        one host wait of a frame's exchange."""
        sb, rb, sp = (C.c_uint64 * self.world)(), (C.c_uint64 * self.world)(), C.c_int(0)
        _check(_lib().sar_exchange_pack(self._h, C.c_void_p(flags_all_dev_ptr) if flags_all_dev_ptr else None, float(dense_above),
                                        C.c_void_p(send_dev_ptr), sb, rb, C.byref(sp)), "sar_exchange_pack")
        return bool(sp.value), [int(v) for v in sb], [int(v) for v in rb]

    def merge(self, recv_dev_ptr: int, scalars_dev_ptr: int):
        _check(_lib().sar_exchange_merge(self._h, C.c_void_p(recv_dev_ptr), C.c_void_p(scalars_dev_ptr)), "sar_exchange_merge")

    def finish(self, scalars_dev_ptr: int):
        _check(_lib().sar_exchange_finish(self._h, C.c_void_p(scalars_dev_ptr)), "sar_exchange_finish")

    def rooted(self, step: int, key_i64_dev_ptr: int, sum_i32_dev_ptr: int = 0):
        _check(_lib().sar_exchange_rooted(self._h, step, C.c_void_p(key_i64_dev_ptr), C.c_void_p(sum_i32_dev_ptr) if sum_i32_dev_ptr else None),
               "sar_exchange_rooted")


def exchange_slice_pixels(npix: int, world: int) -> int:
    """Pixels per owned slice of the sliced multi-GPU exchange (the last slice may be shorter)."""
    s = C.c_uint32()
    _check(_lib().sar_exchange_slice_pixels(npix, world, C.byref(s)), "sar_exchange_slice_pixels")
    return int(s.value)


def bin_geometry(width: int, height: int, bin_shift: int = 0, bin_interleave: int = 0) -> dict:
    """The pixel -> (bin, record) map of the LDS-binned path for this image shape (host arithmetic, no device needed)."""
    out = (C.c_uint32 * 8)()
    _check(_lib().sar_bin_geometry(width, height, bin_shift, bin_interleave, out), "sar_bin_geometry")
    keys = ("ok", "bins", "bin_shift", "interleaved", "seg_shift", "bin_bits", "hi_shift", "low_mask")
    return dict(zip(keys, (int(v) for v in out)))


def colorize_range_device(config: Config, runtime: Runtime, first_px: int, n_px: int, rgba_dev_ptr: int):
    """colorize of a pixel range with the max / depth range the runtime's scalars hold (n_px*8 bytes out); stream-ordered."""
    _check(_lib().sar_colorize_range_device(C.byref(config.c), runtime.handle, first_px, n_px, C.c_void_p(rgba_dev_ptr)),
           "sar_colorize_range_device")


def _starts_ptr(starts, n_jobs: int):
    if starts is None:
        return None, None
    s = np.ascontiguousarray(starts, dtype=np.float64)
    if s.shape != (n_jobs, 3):
        raise ValueError(f"starts must have shape ({n_jobs}, 3), got {s.shape}")
    return s, s.ctypes.data_as(C.POINTER(C.c_double))


def render(config: Config, runtime: Runtime):
    """``render(&config, &mut runtime)``: ONE trajectory of config.iterations (src/lib.rs:747)."""
    _check(_lib().sar_render(C.byref(config.c), runtime.handle), "sar_render")


def render_jobs(config: Config, runtime: Runtime, starts=None):
    """config.jobs_total trajectories of config.iterations // jobs_total each, with the sequential
    (job-major) semantics of calling ``render`` that many times on one un-reset runtime."""
    keep, ptr = _starts_ptr(starts, config.c.jobs_total)
    _check(_lib().sar_render_jobs(C.byref(config.c), runtime.handle, ptr), "sar_render_jobs")
    del keep


def render_jobs_batch(configs, runtimes, starts=None):
    """F frames of a sweep through ONE set of launches (sar_render_jobs_batch): frame i is ``render_jobs(configs[i],
    runtimes[i], starts[i])``, bit for bit — the frames share the chip instead of following each other."""
    n = len(configs)
    if len(runtimes) != n or (starts is not None and len(starts) != n):
        raise ValueError("configs, runtimes and starts must have the same length")
    cfgs = (C.POINTER(SarConfig) * n)(*[C.pointer(c.c) for c in configs])
    rts = (C.c_void_p * n)(*[r.handle for r in runtimes])
    keep, ptrs = [], None
    if starts is not None:
        ptrs = (C.POINTER(C.c_double) * n)()
        for i, (c, s) in enumerate(zip(configs, starts)):
            k, p = _starts_ptr(s, c.c.jobs_total)
            keep.append(k)
            if p is not None:
                ptrs[i] = p
    _check(_lib().sar_render_jobs_batch(n, cfgs, rts, ptrs), "sar_render_jobs_batch")
    del keep


def batch_frames(config: Config, runtime: "Runtime | None" = None) -> int:
    """How many frames like `config` fill the chip (sar_runtime_batch_frames): the length to call render_jobs_batch with; 1 = frames
    of this shape do not share launches. runtime None: the answer for a runtime yet to be made."""
    n = C.c_uint32()
    _check(_lib().sar_runtime_batch_frames(C.byref(config.c), runtime.handle if runtime is not None else None, C.byref(n)), "sar_runtime_batch_frames")
    return int(n.value)


def render_job_range(config: Config, runtime: Runtime, iters_per_job: int, starts):
    """A shard: the given jobs (explicit start points) with iters_per_job iterations each."""
    s = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
    _check(_lib().sar_render_job_range(C.byref(config.c), runtime.handle, s.shape[0], iters_per_job,
                                       s.ctypes.data_as(C.POINTER(C.c_double))), "sar_render_job_range")


def render_job_range_device(config: Config, runtime: Runtime, n_jobs: int, iters_per_job: int, starts_dev_ptr: int):
    """The same shard with its start points already in device memory (n_jobs*3 float64, [job][xyz]); stream-ordered."""
    _check(_lib().sar_render_job_range_device(C.byref(config.c), runtime.handle, n_jobs, iters_per_job,
                                              C.c_void_p(starts_dev_ptr)), "sar_render_job_range_device")


def prefetch_device(config: Config, runtime: Runtime, n_jobs: int, iters_per_job: int, starts_dev_ptr: int):
    """Announces the next render_job_range_device call (same arguments): its warm-up may run ahead, under the tail of
    the frame in flight. Results do not depend on it."""
    _check(_lib().sar_runtime_prefetch_device(C.byref(config.c), runtime.handle, n_jobs, iters_per_job,
                                              C.c_void_p(starts_dev_ptr)), "sar_runtime_prefetch_device")


def colorize(config: Config, runtime: Runtime) -> np.ndarray:
    """``colorize(&config, &runtime) -> FinalImage`` as an (H, W, 4) uint16 array (src/lib.rs:841)."""
    out = np.empty((config.c.height, config.c.width, 4), dtype=np.uint16)
    _check(_lib().sar_colorize(C.byref(config.c), runtime.handle, out.ctypes.data_as(C.POINTER(C.c_uint16))),
           "sar_colorize")
    return out


def reset_batch(runtimes):
    """Runtime::reset for every runtime of the list, in ONE launch where they share a stream (sar_runtime_reset_batch)."""
    n = len(runtimes)
    handles = (C.c_void_p * n)(*[rt.handle for rt in runtimes])
    _check(_lib().sar_runtime_reset_batch(n, handles), "sar_runtime_reset_batch")


def colorize_device_batch(configs, runtimes, rgba_dev_ptrs):
    """colorize of frame i = (configs[i], runtimes[i]) into rgba_dev_ptrs[i] (device memory), ONE launch for Gas frames of one
    palette on one stream (sar_colorize_device_batch)."""
    n = len(runtimes)
    cfgs = (C.POINTER(_abi.SarConfig) * n)(*[C.pointer(c.c) for c in configs])
    handles = (C.c_void_p * n)(*[rt.handle for rt in runtimes])
    outs = (C.c_void_p * n)(*[C.c_void_p(p) for p in rgba_dev_ptrs])
    _check(_lib().sar_colorize_device_batch(n, cfgs, handles, outs), "sar_colorize_device_batch")


def colorize_device(config: Config, runtime: Runtime, rgba_dev_ptr: int):
    """colorize into a caller-provided device buffer (H*W*8 bytes); stream-ordered, no host sync."""
    _check(_lib().sar_colorize_device(C.byref(config.c), runtime.handle, C.c_void_p(rgba_dev_ptr)),
           "sar_colorize_device")


def attractor_extent(config: Config, runtime: Runtime, n_jobs: int, iters_per_job: int, starts=None) -> np.ndarray:
    """The first pass the reference leaves as a TODO (src/lib.rs:326-333): ``[xmin, xmax, ymin, ymax, zmin, zmax]`` of
    the screen-space points followed by the same six numbers for the raw points."""
    out = np.zeros(12)
    sp = None
    if starts is not None:
        st = np.ascontiguousarray(starts, dtype=np.float64)
        if st.shape != (n_jobs, 3):
            raise ValueError("starts must be (n_jobs, 3)")
        sp = st.ctypes.data_as(C.POINTER(C.c_double))
    _check(_lib().sar_runtime_extent(C.byref(config.c), runtime.handle, n_jobs, iters_per_job, sp,
                                     out.ctypes.data_as(C.POINTER(C.c_double))), "sar_runtime_extent")
    return out


# ---- image export (src/bin/main.rs:40-100) -------------------------------------------------------------------
_FMT_SHAPE = {_abi.SAR_FMT_RGBA16: (4, np.uint16), _abi.SAR_FMT_RGB16: (3, np.uint16),
              _abi.SAR_FMT_RGBA8: (4, np.uint8), _abi.SAR_FMT_RGB8: (3, np.uint8)}


def image_format(transparent: bool, eight_bit: bool) -> int:
    """The format ``write_image_matches`` converts to for (--transparent, --8bit) (src/bin/main.rs:52-57)."""
    return int(_lib().sar_image_format(int(bool(transparent)), int(bool(eight_bit))))


def colorize_format(config: Config, runtime: Runtime, fmt: int) -> np.ndarray:
    """``colorize`` followed by the CLI's format conversion, both on the device; one copy of the converted image."""
    if fmt not in _FMT_SHAPE:
        raise ValueError(f"unknown image format {fmt}")
    ch, dt = _FMT_SHAPE[fmt]
    out = np.empty((config.c.height, config.c.width, ch), dtype=dt)
    _check(_lib().sar_colorize_format(C.byref(config.c), runtime.handle, fmt, out.ctypes.data_as(C.c_void_p)),
           "sar_colorize_format")
    return out


def host_reserve(nbytes: int, count: int):
    """Announces `count` page-locked blocks of `nbytes` to come (sar_host_reserve): helper threads map and zero them ahead of the
    HostImage()s that take them. (0, 0) releases what is left."""
    _check(_lib().sar_host_reserve(int(nbytes), int(count)), "sar_host_reserve")


def image_bytes(fmt: int, width: int, height: int) -> int:
    return int(_lib().sar_image_bytes(fmt, width, height))


class HostImage:
    """A page-locked host image of (height, width, channels-of-fmt) for ``colorize_format_async``; `array` is a numpy
    view of it, valid until ``close``."""

    def __init__(self, width: int, height: int, fmt: int):
        if fmt not in _FMT_SHAPE:
            raise ValueError(f"unknown image format {fmt}")
        ch, dt = _FMT_SHAPE[fmt]
        self.fmt = fmt
        nbytes = int(_lib().sar_image_bytes(fmt, width, height))
        p = C.c_void_p()
        _check(_lib().sar_host_alloc(nbytes, C.byref(p)), "sar_host_alloc")
        self.ptr = p.value
        self.array = np.frombuffer((C.c_ubyte * nbytes).from_address(self.ptr), dtype=dt).reshape(height, width, ch)

    def close(self):
        if self.ptr:
            self.array = None
            _check(_lib().sar_host_free(C.c_void_p(self.ptr)), "sar_host_free")
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def colorize_format_async(config: Config, runtime: Runtime, image: HostImage) -> int:
    """``colorize_format`` into a page-locked image, enqueued only: returns the ticket ``wait_image`` takes. The runtime
    may be reset and rendered into again meanwhile (the `sequence` sweep reads frame k back under frame k+1)."""
    t = C.c_uint64()
    _check(_lib().sar_colorize_format_async(C.byref(config.c), runtime.handle, image.fmt, C.c_void_p(image.ptr), C.byref(t)),
           "sar_colorize_format_async")
    return int(t.value)


def colorize_format_device(config: Config, runtime: Runtime, fmt: int):
    """colorize + the CLI's conversion, the image left in the runtime's device memory (sar_colorize_format_async with no host
    image): `read_image_async` fetches it later. Enqueues only."""
    _check(_lib().sar_colorize_format_async(C.byref(config.c), runtime.handle, fmt, None, None), "sar_colorize_format_async")


def read_image_async(runtime: Runtime, image: "HostImage") -> int:
    """The read-back of the image `colorize_format_device` left, into a page-locked host image; returns the ticket."""
    t = C.c_uint64()
    _check(_lib().sar_runtime_read_image_async(runtime.handle, C.c_void_p(image.ptr), C.byref(t)), "sar_runtime_read_image_async")
    return int(t.value)


def image_done(runtime: Runtime, ticket: int) -> bool:
    """Whether the read-back behind `ticket` has completed (no wait)."""
    d = C.c_int(0)
    _check(_lib().sar_runtime_image_done(runtime.handle, ticket, C.byref(d)), "sar_runtime_image_done")
    return bool(d.value)


def wait_image(runtime: Runtime, ticket: int):
    _check(_lib().sar_runtime_wait_image(runtime.handle, ticket), "sar_runtime_wait_image")


def convert_device(runtime: Runtime, rgba16_dev_ptr: int, fmt: int, out_dev_ptr: int):
    """RGBA16 -> fmt between two device buffers, ordered on the runtime's stream."""
    _check(_lib().sar_image_convert_device(runtime.handle, C.c_void_p(rgba16_dev_ptr), fmt, C.c_void_p(out_dev_ptr)),
           "sar_image_convert_device")


def _fmt_of(image: np.ndarray) -> int:
    for fmt, (ch, dt) in _FMT_SHAPE.items():
        if image.ndim == 3 and image.shape[2] == ch and image.dtype == dt:
            return fmt
    raise ValueError("image must be (H, W, 3|4) of uint8 or uint16")


def write_image(image: np.ndarray, path: str, kind: str = "png"):
    """Encodes a host image (as returned by ``colorize_format``) as PNG, BMP or PAM."""
    fmt = _fmt_of(image)
    img = np.ascontiguousarray(image)
    fn = {"png": _lib().sar_write_png, "bmp": _lib().sar_write_bmp, "pam": _lib().sar_write_pam}[kind]
    _check(fn(os.fsencode(path), fmt, img.shape[1], img.shape[0], img.ctypes.data_as(C.c_void_p)), f"sar_write_{kind}")


def write_image_matches(config: Config, runtime: Runtime, name: str, transparent: bool = False, eight_bit: bool = False,
                        pam: bool = False, bmp: bool = False) -> str:
    """``write_image_matches`` (src/bin/main.rs:40-100): convert by (transparent, 8bit), pick the encoder by
    (pam, bmp) — both need 8bit (:256-258) — and replace the extension of ``name``. Returns the path written."""
    if (pam or bmp) and not eight_bit:
        raise ValueError("--pam / --bmp require --8bit (src/bin/main.rs:256-258)")
    kind = "pam" if pam else ("bmp" if bmp else "png")
    path = os.path.splitext(name)[0] + "." + kind
    write_image(colorize_format(config, runtime, image_format(transparent, eight_bit)), path, kind)
    return path


class ParallelRenderer:
    """``ParallelRenderer`` (src/lib.rs:908): `units` stands in for the thread count the job split
    divides by (0 = one trajectory per SIMD lane of the device)."""

    def __init__(self, device: int = 0, units: int = 0, seed: int = 0, devices=None):
        """devices: a list of HIP device ordinals -> one renderer over several GPUs (sar_renderer_new_multi): the jobs
        are sharded over them, the partial buffers merged point-to-point over xGMI, all behind the C ABI."""
        h = C.c_void_p()
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            _check(_lib().sar_renderer_new_multi(devs, len(devices), units, seed, C.byref(h)), "sar_renderer_new_multi")
            device = int(devices[0])
        else:
            _check(_lib().sar_renderer_new(device, units, seed, C.byref(h)), "sar_renderer_new")
        self._h = h
        self.device = device

    def num_devices(self) -> int:
        n = C.c_uint32()
        _check(_lib().sar_renderer_num_devices(self._h, C.byref(n)), "sar_renderer_num_devices")
        return int(n.value)

    def last_timing(self) -> dict:
        t = SarParallelTiming()
        _check(_lib().sar_renderer_last_timing(self._h, C.byref(t)), "sar_renderer_last_timing")
        return {k: getattr(t, k) for k, _ in SarParallelTiming._fields_}

    def set_exchange(self, mode: int):
        """0 automatic, 1 dense (whole slices by peer copies), 2 sparse (kernels push the touched segments' records)."""
        _check(_lib().sar_renderer_set_exchange(self._h, int(mode)), "sar_renderer_set_exchange")

    def num_threads(self) -> int:
        n = C.c_uint32()
        _check(_lib().sar_renderer_num_units(self._h, C.byref(n)), "sar_renderer_num_units")
        return int(n.value)

    def runtime(self) -> Runtime:
        h = C.c_void_p()
        _check(_lib().sar_renderer_runtime(self._h, C.byref(h)), "sar_renderer_runtime")
        return Runtime(None, self.device, _borrowed=h)

    def shutdown(self):
        if getattr(self, "_h", None):
            _lib().sar_renderer_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass


def render_parallel(renderer: ParallelRenderer, config: Config, jobs_per_thread: int) -> np.ndarray:
    """``render_parallel(&mut renderer, config, jobs_per_thread) -> FinalImage`` (src/lib.rs:1051)."""
    out = np.empty((config.c.height, config.c.width, 4), dtype=np.uint16)
    _check(_lib().sar_render_parallel(renderer._h, C.byref(config.c), jobs_per_thread,
                                      out.ctypes.data_as(C.POINTER(C.c_uint16))), "sar_render_parallel")
    return out


def render_parallel_into(renderer: ParallelRenderer, config: Config, jobs_per_thread: int, rgba_host_ptr: int):
    """render_parallel into caller-owned host memory (width*height*4 uint16; e.g. pinned memory: every device then
    copies its slice straight into it)."""
    _check(_lib().sar_render_parallel(renderer._h, C.byref(config.c), jobs_per_thread,
                                      C.cast(C.c_void_p(rgba_host_ptr), C.POINTER(C.c_uint16))), "sar_render_parallel")
