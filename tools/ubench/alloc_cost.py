"""What allocation costs on this box — the numbers behind the frame group's slab (csrc/sar_runtime.cpp: sar_runtime_new_group) and
the sweep's ring of page-locked images (sequence.py): hipMalloc / hipHostMalloc / hipFree of the sizes a 1800x2000 sweep uses,
as many small calls or one large one, with the GPU idle and with a kernel running (a configs[1] frame on another stream).

python tools/ubench/alloc_cost.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import strange_attractor_renderer_amd as S

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipHostFree.argtypes = [C.c_void_p]

IMG = 1800 * 2000 * 6
MB = 1 << 20


def t(fn):
    t0 = time.perf_counter()
    r = fn()
    return (time.perf_counter() - t0) * 1e3, r


def dev_alloc(n, size):
    ps = []
    for _ in range(n):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), size) == 0
        ps.append(p)
    return ps


def host_alloc(n, size):
    ps = []
    for _ in range(n):
        p = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(p), size, 0) == 0
        ps.append(p)
    return ps


def free_all(ps, host=False):
    for p in ps:
        (hip.hipHostFree if host else hip.hipFree)(p)


cfg = S.Config.poisson_saturne(iterations=7629 * 131072, width=2048, height=2048, jobs_total=131072, transparent=0, seed=1)
rt = S.Runtime(cfg)
starts = S.start_points(1, 0, 131072)
S.render_jobs(cfg, rt, starts)
rt.synchronize()


def busy(frames=6):
    for _ in range(frames):       # ~6 ms of kernels each, enqueued only
        rt.reset()
        S.render_jobs(cfg, rt, starts)


# page-locking memory the process already owns: anonymous mmap, transparent huge pages asked for, touched, then hipHostRegister
import mmap
libc = C.CDLL("libc.so.6", use_errno=True)
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
for huge in (False, True):
    for _ in range(2):
        n = (17 * IMG + (2 << 20) - 1) & ~((2 << 20) - 1)
        t0 = time.perf_counter()
        m = mmap.mmap(-1, n, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
        addr = C.addressof(C.c_char.from_buffer(m))
        if huge:
            libc.madvise(C.c_void_p(addr), C.c_size_t(n), 14)   # MADV_HUGEPAGE
        t1 = time.perf_counter()
        np.frombuffer(m, dtype=np.uint8)[::4096] = 1            # touch every page
        t2 = time.perf_counter()
        rc = hip.hipHostRegister(C.c_void_p(addr), n, 0)
        t3 = time.perf_counter()
        hip.hipHostUnregister(C.c_void_p(addr))
        t4 = time.perf_counter()
        print(f"mmap {n >> 20} MiB huge={huge}: mmap {1e3 * (t1 - t0):.2f} touch {1e3 * (t2 - t1):.2f} hipHostRegister {1e3 * (t3 - t2):.2f} (rc {rc}) unregister {1e3 * (t4 - t3):.2f} ms", flush=True)
        del m
try:
    print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
except Exception as e:
    print("THP: ?", e)

for state in ("idle", "busy"):
    for label, fn, host in (
        ("hipHostMalloc 17 x 21.6 MB", lambda: host_alloc(17, IMG), True),
        ("hipHostMalloc 1 x 367 MB", lambda: host_alloc(1, 17 * IMG), True),
        ("hipHostMalloc 33 x 21.6 MB", lambda: host_alloc(33, IMG), True),
        ("hipHostMalloc 1 x 713 MB", lambda: host_alloc(1, 33 * IMG), True),
        ("hipMalloc 320 x 32 MB", lambda: dev_alloc(320, 32 * MB), False),
        ("hipMalloc 16 x 640 MB", lambda: dev_alloc(16, 640 * MB), False),
        ("hipMalloc 1 x 10 GB", lambda: dev_alloc(1, 10240 * MB), False),
        ("hipMalloc 4 x 2.5 GB", lambda: dev_alloc(4, 2560 * MB), False),
    ):
        reps = []
        for _ in range(3):
            if state == "busy":
                busy()
            ms, ps = t(fn)
            if state == "busy":
                busy(2)
            fms, _ = t(lambda: free_all(ps, host))
            rt.synchronize()
            reps.append((round(ms, 2), round(fms, 2)))
        print(f"{state:5s} {label:30s} alloc / free ms: {reps}", flush=True)
