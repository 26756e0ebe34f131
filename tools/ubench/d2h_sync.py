"""Synchronous hipMemcpy of one image (21.6 MB) device -> page-locked host: its rate, alone and while a compute kernel keeps the
chip busy — and (under `rocprofv3 --memory-copy-trace --kernel-trace`) whether a copy engine or a blit kernel carries it.
    python tools/ubench/d2h_sync.py"""
import ctypes as C
import threading
import time

import torch

hip = C.CDLL("libamdhip64.so")
n = 1800 * 2000 * 6
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]


def rate(fn, reps=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


sync = lambda: hip.hipMemcpy(dst.data_ptr(), src.data_ptr(), n, 2)
s = torch.cuda.Stream()
asyn = lambda: hip.hipMemcpyAsync(dst.data_ptr(), src.data_ptr(), n, 2, s.cuda_stream)
for name, fn in (("hipMemcpy (synchronous)", sync), ("hipMemcpyAsync", asyn)):
    el = rate(fn)
    print(f"{name}: {el * 1e3:.3f} ms per image, {n / el / 1e9:.1f} GB/s alone", flush=True)

# the same while a matmul loop keeps the CUs busy on another stream
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
busy = torch.cuda.Stream()
stop = False


def burn():
    with torch.cuda.stream(busy):
        while not stop:
            for _ in range(4):
                torch.mm(a, a)
            busy.synchronize()


t0 = time.perf_counter()
with torch.cuda.stream(busy):
    for _ in range(8):
        torch.mm(a, a)
busy.synchronize()
mm = (time.perf_counter() - t0) / 8
th = threading.Thread(target=burn)
th.start()
time.sleep(0.2)
for name, fn in (("hipMemcpy (synchronous)", sync), ("hipMemcpyAsync", asyn)):
    el = rate(fn)
    print(f"{name}: {el * 1e3:.3f} ms per image, {n / el / 1e9:.1f} GB/s under a matmul loop ({mm * 1e3:.1f} ms each alone)", flush=True)
stop = True
th.join()
