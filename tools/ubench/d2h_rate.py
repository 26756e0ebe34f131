"""Device -> page-locked host copy rate of one image (21.6 MB, configs[4]'s RGB16 frame) on this box: one stream, and the image cut
into 2 / 4 pieces on as many streams (does a second copy engine take part?).   python tools/ubench/d2h_rate.py"""
import time

import torch

n = 1800 * 2000 * 6
src = torch.empty(n, dtype=torch.uint8, device="cuda")
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
for pieces in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(pieces)]
    step = n // pieces
    def once():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    print(f"{pieces} piece(s) on {pieces} stream(s): {el * 1e3:.3f} ms per image, {n / el / 1e9:.1f} GB/s", flush=True)
