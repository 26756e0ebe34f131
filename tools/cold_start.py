"""Wall time of the FIRST frame of a process (what a one-image CLI run pays): library load, runtime, allocations, kernels.
python tools/cold_start.py"""
import sys, time
sys.path.insert(0, ".")
t0 = time.perf_counter()
import numpy as np
import strange_attractor_renderer_amd as S
t1 = time.perf_counter()
cfg = S.Config.poisson_saturne(iterations=1_000_000_000, width=2048, height=2048, jobs_total=131072)
starts = S.start_points(1, 0, 131072)
t2 = time.perf_counter()
rt = S.Runtime(cfg)
rt.synchronize()
t3 = time.perf_counter()
S.render_jobs(cfg, rt, starts)
rt.synchronize()
t4 = time.perf_counter()
img = S.colorize(cfg, rt)
t5 = time.perf_counter()
rt.reset(); S.render_jobs(cfg, rt, starts); rt.synchronize()
t6 = time.perf_counter()
img = S.colorize(cfg, rt)
t7 = time.perf_counter()
print(f"import {t1 - t0:.3f} s | config + start points {t2 - t1:.3f} | Runtime() {t3 - t2:.3f} | first render {t4 - t3:.3f} | "
      f"first colorize + read-back {t5 - t4:.3f} | second render {t6 - t5:.4f} | second colorize + read-back {t7 - t6:.4f}")
