import sys, time
sys.path.insert(0, ".")
import numpy as np
import strange_attractor_renderer_amd as S
cfg = S.Config.poisson_saturne(iterations=1_000_000_000, width=2048, height=2048, transparent=0)
pr = S.ParallelRenderer(units=16384, seed=1)
for k in range(3): S.render_parallel(pr, cfg.replace(angle=0.01 * k), 8)
t0 = time.perf_counter()
for k in range(20): img = S.render_parallel(pr, cfg.replace(angle=0.1 + 0.01 * k), 8)
print("render_parallel into a fresh numpy image (pageable): %.2f ms per frame" % ((time.perf_counter() - t0) / 20 * 1e3))
out = np.empty((2048, 2048, 4), np.uint16)
t0 = time.perf_counter()
for k in range(20): S.render_parallel_into(pr, cfg.replace(angle=0.5 + 0.01 * k), 8, out.ctypes.data)
print("render_parallel_into one pageable image: %.2f ms per frame" % ((time.perf_counter() - t0) / 20 * 1e3))
pr.shutdown()
