"""Wall time per frame of one GPU's share (45 frames) of BASELINE configs[4]: the 360-frame solar-sail sweep, 1e8 iterations,
1800x2000, RGB16 conversion on the device, read-back included, no encoder."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strange_attractor_renderer_amd as S
from strange_attractor_renderer_amd.sequence import render_sequence
cfg = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, transparent=0)
n = [0]
def sink(k, name, img): n[0] += 1
for rep in range(2):
    n[0] = 0
    t0 = time.perf_counter()
    render_sequence(cfg, 0.0, 360.0, 1.0, rank=0, world=8, seed=4, sink=sink, image_format=S.SAR_FMT_RGB16)
    dt = time.perf_counter() - t0
    print("frames", n[0], "total s %.3f" % dt, "ms/frame %.2f" % (dt / n[0] * 1e3), flush=True)
