"""Delivery time of every frame of a `sequence` sweep on one GPU (configs[4] shape): python tools/seq_frame_times.py [lanes] [frames]."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import strange_attractor_renderer_amd as S
from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 360
cfg = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, transparent=0)
with SequenceRenderer(cfg, units=16384, jobs_per_thread=4, seed=4, image_format=S.SAR_FMT_RGB16, lanes=lanes) as seq:
    for sweep in range(2):
        t = []
        seq.run(frames(0.0, float(n), 1.0), sink=lambda k, name, img: t.append(time.perf_counter()))
        d = np.diff(np.array(t)) * 1e3
        print(f"lanes {lanes} sweep {sweep}: mean {d.mean():.3f} ms; per 30 frames:", " ".join(f"{d[i:i + 30].mean():.2f}" for i in range(0, len(d), 30)))
