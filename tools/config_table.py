"""Times one GPU's share of every BASELINE.json config through the public API (reset -> render_job_range ->
colorize_device), with the library's own HIP-event spans per stage. A tool for DESIGN.md's tables, not the bench.

  C2  poisson-saturne, 1e9 iterations, 2048x2048, gas
  C3  solar-sail, 1e9 iterations, 1800x2000, depth texture, scale 1.0 (what the CLI renders)
  C4  poisson-saturne, 4096x4096, 1.25e9 iterations = one GPU's eighth of 1e10
  C5  solar-sail sequence frame: 1e8 iterations, 1800x2000, gas
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strange_attractor_renderer_amd as S  # noqa: E402
S.use_hooks_build()   # this tool turns A/B options (include/sar_test_hooks.h)
import torch  # noqa: E402

CONFIGS = {
    "C2": dict(preset="poisson_saturne", iters=1e9, w=2048, h=2048, kind=0),
    "C3": dict(preset="solar_sail", iters=1e9, w=1800, h=2000, kind=1, scale=1.0),
    "C4/8": dict(preset="poisson_saturne", iters=1.25e9, w=4096, h=4096, kind=0),
    "C5-frame": dict(preset="solar_sail", iters=1e8, w=1800, h=2000, kind=0, scale=1.0),
    # not BASELINE configs: intermediate sizes for choosing size-dependent defaults (select with --only)
    "X2560": dict(preset="poisson_saturne", iters=1e9, w=2560, h=2560, kind=0),
    "X3072": dict(preset="poisson_saturne", iters=1e9, w=3072, h=3072, kind=0),
    "XC4": dict(preset="poisson_saturne", iters=1e10, w=4096, h=4096, kind=0),  # the whole configs[3] frame: --jobs 1048576
    "X8192": dict(preset="poisson_saturne", iters=1e9, w=8192, h=8192, kind=0),  # 64 Mpx: 1024 bins of 65536 pixels
    "X1024": dict(preset="poisson_saturne", iters=1e9, w=1024, h=1024, kind=0),
    "X1448": dict(preset="poisson_saturne", iters=1e9, w=1448, h=1448, kind=0),
    "XHD": dict(preset="poisson_saturne", iters=1e9, w=1920, h=1080, kind=0),
    "X4K": dict(preset="poisson_saturne", iters=1e9, w=3840, h=2160, kind=0),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, nargs="+", default=[131072])
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--option", nargs="*", default=[], help="name=value runtime options")
    ap.add_argument("--out", default="gpurun_out/config_table.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    opts = dict((kv.split("=")[0], int(kv.split("=")[1], 0)) for kv in a.option)
    with open(a.out, "a") as fh:
        for name, c in CONFIGS.items():
            if (a.only and name not in a.only) or (not a.only and name.startswith("X")):
                continue
            for jobs in a.jobs:
                n = int(c["iters"]) // jobs
                kw = dict(iterations=jobs * n, width=c["w"], height=c["h"], jobs_total=jobs, render_kind=c["kind"], seed=1)
                if "scale" in c:
                    kw["scale"] = c["scale"]
                cfg = getattr(S.Config, c["preset"])(**kw)
                starts = S.start_points(1, 0, jobs)
                rt = S.Runtime(cfg)
                rt.enable_timing(True)
                for k, v in opts.items():
                    rt.set_option(k, v)
                rgba = torch.empty(c["w"] * c["h"] * 4, dtype=torch.int16, device="cuda")
                best = None
                for _ in range(a.reps):
                    rt.synchronize()
                    t0 = time.perf_counter()
                    rt.reset()
                    S.render_job_range(cfg, rt, n, starts)
                    S.colorize_device(cfg, rt, rgba.data_ptr())
                    rt.synchronize()
                    wall = (time.perf_counter() - t0) * 1e3
                    t = rt.last_timing()
                    rec = dict(config=name, jobs=jobs, iters=jobs * n, wall_ms=round(wall, 3), iterate_ms=round(t.iterate_ms, 3),
                               fold_ms=round(t.resolve_ms, 3), colorize_ms=round(t.colorize_ms, 3), warmup_ms=round(t.warmup_ms, 3), launches=t.iterate_launches,
                               depth_atomics=t.depth_atomics, depth_candidates=t.depth_candidates, git_per_s=round(jobs * n / wall / 1e6, 2), launch=rt.describe_last_launch().split(" | ")[0], **opts)
                    if best is None or rec["wall_ms"] < best["wall_ms"]:
                        best = rec
                rt.close()
                print(json.dumps(best), flush=True)
                fh.write(json.dumps(best) + "\n")
