"""Times sar_render_jobs_batch on BASELINE configs[4]'s frame (solar-sail, 1e8 iterations, 1800x2000, 65 536 jobs) for a list of batch sizes
and runtime options: per-frame wall time of reset + batched render + colorize to RGBA16 in HBM on ONE stream, and the library's own HIP-event
spans of the warm-up / iterate / accumulate+fold launches.

    python tools/batch_time.py --frames 1,2,3,4,6,8 [--opt hint_bits=16] [--reps 40] [--preset solar_sail]
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,2,3,4,6,8")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--jobs", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=100_000_000)
    ap.add_argument("--size", default="1800x2000")
    ap.add_argument("--preset", default="solar_sail")
    ap.add_argument("--opt", action="append", default=[], help="name=value runtime option of the batch leader")
    a = ap.parse_args()
    import torch
    import strange_attractor_renderer_amd as S
    if a.opt:
        S.use_hooks_build()   # A/B options live in the hooks build (include/sar_test_hooks.h)
    from strange_attractor_renderer_amd.sequence import frame_seed
    w, h = (int(v) for v in a.size.split("x"))
    n = a.iters // a.jobs
    for F in (int(v) for v in a.frames.split(",")):
        cfgs = [getattr(S.Config, a.preset)(iterations=n * a.jobs, width=w, height=h, scale=1.0, transparent=0, jobs_total=a.jobs,
                                            angle=k * math.pi / 180.0) for k in range(F)]
        starts = [S.start_points(frame_seed(4, k), 0, a.jobs) for k in range(F)]
        rts = [S.Runtime(c) for c in cfgs]
        for rt in rts[1:]:
            rt.set_stream(rts[0].stream())
        for o in a.opt:
            k, v = o.split("=")
            rts[0].set_option(k, int(v))
        rts[0].enable_timing(True)
        out = [torch.empty(w * h * 4, dtype=torch.int16, device="cuda") for _ in range(F)]

        def batch():
            for rt in rts:
                rt.reset()
            if F > 1:
                S.render_jobs_batch(cfgs, rts, starts)
            else:
                S.render_jobs(cfgs[0], rts[0], starts[0])
            for c, rt, o in zip(cfgs, rts, out):
                S.colorize_device(c, rt, o.data_ptr())

        for _ in range(4):
            batch()
        rts[0].synchronize()
        rts[0].set_option("timing_accumulate", 1)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            batch()
        rts[0].synchronize()
        el = time.perf_counter() - t0
        tm = rts[0].last_timing()
        per = a.reps * F
        print(json.dumps({"frames_per_launch": F, "ms_per_frame": el / per * 1e3, "warmup_ms_per_frame": tm.warmup_ms / per,
                          "iterate_ms_per_frame": tm.iterate_ms / per, "accumulate_fold_ms_per_frame": tm.resolve_ms / per,
                          "iterate_ms_per_launch": tm.iterate_ms / a.reps, "options": a.opt, "launch": rts[0].describe_last_launch()}), flush=True)
        for rt in reversed(rts):
            rt.close()


if __name__ == "__main__":
    main()
