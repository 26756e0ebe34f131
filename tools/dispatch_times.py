"""Dispatch-by-dispatch duration of the headline iterate kernel over the bench's own frame loop (BASELINE configs[1], announced
frames), from the HIP events the library records around every launch. Run it plain and under `rocprofv3 --kernel-trace`: the same
dispatches, two clocks — does the tracer's bimodal population (profiles/r05_rocprofv3_summary.md: pairs of dispatches at +0.75 ms)
exist without the tracer?

python tools/dispatch_times.py [--steps 24] [--no-prefetch] [--json out.json]"""
import argparse
import json
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    import numpy as np
    import torch
    import strange_attractor_renderer_amd as S
    S.use_hooks_build()
    jobs, n = 131072, 1_000_000_000 // 131072
    cfg = S.Config.poisson_saturne(iterations=n * jobs, width=2048, height=2048, jobs_total=jobs, transparent=0, seed=1)
    starts = S.start_points(1, 0, jobs)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rt = S.Runtime(cfg, device=0)
        rt.set_stream(stream.cuda_stream)
        rt.enable_timing(True)
        rgba = torch.empty(2048 * 2048 * 4, dtype=torch.int16, device="cuda")
        starts_dev = torch.from_numpy(np.ascontiguousarray(starts)).cuda()

        def step(more):
            rt.reset()
            S.render_job_range_device(cfg, rt, jobs, n, starts_dev.data_ptr())
            if more and not a.no_prefetch:
                S.prefetch_device(cfg, rt, jobs, n, starts_dev.data_ptr())
            S.colorize_device(cfg, rt, rgba.data_ptr())

        for k in range(a.warmup):
            step(k + 1 < a.warmup)
        torch.cuda.synchronize()
        rt.last_timing()
        rt.set_option("timing_accumulate", 1)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        marks[0].record()
        for k in range(a.steps):
            step(k + 1 < a.steps)
            marks[k + 1].record()
        torch.cuda.synchronize()
        it = rt.debug_spans(0)
        fold = rt.debug_spans(1)
        warm = rt.debug_spans(2)
        frames = [marks[k].elapsed_time(marks[k + 1]) for k in range(a.steps)]
    srt = sorted(it)
    out = {"steps": a.steps, "prefetch": not a.no_prefetch, "iterate_ms": [round(x, 4) for x in it],
           "fold_ms": [round(x, 4) for x in fold], "warm_ms": [round(x, 4) for x in warm], "frame_ms": [round(x, 4) for x in frames],
           "iterate_min_median_max": [srt[0], srt[len(srt) // 2], srt[-1]], "iterate_mean": sum(it) / len(it),
           "slow_dispatches": [k for k, x in enumerate(it) if x > srt[len(srt) // 2] * 1.08]}
    print(json.dumps(out))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
