"""GPU-side exploration: times kernel variants of the iterate/accumulate path (HIP events inside the
library). Not a test and not the bench — a tool to choose defaults and to look for the bottleneck."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import strange_attractor_renderer_amd as S  # noqa: E402


def run(cfg, starts, block, stride, variant, reps=4, **more):
    rt = S.Runtime(cfg)
    rt.enable_timing(True)
    rt.set_tuning(block_threads=block, checkpoint_stride=stride, variant=variant, **more)
    best = None
    for _ in range(reps):
        rt.reset()
        rt.synchronize()
        t0 = time.perf_counter()
        S.render_jobs(cfg, rt, starts)
        rt.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        t = rt.last_timing()
        rec = dict(iter_ms=t.iterate_ms, fold_ms=t.resolve_ms, wall_ms=wall, iters=t.iterations_counted, depth_atomics=t.depth_atomics)
        if best is None or rec["iter_ms"] < best["iter_ms"]:
            best = rec
    rt.close()
    return best


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e9)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--preset", default="poisson_saturne")
    ap.add_argument("--jobs", type=int, nargs="+", default=[65536, 131072, 262144])
    ap.add_argument("--blocks", type=int, nargs="+", default=[64, 256])
    ap.add_argument("--variants", type=lambda s: int(s, 0), nargs="+",
                    default=[0x01, 0x02, 0x11, 0x12, 0x21])
    ap.add_argument("--stride", type=int, nargs="+", default=[64])
    ap.add_argument("--bin-shift", type=int, nargs="+", default=[0])
    ap.add_argument("--splits", type=int, nargs="+", default=[0])
    ap.add_argument("--acc-threads", type=int, nargs="+", default=[0])
    ap.add_argument("--records", type=int, nargs="+", default=[0])
    ap.add_argument("--opt", action="append", default=[], help="extra runtime option name=value (repeatable), e.g. stager=1")
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default="gpurun_out/perf_explore.jsonl")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "a") as f:
        for jobs in a.jobs:
            cfg = getattr(S.Config, a.preset)(iterations=int(a.n), width=a.size, height=a.size, jobs_total=jobs,
                                              scale=1.0)
            starts = S.start_points(1, 0, jobs)
            for block in a.blocks:
                for variant in a.variants:
                    for stride, bs, sp, at, rc in [(x, y, z, w, v) for x in a.stride for y in a.bin_shift for z in a.splits for w in a.acc_threads for v in a.records]:
                        extra = {k: int(v) for k, v in (o.split("=") for o in a.opt)}
                        r = run(cfg, starts, block, stride, variant, bin_shift=bs, splits=sp, acc_threads=at, chunk_records=rc, **extra)
                        r.update(extra, tag=a.tag, lib=os.environ.get("SAR_LIBRARY", ""))
                        r.update(jobs=jobs, block=block, variant=hex(variant), stride=stride, size=a.size,
                                 bin_shift=bs, splits=sp, acc_threads=at, records=rc,
                                 preset=a.preset,
                                 git_per_s_kernel=r["iters"] / r["iter_ms"] / 1e6,
                                 git_per_s_wall=r["iters"] / r["wall_ms"] / 1e6)
                        print(json.dumps(r), flush=True)
                        f.write(json.dumps(r) + "\n")
