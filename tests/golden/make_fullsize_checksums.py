"""Generates tests/golden/fullsize_checksums.json: FNV-1a-64 checksums of the CPU oracle's buffers for the
FULL-SIZE frames of BASELINE.json's configs (run in the build container; ~1-2 minutes on 8 cores).

The oracle is sar_oracle_render_jobs_mt — bit-identical to the sequential sar_oracle_render_jobs (asserted on a
small frame before anything is written). tests/test_gpu_fullsize.py holds the GPU frames to these checksums AND to
a live oracle render on the GPU box's host cores, so a full-size result is pinned between builds and machines.

    python tests/golden/make_fullsize_checksums.py
"""
import json
import math
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from fullsize_cases import CASES, build_case  # noqa: E402


def checksums(cfg, rt):
    img = O.colorize(cfg, rt)
    return {"max": rt.max, "count_sum": int(rt.count.sum(dtype=np.uint64)), "touched": int((rt.count > 0).sum()),
            "count_fnv": f"{O.fnv1a64(rt.count):016x}", "zbuf_fnv": f"{O.fnv1a64(rt.zbuf):016x}",
            "steps_fnv": f"{O.fnv1a64(rt.steps):016x}", "rgba_fnv": f"{O.fnv1a64(img):016x}"}


if __name__ == "__main__":
    # the threaded oracle is the sequential oracle (small frame, both presets) before it is trusted at full size
    for preset in (O.poisson_saturne, O.solar_sail):
        c = preset()
        c.width, c.height, c.scale = 200, 160, 1.0
        st = O.start_points(11, 0, 53)
        a, b = O.Runtime(200, 160), O.Runtime(200, 160)
        O.render_jobs(c, a, st, 5000)
        O.render_jobs_mt(c, b, st, 5000, 5)
        assert np.array_equal(a.count, b.count) and a.max == b.max
        assert np.array_equal(a.zbuf.view(np.uint32), b.zbuf.view(np.uint32))
        assert np.array_equal(a.steps.view(np.uint64), b.steps.view(np.uint64))
    out = {}
    only = sys.argv[1:]   # `make_fullsize_checksums.py <case> ...`: (re)generate these cases only, keep the others
    path = os.path.join(HERE, "fullsize_checksums.json")
    if only and os.path.exists(path):
        out = json.load(open(path))
    for name in (only or CASES):
        cfg, starts, n = build_case(name, O)
        rt = O.Runtime(cfg.width, cfg.height)
        t0 = time.time()
        O.render_jobs_mt(cfg, rt, starts, n)
        out[name] = checksums(cfg, rt)
        out[name]["jobs"], out[name]["iters_per_job"] = int(starts.shape[0]), int(n)
        print(name, out[name], f"{time.time() - t0:.1f} s", flush=True)
    with open(os.path.join(HERE, "fullsize_checksums.json"), "w") as f:
        json.dump(out, f, indent=1)
