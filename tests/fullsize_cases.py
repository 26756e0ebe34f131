"""The full-size frames of BASELINE.json's configs as (oracle config, start points, iterations per job) — shared by
tests/golden/make_fullsize_checksums.py (which freezes the oracle's checksums) and tests/test_gpu_fullsize.py."""
import math

import numpy as np

CASES = ("c2_131072", "c2_65536", "c3_solar_depth", "c4_rank5_share", "c4_all_jobs", "c4_full_1e10", "c5_frame37",
         "c5_frame37_65536")


def frame_seed(seed: int, k: int) -> int:   # strange_attractor_renderer_amd.sequence.frame_seed, restated
    return (seed + 0x9E3779B97F4A7C15 * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def build_case(name: str, O):
    """O: tests/oracle_lib. Returns (SarConfig, starts[n_jobs,3], iters_per_job)."""
    if name in ("c2_131072", "c2_65536"):          # configs[1]: poisson-saturne, 1e9, 2048^2 (SURVEY 8d C2 = 65 536 jobs;
        jobs = int(name.split("_")[1])             # bench.py runs 131 072)
        cfg = O.poisson_saturne()
        cfg.width = cfg.height = 2048
        cfg.transparent = 0
        n = 1_000_000_000 // jobs
        return _fin(cfg, jobs, n), O.start_points(1, 0, jobs), n
    if name == "c3_solar_depth":                   # configs[2]: solar-sail, 1e9, 1800x2000, depth texture, CLI scale 1
        jobs = 131072
        cfg = O.solar_sail()
        cfg.width, cfg.height, cfg.scale, cfg.render_kind = 1800, 2000, 1.0, O.SAR_RENDER_DEPTH
        n = 1_000_000_000 // jobs
        return _fin(cfg, jobs, n), O.start_points(1, 0, jobs), n
    if name == "c4_rank5_share":                   # configs[3]: rank 5's eighth of 1e10 iterations / 524 288 jobs, 4096^2
        jobs, n = 65536, 19073
        cfg = O.poisson_saturne()
        cfg.width = cfg.height = 4096
        return _fin(cfg, jobs, n), O.start_points(3, 5 * jobs, jobs), n
    if name == "c4_all_jobs":                      # configs[3]'s whole job list on ONE GPU (bench.py --config c4: 1 048 576 jobs,
        jobs, n = 1048576, 953                     # 8 rounds of resident workgroups in a launch), 1e9 iterations instead of 1e10
        cfg = O.poisson_saturne()
        cfg.width = cfg.height = 4096
        return _fin(cfg, jobs, n), O.start_points(3, 0, jobs), n
    if name == "c4_full_1e10":                     # configs[3] itself: 1e10 iterations as bench.py --config c4 cuts them (1 048 576 jobs
        jobs, n = 1048576, 9536                    # x 9536), the whole frame on ONE GPU: three launch chunks of whole rounds
        cfg = O.poisson_saturne()
        cfg.width = cfg.height = 4096
        return _fin(cfg, jobs, n), O.start_points(3, 0, jobs), n
    if name in ("c5_frame37", "c5_frame37_65536"):  # configs[4]: frame 37 of the 360-frame solar-sail sweep, 1e8 iterations
        units, jpt, k = 16384, 12, 37              # (CLI defaults: 12 jobs per thread, scale 1, Gas)
        seed = 0
        if name == "c5_frame37_65536":             # ... as `bench.py --config c5` cuts it: 65 536 jobs per frame, seed 4 — the
            jpt, seed = 4, 4                       # frame its line checks itself against
        jobs = units * jpt
        n = 100_000_000 // units // jpt
        cfg = O.solar_sail()
        cfg.width, cfg.height, cfg.scale, cfg.transparent = 1800, 2000, 1.0, 0
        cfg.angle = k * math.pi / 180.0
        return _fin(cfg, jobs, n), O.start_points(frame_seed(seed, k), 0, jobs), n
    raise KeyError(name)


def _fin(cfg, jobs, n):
    cfg.jobs_total = jobs
    cfg.iterations = jobs * n
    return cfg
