"""Offline differential fuzz: the two seeded random-configuration parity tests of tests/test_gpu_parity.py over many more
seeds than the suite runs. python tools/fuzz_parity.py FIRST LAST   (needs a GPU; prints the seeds that differ)"""
import sys, os, traceback
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import strange_attractor_renderer_amd as sar
sar.use_hooks_build()   # this tool turns A/B options (include/sar_test_hooks.h)
import oracle_lib as oracle
import test_gpu_parity as T
oracle.lib()

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in ([] if len(sys.argv) > 3 else range(first, last)):
    for fn in (T.test_random_configurations_bit_exact, T.test_custom_attractors_views_and_transforms_bit_exact):
        try:
            getattr(fn, "__wrapped__", fn)(sar, oracle, None, seed)
        except Exception as e:  # noqa: BLE001
            bad.append((fn.__name__, seed))
            print("MISMATCH", fn.__name__, seed, repr(e)[:300], flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "big":
    # larger shapes than the suite's: images up to 5000 x 5000 (bins of 65 536 pixels, narrow hints, shared hint array),
    # up to 150 000 jobs, ~5e7 iterations per case; both presets, both kinds, either hint width / stager / kernel form
    for seed in range(first, last):
        rng = np.random.default_rng(555_000 + seed)
        preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
        w, h = int(rng.integers(500, 5000)), int(rng.integers(500, 5000))
        jobs = int(rng.integers(1000, 150000))
        n = max(1, 50_000_000 // jobs)
        kw = dict(iterations=jobs * n, width=w, height=h, jobs_total=jobs, render_kind=int(rng.integers(2)),
                  angle=float(rng.uniform(0, 6.3)), scale=float(rng.uniform(0.5, 2.0)))
        cfg = getattr(sar.Config, preset)(**kw)
        st = sar.start_points(int(rng.integers(1 << 30)), 0, jobs)
        opts = {}
        if rng.integers(3) == 0: opts["hint_bits"] = [16, 32][int(rng.integers(2))]
        if rng.integers(3) == 0: opts["split_waves"] = [1, 2][int(rng.integers(2))]
        if rng.integers(4) == 0: opts["hint_shared"] = [1, 2][int(rng.integers(2))]
        if rng.integers(4) == 0: opts["debug_chunk_jobs"] = int(rng.integers(2000, 40000))
        if rng.integers(4) == 0:  # a power-of-two width (and a height that is a multiple of eight): the narrow hints then live in 8 x 8 tiles
            w, h = 1 << int(rng.integers(9, 13)), 8 * int(rng.integers(60, 600))
            kw.update(width=w, height=h)
            cfg = getattr(sar.Config, preset)(**kw)
            opts["hint_bits"] = 16
            if rng.integers(3) == 0: opts["hint_tile"] = 1
        rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
        for k, v in opts.items(): rt.set_option(k, v)
        try:
            sar.render_jobs(cfg, rt, st)
            oracle.render_jobs(cfg.c, ort, st, n)
            T.assert_state_equal(rt, ort, f"big seed {seed}")
            np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))
            print("ok", seed, preset, f"{w}x{h}", jobs, n, opts, rt.describe_last_launch()[:60], flush=True)
        except Exception as e:  # noqa: BLE001
            bad.append(("big", seed)); print("MISMATCH big", seed, preset, w, h, jobs, n, opts, repr(e)[:300], flush=True)
        rt.close()
if len(sys.argv) > 3 and sys.argv[3] == "shards":
    # the native multi-device renderer over 1..8 shards of device 0, two frames each (the start-point stream runs on)
    for seed in range(first, last):
        rng = np.random.default_rng(888_000 + seed)
        preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
        k = int(rng.integers(1, 9))
        w, h = int(rng.integers(30, 3000)), int(rng.integers(30, 2500))
        units, jpu = int(rng.integers(1, 4000)), int(rng.integers(1, 6))
        n = max(1, 20_000_000 // (units * jpu))
        cfg = getattr(sar.Config, preset)(iterations=units * jpu * n + int(rng.integers(units * jpu)), width=w, height=h,
                                          render_kind=int(rng.integers(2)), transparent=int(rng.integers(2)),
                                          angle=float(rng.uniform(0, 6.3)), scale=float(rng.uniform(0.5, 2.0)))
        sd = int(rng.integers(1 << 30))
        try:
            pr = sar.ParallelRenderer(devices=[0] * k, units=units, seed=sd)
            for frame in range(2):
                c = cfg.replace(angle=cfg.angle + 0.3 * frame)
                img = sar.render_parallel(pr, c, jpu)
                ort = oracle.Runtime(w, h)
                oracle.render_jobs(c.replace(jobs_total=units * jpu).c, ort, sar.start_points(sd, frame * units * jpu, units * jpu), n)
                np.testing.assert_array_equal(img, oracle.colorize(c.c, ort))
            T.assert_state_equal(pr.runtime(), ort, f"shards seed {seed}")
            pr.shutdown()
            print("ok", seed, preset, f"{w}x{h}", "shards", k, "units", units, "jpu", jpu, "n", n, flush=True)
        except Exception as e:  # noqa: BLE001
            bad.append(("shards", seed)); print("MISMATCH shards", seed, preset, w, h, k, units, jpu, n, repr(e)[:300], flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "announce":
    # one runtime, a random series of calls: sizes up and down, one or several launch chunks, announced / announced with
    # other points / not announced, with and without a reset in between (an un-reset runtime accumulates, :742-744)
    import torch
    for seed in range(first, last):
        rng = np.random.default_rng(333_000 + seed)
        preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
        w, h = int(rng.integers(40, 1500)), int(rng.integers(40, 1200))
        base = getattr(sar.Config, preset)(width=w, height=h, render_kind=int(rng.integers(2)))
        rt, ort = sar.Runtime(base), oracle.Runtime(w, h)
        if rng.integers(2): rt.set_option("hint_bits", [16, 32][int(rng.integers(2))])
        log = []
        try:
            keep = []
            for call in range(8):
                jobs, n = int(rng.integers(1, 30000)), int(rng.integers(1, 400))
                cfg = base.replace(iterations=jobs * n, jobs_total=jobs, angle=float(rng.uniform(0, 6.3)))
                if rng.integers(3) == 0: rt.set_option("debug_chunk_jobs", int(rng.integers(300, 20000)))
                st = sar.start_points(int(rng.integers(1 << 30)), 0, jobs)
                dev = torch.from_numpy(st).cuda(); keep.append(dev)
                torch.cuda.synchronize()
                mode = int(rng.integers(4))       # 0 not announced, 1 announced, 2 announced under another view, 3 announced with other points
                if rng.integers(3) > 0:
                    rt.reset(); ort.reset()
                if mode == 1: sar.prefetch_device(cfg, rt, jobs, n, dev.data_ptr())
                if mode == 2: sar.prefetch_device(cfg.replace(angle=1.0, scale=1.3), rt, jobs, n, dev.data_ptr())
                if mode == 3 and keep[:-1]: sar.prefetch_device(cfg, rt, min(jobs, keep[0].shape[0]), n, keep[0].data_ptr())
                sar.render_job_range_device(cfg, rt, jobs, n, dev.data_ptr())
                oracle.render_jobs(cfg.c, ort, st, n)
                log.append((jobs, n, mode))
                T.assert_state_equal(rt, ort, f"announce seed {seed} call {call} {log}")
            print("ok", seed, preset, f"{w}x{h}", log, flush=True)
        except Exception as e:  # noqa: BLE001
            bad.append(("announce", seed)); print("MISMATCH announce", seed, preset, w, h, log, repr(e)[:300], flush=True)
        rt.close()
cases = (last - first) * (1 if len(sys.argv) > 3 else 2)
print(f"seeds {first}..{last - 1}: {cases - len(bad)} of {cases} cases bit-exact, {len(bad)} differ: {bad}")
