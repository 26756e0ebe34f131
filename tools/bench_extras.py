"""tools/bench_extras.py — what bench.py reports NEXT to its one timed loop, imported on demand: the PMC child passes behind
`roofline.traffic`, the CPU baseline (the only place that touches oracle/), the frame checksums against the committed goldens,
the sustained / two-stream loops, the C-ABI-only multi-device renderer, the `sequence` sweep (BASELINE configs[4]) and the
strong-scaling + native legs of an N > 1 run. None of it runs inside bench.py's timed region.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COLD_SETTLE_S = 1.0   # idle time before every cold construction of the configs[4] sweep (see cold())

def pmc_traffic_bytes():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of the headline workload
    (profiles/rNN_pmc.json): (2*FETCH_SIZE + WRITE_SIZE) * 1024 — the fallback when the run cannot measure them itself."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")) if re.fullmatch(r"r\d+_pmc\.json", os.path.basename(f)))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        d = d.get("k_iterate_split") or d.get("k_iterate_lean") or d["k_iterate_binned"]
        return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def measure_traffic_live(kernel: str, timeout_s: float = 90.0):
    """HBM bytes per launch of `kernel`, MEASURED by this run the way MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE
    and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass; --kernel-trace only, no other trace domain),
    each over a short child run of this very bench (3 timed steps), unit KiB, FETCH_SIZE doubled (on gfx950 it tallies the
    128-byte fabric requests at 64 B). Returns (bytes per launch, description) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="sar_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                   BENCH, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pipeline", "--sustained-seconds", "0",
                   "--no-traffic"]
            env = dict(os.environ, TMPDIR="/tmp")
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            got = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        got.append(float(r["Counter_Value"]))
            if not got:
                return None, f"no {counter} rows for {kernel}"
            vals[counter] = sum(got) / len(got)
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, (
            f"measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in two separate passes over a 3-step child run, "
            f"mean per launch of {kernel}, (2*FETCH_SIZE + WRITE_SIZE)*1024 with the guide's gfx950 read correction "
            f"(FETCH_SIZE {vals['FETCH_SIZE'] * 1024 / 1e9:.3f} GB uncorrected — an upper estimate for this kernel's scattered 4-byte reads — "
            f"+ WRITE_SIZE {vals['WRITE_SIZE'] * 1024 / 1e9:.3f} GB)")
    except Exception as e:  # a missing tool, a time-out, a changed csv: the committed passes stand in
        return None, f"live PMC passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(seconds_hint: float, WIDTH: int, HEIGHT: int, ITERS_PER_GPU: int):
    """The oracle's render_parallel-shaped port (threads + private buffers + serial merge + serial
    colorize) on this box's host cores. Reported beside the GPU number, never part of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cfg = O.poisson_saturne()
    cfg.width, cfg.height, cfg.transparent = WIDTH, HEIGHT, 0
    # the full C2 frame when the host gets through it in the time hint (~2e7 it/s/thread), else a cut
    est_rate = 2.0e7 * threads
    iters = ITERS_PER_GPU if ITERS_PER_GPU / est_rate <= seconds_hint else int(est_rate * seconds_hint)
    cfg.iterations = iters
    # The reference runs one worker per hardware thread (available_parallelism, src/lib.rs:920-922) and merges their
    # private buffer sets serially (:1070-1076): on a many-core host that merge dominates and FEWER threads are faster.
    # `value` is the best thread count of a short sweep (the most favourable number for the CPU); the reference's
    # own default (all threads) is reported next to it.
    runs = []
    for t in sorted({min(16, threads), min(32, threads), min(64, threads), threads}):
        secs, done, _ = O.render_parallel(cfg, t, 12, 1, want_image=True)
        runs.append({"cores": t, "value": done / secs, "seconds": round(secs, 2), "iterations": done})
    best = max(runs, key=lambda r: r["value"])
    allt = next(r for r in runs if r["cores"] == threads)
    # `--single-thread` semantics (render + colorize on one core, src/bin/main.rs:483-490) on a shorter sample
    import time as _t
    import numpy as _np
    st_iters = 100_000_000
    rt = O.Runtime(WIDTH, HEIGHT)
    t0 = _t.perf_counter()
    O.render(cfg, rt, _np.array([0.05, 0.031, 0.077]), st_iters)
    O.colorize(cfg, rt)
    st_secs = _t.perf_counter() - t0
    return {
        "value": best["value"], "unit": "iterations/s", "cores": best["cores"], "kind": "port",
        "sample": f"poisson-saturne {WIDTH}x{HEIGHT}, {best['iterations']} iterations, {best['cores']} threads x 12 "
                  f"jobs/thread, private buffers + serial merge + serial colorize ({best['seconds']} s); best of the "
                  f"thread counts {[r['cores'] for r in runs]}; C restatement of the reference (clang -O3 "
                  "-ffp-contract=off), not rustc output",
        "thread_sweep": [{"cores": r["cores"], "value": r["value"], "seconds": r["seconds"]} for r in runs],
        "all_hardware_threads": {"cores": allt["cores"], "value": allt["value"], "unit": "iterations/s",
                                 "sample": f"the reference's default thread count; {allt['seconds']} s, dominated by "
                                           f"the serial merge of {allt['cores']} buffer sets"},
        "single_thread": {"value": st_iters / st_secs, "unit": "iterations/s",
                          "sample": f"one trajectory, {st_iters} iterations + colorize ({st_secs:.2f} s)"},
    }


def native_measure(S, torch, devices, config, steps, warmup, K):
    """The same frame through the C ABI alone: sar_renderer_new_multi over `devices` (one host thread + one stream per
    device, slices exchanged with hipMemcpyPeerAsync, colorized per slice into a pinned host image). A step is one
    sar_render_parallel call: reset, render, exchange, colorize, image in host memory — the next frame's start points are
    drawn meanwhile on one helper thread per device, uploaded from page-locked memory and announced."""
    g = len(devices)
    if config == "c4":
        width, total_jobs, iters = K["C4_SIZE"], K["C4_JOBS"], K["C4_ITERS"]
    else:
        width, total_jobs, iters = K["WIDTH"], K["DEFAULT_JOBS"] * g, K["ITERS_PER_GPU"] * g
    jpu = 8
    units = total_jobs // jpu
    n = iters // units // jpu
    cfg = S.Config.poisson_saturne(iterations=iters, width=width, height=width, transparent=0, seed=1)
    r = S.ParallelRenderer(devices=devices, units=units, seed=1)
    img = torch.empty((width, width, 4), dtype=torch.int16).pin_memory()
    phases = {"render_ms": 0.0, "exchange_ms": 0.0, "colorize_ms": 0.0, "host_ms_before_exchange": 0.0, "host_ms_enqueue": 0.0,
              "draw_ahead_ms": 0.0}
    for _ in range(warmup):
        S.render_parallel_into(r, cfg, jpu, img.data_ptr())
    t0 = time.perf_counter()
    for _ in range(steps):
        S.render_parallel_into(r, cfg, jpu, img.data_ptr())
        t = r.last_timing()
        for k in phases:
            phases[k] += t[k]
    el = time.perf_counter() - t0
    t = r.last_timing()
    r.shutdown()
    parity = None
    if config == "c4" and total_jobs == K["C4_JOBS"]:
        # the same frame as tests/golden holds it (c4_full_1e10: seed 3, the preset's transparent flag): the FIRST frame of a renderer
        # seeded with 3, rendered by these devices, exchanged, colorized — its merged buffers gathered into device 0's runtime
        try:
            import numpy as np
            cfg_g = S.Config.poisson_saturne(iterations=iters, width=width, height=width, seed=3)
            rg = S.ParallelRenderer(devices=devices, units=units, seed=3)
            img_g = S.render_parallel(rg, cfg_g, jpu)
            rm = rg.runtime()
            parity = frame_parity(S, "c4_full_1e10", rm.count(), rm.zbuf(), rm.steps(), img_g, rm.max())
            rg.shutdown()
            print(f"[parity] native c4_full_1e10 over {g} shard(s): {parity['result']}", file=sys.stderr)
        except Exception as e:  # evidence next to the number: never lose the line over it
            parity = {"result": "not checked", "error": repr(e)}
    return {"value": n * total_jobs * steps / el, "parity": parity, "unit": "iterations/s", "ms_per_step": el / steps * 1e3, "steps": steps,
            "scaling": "strong" if config == "c4" else "weak", "devices": list(devices), "jobs_total": total_jobs,
            "iterations_per_job": n, "image": f"{width}x{width}",
            "phase_ms_per_step_slowest_device": {k: v / steps for k, v in phases.items() if k.endswith("_ms") and not k.startswith(("host", "draw"))},
            "host_ms_per_step": {"between_render_and_exchange_enqueue": phases["host_ms_before_exchange"] / steps,
                                 "until_the_frame_is_enqueued": phases["host_ms_enqueue"] / steps,
                                 "next_frame_points_drawn_on_helper_threads": phases["draw_ahead_ms"] / steps},
            "exchange_bytes_per_device": int(t["exchange_bytes_per_device"]), "peer_access_failures": int(t["peer_access_failures"]),
            "note": "sar_render_parallel end to end, image in pinned host memory (PCIe and the host-side job list included)"}




def n1_same_node(S, torch, np, device, width, total_jobs, iters_total, steps, warmup=1, seed=1):
    """The N = 1 denominator of a strong-scaling entry, measured by THIS job on THIS node: rank 0 alone renders the whole frame
    (every job, no exchange) on its GPU — reset, render, colorize to RGBA16 in HBM, every frame announcing its successor as the
    timed loop of the line does — while the other ranks wait. (The pool's boxes differ by ~6 %: a figure from another box's
    profile is not a denominator.)"""
    n = int(iters_total) // total_jobs
    cfg = S.Config.poisson_saturne(iterations=n * total_jobs, width=width, height=width, jobs_total=total_jobs, transparent=0, seed=seed)
    starts = S.start_points(seed, 0, total_jobs)
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        rt = S.Runtime(cfg, device=device)
        rt.set_stream(stream.cuda_stream)
        rgba = torch.empty(width * width * 4, dtype=torch.int16, device=f"cuda:{device}")
        starts_dev = torch.from_numpy(np.ascontiguousarray(starts)).to(f"cuda:{device}")

        def step(more):
            rt.reset()
            S.render_job_range_device(cfg, rt, total_jobs, n, starts_dev.data_ptr())
            if more:
                S.prefetch_device(cfg, rt, total_jobs, n, starts_dev.data_ptr())
            S.colorize_device(cfg, rt, rgba.data_ptr())

        for k in range(warmup):
            step(k + 1 < warmup)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for k in range(steps):
            step(k + 1 < steps)
        torch.cuda.synchronize(device)
        el = time.perf_counter() - t0
        launch = rt.describe_last_launch()
        rt.close()
    return {"value": n * total_jobs * steps / el, "unit": "iterations/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup,
            "jobs_total": total_jobs, "launch": launch, "device": device,
            "note": "rank 0 alone, the whole frame on one GPU of this node, RGBA16 in HBM, the other ranks idle"}


# ---- frame checksums against the committed goldens --------------------------------------------------------------------

def fnv1a64(S, a) -> str:
    """FNV-1a (64 bit) of a numpy array's bytes through the library (sar_checksum_fnv1a64), as tests/golden freezes them."""
    import ctypes as C
    import numpy as np
    a = np.ascontiguousarray(a)
    out = C.c_uint64()
    st = S.load_library().sar_checksum_fnv1a64(a.ctypes.data_as(C.c_void_p), a.nbytes, C.byref(out))
    if st != 0:
        raise RuntimeError(f"sar_checksum_fnv1a64: status {st}")
    return f"{out.value:016x}"


def golden_case(name: str):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_checksums.json")))[name]
    except Exception:
        return None


def frame_parity(S, case: str, count, zbuf, steps, rgba, mx) -> dict:
    """The merged frame's buffers (numpy, whole image) against tests/golden/fullsize_checksums.json[case] — the checksums the
    CPU oracle's frame has (frozen by tests/golden/make_fullsize_checksums.py, held by tests/test_gpu_fullsize.py)."""
    import numpy as np
    g = golden_case(case)
    got = {"max": int(mx), "count_sum": int(np.asarray(count).sum(dtype=np.uint64)), "touched": int((np.asarray(count) > 0).sum()),
           "count_fnv": fnv1a64(S, count), "zbuf_fnv": fnv1a64(S, zbuf), "steps_fnv": fnv1a64(S, steps), "rgba_fnv": fnv1a64(S, rgba)}
    if g is None:
        return {"result": "no golden", "against": f"tests/golden/fullsize_checksums.json[{case}]", **got}
    differs = [k for k in got if got[k] != g.get(k)]
    return {"result": "equal" if not differs else "differs", "against": f"tests/golden/fullsize_checksums.json[{case}]",
            "differing_fields": differs, **got}


def gather_merged_frame(S, torch, dist, np, rt, ex, cfg, rank, world, backend):
    """After the sliced exchange every rank holds the merged frame inside its own slice: the slices of count / zbuf / steps and
    the gathered RGBA16, assembled on rank 0 (None elsewhere). With ex None (one rank, or the rooted exchange) rank 0 holds all."""
    cnt, z, st = rt.count().ravel(), rt.zbuf().ravel(), rt.steps().ravel()
    if ex is None:
        img = S.colorize(cfg, rt) if rank == 0 else None
        return (cnt, z, st, img, rt.max()) if rank == 0 else None
    img = ex.colorize(dist, dst=0)
    torch.cuda.synchronize()
    mx = rt.max()                                      # (global after the scalar all-reduce)
    sp, first, count = ex.slice_pixels, ex.first, ex.count

    def gather(a_np, dtype):
        mine = np.zeros(sp, dtype)
        mine[:count] = a_np[first:first + count]
        t = torch.from_numpy(mine.view(np.uint8))
        t = t.cuda() if backend == "nccl" else t
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, parts, dst=0)
        if rank != 0:
            return None
        return np.concatenate([p.cpu().numpy().view(dtype) for p in parts])[: a_np.size]

    out = [gather(cnt, np.uint32), gather(z, np.float32), gather(st, np.float64)]
    if rank != 0:
        return None
    w, h = rt.dims()
    return out[0], out[1], out[2], img.cpu().numpy().view(np.uint16).reshape(h, w, 4), mx


# ---- loops reported next to `value` --------------------------------------------------------------------------------------

def sustained_loop(torch, stream, step, per_step_s: float, steps: int, seconds: float, counted_per_frame: int) -> dict:
    """The same step loop for a few seconds, every frame timed by events on the launch stream: min / median / max per frame, so that
    clock droop under a seconds-long fp64 load is on record."""
    try:
        with torch.cuda.stream(stream):
            frames = int(min(4000, max(steps, seconds / per_step_s)))
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(frames + 1)]
            marks[0].record()
            t0s = time.perf_counter()
            for k in range(frames):
                step(more=k + 1 < frames)
                marks[k + 1].record()
            torch.cuda.synchronize()
            els = time.perf_counter() - t0s
        order = [marks[k].elapsed_time(marks[k + 1]) for k in range(frames)]
        ms = sorted(order)
        third = max(frames // 3, 1)
        return {"seconds": els, "frames": frames, "value": counted_per_frame * frames / els, "unit": "iterations/s",
                "ms_per_frame": {"min": ms[0], "median": ms[frames // 2], "max": ms[-1],
                                 "mean_first_third": sum(order[:third]) / third, "mean_last_third": sum(order[-third:]) / third},
                "note": "frame k's event-to-event time on the launch stream (the first frame runs its own warm-up, the others were announced)"}
    except Exception as e:
        return {"error": repr(e)}


def pipelined_loop(S, torch, cfg, device, jobs, n, starts_ptr, npix, steps, warmup, tuning) -> dict:
    """The same frames on two runtimes and two streams, alternating: frame k's tail (accumulate, fold, colorize) and frame k+1's
    head (reset, warm-up) may share the chip. Reported next to `value`, never as it."""
    try:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        rts, bufs = [], []
        for st in streams:
            with torch.cuda.stream(st):
                r2 = S.Runtime(cfg, device=device)
                r2.set_stream(st.cuda_stream)
                r2.set_tuning(**tuning)
                rts.append(r2)
                bufs.append(torch.empty(npix * 4, dtype=torch.int16, device="cuda"))

        def frame(i):
            r2, st = rts[i & 1], streams[i & 1]
            with torch.cuda.stream(st):
                r2.reset()
                S.render_job_range_device(cfg, r2, jobs, n, starts_ptr)
                S.colorize_device(cfg, r2, bufs[i & 1].data_ptr())

        for i in range(max(warmup, 2)):
            frame(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            frame(i)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t0
        for r2 in rts:
            r2.close()
        return {"value": n * jobs * steps / el2, "unit": "iterations/s", "ms_per_step": el2 / steps * 1e3, "streams": 2,
                "note": "two runtimes on two streams, frames alternating; every frame does the full work"}
    except Exception as e:  # an optional extra: never lose the bench line over it
        return {"error": repr(e)}


# ---- BASELINE configs[4]: the `sequence` sweep ------------------------------------------------------------------------------

def run_c5(a, S, torch, dist, world: int, rank: int, local_rank: int, jobs_given: bool):
    """`sequence --start 0 --end 360 --step 1` (src/bin/main.rs:107-176, 493-517), frame k -> rank k mod N, replicas only. A step is
    one frame: reset, render_parallel's job split with a fresh start-point stream per frame, colorize, RGB16 conversion on the device,
    read-back into host memory (the PNG encoder, which the CLI runs on other threads, is excluded). Frames go through the library in
    BATCHES (sar_render_jobs_batch). Two sweeps: with the read-back (`value`), and to RGBA16 in HBM (`rgba16_in_hbm`)."""
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames as sequence_frames
    if a.rt_opt:
        S.use_hooks_build()   # A/B options live in the hooks build (include/sar_test_hooks.h); the default sweep runs on the product
    frame_jobs = a.jobs if jobs_given else 65536
    units, jpt = frame_jobs // 4, 4
    scfg = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, transparent=0)
    per_job = scfg.iterations // units // jpt
    done = [0]

    def sink(k, name, img):
        done[0] += 1

    def measure(seq):
        """--steps frames per rank through `seq`, after an untimed sweep; the slowest rank's wall time."""
        def sweep(frames_per_rank):
            done[0] = 0
            todo = [f for f in sequence_frames(0.0, float(frames_per_rank * world), 1.0) if f[0] % world == rank]
            seq.run(todo, sink, zero_copy=True)   # the sink only counts: no copy of the page-locked image
            torch.cuda.synchronize()
            assert done[0] == frames_per_rank
        sweep(max(a.warmup, (a.lanes or 2) * max(a.batch, a.max_batch) * 2))
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        sweep(a.steps)
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        sizes = list(seq.frames_per_launch)
        launch = seq.groups[0][0].describe_last_launch() if seq.groups and seq.groups[0] else ""
        seq.close()
        return el, sizes, launch

    def cold(frames_total: int, reps: int, of_world: int, as_rank: int, **kind):
        """BASELINE configs[4] AS STATED: one `sequence --start 0 --end 360 --step 1` sweep, this rank's frames_total / N of them, from
        nothing: wall time from the construction of the SequenceRenderer (runtimes, streams, page-locked images, start points: all
        inside) to the last delivered frame. Every repetition builds everything anew; the best of `reps` is reported next to all."""
        todo = [f for f in sequence_frames(0.0, float(frames_total), 1.0) if f[0] % of_world == as_rank]
        runs = []
        for _ in range(reps):
            done[0] = 0
            # What the sweep BEFORE this one freed (10-20 GB of runtimes) is wiped by the driver in the background, on the copy
            # engines a sweep's read-backs use: a cold sweep started right behind a close() measured that — +45 ms on some boxes, +220
            # on others (`tools/cold_sweep.py --pause`, profiles/r06_cold_sweep.md) — and the reference's CLI renders ONE sweep per process.
            torch.cuda.synchronize()
            time.sleep(COLD_SETTLE_S)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            seq = SequenceRenderer(scfg, **kind, **common)
            seq.run(todo, sink, zero_copy=True)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert done[0] == len(todo)
            if world > 1:
                t = torch.tensor([el], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            runs.append({"ms": el * 1e3, "setup_ms": ((seq.first_enqueued_at or t0) - t0) * 1e3, "frames_per_launch": list(seq.frames_per_launch),
                         "runtimes": sum(len(g) for g in seq.groups), "host_images": len(seq.images)})
            seq.close()
        best = min(runs, key=lambda r: r["ms"])
        return {"frames_per_gpu": len(todo), "frames": frames_total, "settle_s_before_each_construction": COLD_SETTLE_S, "ms": best["ms"], "ms_per_frame_per_gpu": best["ms"] / max(len(todo), 1),
                "setup_ms": best["setup_ms"], "frames_per_launch": best["frames_per_launch"], "runtimes_built": best["runtimes"],
                "host_images_page_locked": best["host_images"], "all_ms": [round(r["ms"], 2) for r in runs]}

    # the runtimes and the page-locked images live as long as the CLI's sweep does: made once, outside the timed frames
    common = dict(units=units, jobs_per_thread=jpt, seed=4, device=local_rank, lanes=a.lanes, batch=a.batch, max_batch=a.max_batch,
                  options={o.split("=")[0]: int(o.split("=")[1]) for o in a.rt_opt})
    if a.c5_only == "hbm":
        elapsed, sizes, launch = float("nan"), [], ""
    else:
        elapsed, sizes, launch = measure(SequenceRenderer(scfg, image_format=S.SAR_FMT_RGB16, **common))
    # ... and the same sweep to what SURVEY 8(d)'s metric ends with: the colorized frame as RGBA16 in device memory
    slots = ((a.lanes or 2) + 1) * max(a.batch, a.max_batch) + 1
    hbm = [torch.empty(1800 * 2000 * 4, dtype=torch.int16, device="cuda") for _ in range(slots)]
    if a.c5_only == "readback":
        el_hbm, sizes_hbm = float("nan"), []
    else:
        el_hbm, sizes_hbm, launch_hbm = measure(SequenceRenderer(scfg, device_ring=[t.data_ptr() for t in hbm], ring=slots, **common))
        launch = launch or launch_hbm
    # ... and the config as BASELINE states it: ONE cold sweep of 360 frames (360 / N per GPU), everything built inside the clock
    cold_rec = None
    if a.cold_frames > 0 and a.c5_only is None:
        rb, ih = dict(image_format=S.SAR_FMT_RGB16), dict(device_ring=[t.data_ptr() for t in hbm], ring=slots)
        cold_rec = {"read_back": cold(a.cold_frames, 3, world, rank, **rb), "rgba16_in_hbm": cold(a.cold_frames, 3, world, rank, **ih)}
        if world == 1:
            # what ONE of eight GPUs does with that sweep — frames 0, 8, 16, ...: 45 of them. At 8 GPUs the config's time is this
            # figure (set-up and the first batch's latency, not the steady state); stated from the one GPU there is
            cold_rec["one_gpu_of_8"] = {"read_back": cold(a.cold_frames, 3, 8, 0, **rb), "rgba16_in_hbm": cold(a.cold_frames, 3, 8, 0, **ih)}
        for key, steady in (("read_back", elapsed / a.steps * 1e3), ("rgba16_in_hbm", el_hbm / a.steps * 1e3)):
            c = cold_rec[key]
            c["steady_state_ms_per_frame"] = steady
            c["over_steady_state"] = c["ms"] / (c["frames_per_gpu"] * steady) if steady == steady and c["frames_per_gpu"] else None
    if rank != 0:
        return None
    parity = None
    if a.parity:
        # (untimed) the line proves itself: frames 32..39 of the sweep as ONE batched launch, frame 37's buffers and RGBA16 against
        # the oracle's frozen checksums of that frame (tests/golden/fullsize_checksums.json[c5_frame37_65536], held by
        # tests/test_gpu_fullsize.py); a --jobs other than 65 536 has no golden
        import numpy as np
        ring = [torch.empty(1800 * 2000 * 4, dtype=torch.int16, device="cuda") for _ in range(16)]
        proof = dict(common, lanes=1, batch=8, max_batch=8)
        with SequenceRenderer(scfg, device_ring=[t.data_ptr() for t in ring], ring=16, **proof) as seq:
            seq.run([f for f in sequence_frames(0.0, 40.0, 1.0) if f[0] >= 32])
            torch.cuda.synchronize()
            rt37 = seq.groups[0][5]
            img = ring[5].cpu().numpy().view(np.uint16)
            parity = frame_parity(S, "c5_frame37_65536" if frame_jobs == 65536 else f"c5_frame37_{frame_jobs}", rt37.count(), rt37.zbuf(),
                                  rt37.steps(), img, rt37.max())
            parity["frames_in_the_launch"] = list(seq.frames_per_launch)
    frames = a.steps * world
    counted = per_job * units * jpt * frames
    return {
        "metric": "attractor iterations/sec over the solar-sail sequence sweep (1e8 iterations per frame, 1800x2000), one frame per GPU",
        "value": counted / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "ms_per_frame_per_gpu": elapsed / a.steps * 1e3, "frames_per_second": frames / elapsed,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "parity": parity,
        "cold_sweep": cold_rec,
        "rgba16_in_hbm": {"value": counted / el_hbm, "unit": "iterations/s", "ms_per_frame_per_gpu": el_hbm / a.steps * 1e3,
                          "frames_per_second": frames / el_hbm,
                          "note": "the same sweep with every frame left as RGBA16 in device memory (colorize, no conversion, "
                                  "no read-back): what SURVEY 8(d)'s metric ends with"},
        "config": {"workload": "BASELINE configs[4]: sequence --start 0 --end 360 --step 1 (the first steps*N frames), solar-sail, "
                               "1e8 iterations per frame, 1800x2000, scale 1, frame k on rank k mod N; RGB16 conversion on the "
                               "device + read-back included, PNG encoder excluded",
                   "jobs_per_frame": units * jpt, "iterations_per_job": per_job, "frames": frames,
                   "lanes_per_gpu": {"read_back": a.lanes or 1, "in_hbm": a.lanes or 2}, "frames_per_launch": {str(f): sizes.count(f) for f in sorted(set(sizes))},
                   "frames_per_launch_in_hbm": {str(f): sizes_hbm.count(f) for f in sorted(set(sizes_hbm))},
                   "launch": launch,
                   "counted_over_executed_iterations": round(per_job / (per_job + 1000.0), 4),
                   "parallelism": f"{world} replica(s), no collective"}}
