#!/usr/bin/env python3
"""bench.py — attractor iterations/second of the MI355X iterate/accumulate path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(`python bench.py --gpus N` alone spawns the N ranks itself through torch.distributed.run.)

A "step" is one whole frame on every GPU: reset the runtime (as every render_parallel frame does, reference
src/lib.rs:950-951), iterate/accumulate this rank's trajectories (start points resident in HBM), then — for N>1 —
the one exchange step (all-to-all of the image slices, merge in rank order, 4-scalar all-reduce) and the colorize
(sharded for N>1, RGBA16 gathered on rank 0) to an RGBA16 image in device memory.

  --config c2 (default)  BASELINE.json configs[1]: poisson-saturne, 1e9 iterations PER GPU, 2048x2048. WEAK scaling:
                         every GPU renders its own 1e9 iterations of an (N x 1e9)-iteration frame.
  --config c4            BASELINE.json configs[3]: poisson-saturne, 1e10 iterations, 4096x4096, jobs_total = 1 048 576
                         sharded over the ranks (shard_jobs). STRONG scaling: the frame is the same at every N; at
                         N=1 the one GPU runs all the jobs in launch chunks.

With N > 1 and no --config the ONE command answers both scaling questions: `value` is the weak-scaling c2 line, and the
same ranks then render the c4 frame (strong scaling) — reported under `strong_c4` — and rank 0 drives the same two
workloads through the C-ABI-only multi-device renderer (sar_renderer_new_multi: host threads + hipMemcpyPeerAsync, no
torch.distributed) — reported under `native`.

    python bench.py --native --gpus N [--config c2|c4]   only the C-ABI multi-device renderer, one process

One JSON line on rank 0; see DESIGN.md "Measurement" for how each field is derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 2048
ITERS_PER_GPU = 1_000_000_000
DEFAULT_JOBS = 131072            # trajectories per GPU (2 waves per SIMD on 256 CUs); n = floor(1e9 / jobs)
# BASELINE configs[3]. SURVEY 8d sketched 524 288 jobs (65 536 per GPU at 8 GPUs) — but 65 536 trajectories are one per lane
# of the chip and no more, and the iterate kernel wants two per lane to hide its latencies (4096^2, 1.25e9 iterations: 14.2 ms
# with 65 536 jobs, 11.9 ms with 131 072). A strong-scaling frame must be the same frame at every N, so it is cut into 1 048 576 jobs (131 072 per GPU
# at 8): n = 9536 iterations per job. `--jobs 524288` reproduces SURVEY's split.
C4_SIZE, C4_ITERS, C4_JOBS = 4096, 10_000_000_000, 1048576
ALG_BYTES_PER_ITER = 12.0 + 12.0 * 0.0055   # count RMW 8 B + zbuf read 4 B + win-rate * 12 B
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_OPS_PER_ITER = 88           # unfused fp64 ops per counted iteration; FMA is not allowed
FP64_PEAK_OPS = 78.6e12 / 2      # MI355X vector fp64 78.6 TFLOP/s counts an FMA as 2 -> 39.3e12 unfused op/s


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
                   os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-pipeline", "--sustained-seconds", "0",
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


def cpu_baseline(seconds_hint: float):
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


def native_measure(S, torch, devices, config, steps, warmup):
    """The same frame through the C ABI alone: sar_renderer_new_multi over `devices` (one host thread + one stream per
    device, slices exchanged with hipMemcpyPeerAsync, colorized per slice into a pinned host image). A step is one
    sar_render_parallel call: reset, render, exchange, colorize, image in host memory — the next frame's start points are
    drawn meanwhile on one helper thread per device, uploaded from page-locked memory and announced."""
    g = len(devices)
    if config == "c4":
        width, total_jobs, iters = C4_SIZE, C4_JOBS, C4_ITERS
    else:
        width, total_jobs, iters = WIDTH, DEFAULT_JOBS * g, ITERS_PER_GPU * g
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
    return {"value": n * total_jobs * steps / el, "unit": "iterations/s", "ms_per_step": el / steps * 1e3, "steps": steps,
            "scaling": "strong" if config == "c4" else "weak", "devices": list(devices), "jobs_total": total_jobs,
            "iterations_per_job": n, "image": f"{width}x{width}",
            "phase_ms_per_step_slowest_device": {k: v / steps for k, v in phases.items() if k.endswith("_ms") and not k.startswith(("host", "draw"))},
            "host_ms_per_step": {"between_render_and_exchange_enqueue": phases["host_ms_before_exchange"] / steps,
                                 "until_the_frame_is_enqueued": phases["host_ms_enqueue"] / steps,
                                 "next_frame_points_drawn_on_helper_threads": phases["draw_ahead_ms"] / steps},
            "exchange_bytes_per_device": int(t["exchange_bytes_per_device"]), "peer_access_failures": int(t["peer_access_failures"]),
            "note": "sar_render_parallel end to end, image in pinned host memory (PCIe and the host-side job list included)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--jobs", type=int, default=None, help=f"trajectories per GPU (c2: {DEFAULT_JOBS}; c4: the frame's total, {C4_JOBS}; c5: per frame, 65536)")
    ap.add_argument("--iters", type=float, default=ITERS_PER_GPU, help="counted iterations per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl == RCCL; gloo only "
                    "to exercise the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--check", action="store_true", help="N>1: verify the merged count buffer against the sum of "
                    "the per-rank buffers (debug; not timed)")
    ap.add_argument("--host-starts", action="store_true", help="hand the start points over from host memory every step "
                    "(PCIe-inclusive rate; the default keeps them resident in HBM)")
    ap.add_argument("--lanes", type=int, default=2, help="--config c5: groups of runtimes (streams) per GPU the frames are rendered on in turn")
    ap.add_argument("--batch", type=int, default=0, help="--config c5: frames per set of launches (sar_render_jobs_batch); 0 = as many "
                    "as fill the chip (the library's advice, at most --max-batch), 1 = a frame per launch")
    ap.add_argument("--max-batch", type=int, default=16)
    ap.add_argument("--rt-opt", action="append", default=[], help="--config c5: name=value runtime option (sar_runtime_set_option) of every runtime (A/B)")
    ap.add_argument("--c5-only", default=None, choices=["readback", "hbm"], help="--config c5: only one of the two sweeps (profiling)")
    ap.add_argument("--config", default=None, choices=["c2", "c4", "c5"], help="c2: BASELINE configs[1], weak scaling (default); "
                    "c4: BASELINE configs[3] (1e10 iterations, 4096^2, 1048576 jobs sharded over the ranks), strong scaling. "
                    "c5: BASELINE configs[4], the solar-sail `sequence` sweep, 1e8 iterations per frame at 1800x2000, frame k on "
                    "rank k mod N, --steps frames per rank, no collective (replicas only). "
                    "With N > 1 and no --config: c2 as `value`, then c4 under `strong_c4` and both through the C ABI under `native`")
    ap.add_argument("--native", action="store_true", help="only the C-ABI multi-device renderer (sar_renderer_new_multi) over "
                    "--gpus devices in ONE process (device ordinals wrap around on a box with fewer GPUs)")
    ap.add_argument("--extras-seconds", type=float, default=240.0, help="N > 1 without --config: time the strong_c4 + native "
                    "extras may take before the line is printed without them")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="N=1: length of the extra sustained-rate loop (0 = skip)")
    ap.add_argument("--exchange", default="sliced", choices=["sliced", "rooted"], help="N>1: all-to-all of image slices + "
                    "sharded colorize (default) or all-reduce MAX + reduce SUM onto rank 0")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false", help="do not announce the next frame "
                    "(sar_runtime_prefetch_device): every frame runs its warm-up inside its own render call")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false", help="skip the two rocprofv3 --pmc child passes that measure "
                    "roofline.traffic (the child passes themselves run with it)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="skip the two-stream pipelined-throughput "
                    "measurement that is reported next to `value` at N=1")
    ap.add_argument("--variant", type=lambda s: int(s, 0), default=0)
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--stride", type=int, default=0)
    a = ap.parse_args()
    jobs_given = a.jobs is not None
    a.jobs = a.jobs if jobs_given else DEFAULT_JOBS
    both_curves = a.config is None and not a.native  # the driver's SCALE command: N > 1 and nothing else said
    a.config = a.config or "c2"

    if a.native:
        import torch
        import strange_attractor_renderer_amd as S
        ndev = max(S.device_count(), 1)
        res = native_measure(S, torch, [k % ndev for k in range(a.gpus)], a.config, a.steps, a.warmup)
        res.update({"metric": "attractor iterations/sec through sar_render_parallel (C ABI, multi-device)", "n_gpus": a.gpus,
                    "physical_gpus": ndev, "higher_is_better": True, "dtype": "f64", "data": "synthetic", "vs_baseline": None,
                    "config": {"workload": f"BASELINE configs[{1 if a.config == 'c2' else 3}] through sar_renderer_new_multi"}})
        print(json.dumps(res), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU) and relay their output
        import socket
        import subprocess
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.run(cmd).returncode)

    import numpy as np
    import torch
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd.distributed import SlicedExchange, exchange_merge, shard_jobs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if not torch.cuda.is_available() or S.device_count() <= 0:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    ndev = max(torch.cuda.device_count(), 1)
    if world > ndev and a.backend == "nccl":
        # more ranks than GPUs (a 1-GPU box exercising the N-rank path): RCCL refuses two ranks on one device
        a.backend = "gloo"
    local_rank %= ndev
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    if a.config == "c5":
        # BASELINE configs[4]: `sequence --start 0 --end 360 --step 1` (src/bin/main.rs:107-176, 493-517), frame k -> rank k mod N.
        # A step is one frame: reset, render_parallel's job split with a fresh start-point stream per frame, colorize, RGB16
        # conversion on the device, read-back into host memory. The PNG encoder (the CLI runs it on other threads) is excluded.
        from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames as sequence_frames
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
            sweep(max(a.warmup, a.lanes * seq.max_batch * 2))
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

        # the runtimes and the page-locked images live as long as the CLI's sweep does: made once, outside the timed frames
        common = dict(units=units, jobs_per_thread=jpt, seed=4, device=local_rank, lanes=a.lanes, batch=a.batch, max_batch=a.max_batch,
                      options={o.split("=")[0]: int(o.split("=")[1]) for o in a.rt_opt})
        if a.c5_only == "hbm":
            elapsed, sizes, launch = float("nan"), [], ""
        else:
            elapsed, sizes, launch = measure(SequenceRenderer(scfg, image_format=S.SAR_FMT_RGB16, **common))
        # ... and the same sweep to what SURVEY 8(d)'s metric ends with: the colorized frame as RGBA16 in device memory
        slots = (a.lanes + 1) * max(a.batch, a.max_batch) + 1
        hbm = [torch.empty(1800 * 2000 * 4, dtype=torch.int16, device="cuda") for _ in range(slots)]
        if a.c5_only == "readback":
            el_hbm, sizes_hbm = float("nan"), []
        else:
            el_hbm, sizes_hbm, launch_hbm = measure(SequenceRenderer(scfg, device_ring=[t.data_ptr() for t in hbm], ring=slots, **common))
            launch = launch or launch_hbm
        if rank == 0:
            frames = a.steps * world
            counted = per_job * units * jpt * frames
            print(json.dumps({
                "metric": "attractor iterations/sec over the solar-sail sequence sweep (1e8 iterations per frame, 1800x2000), one frame per GPU",
                "value": counted / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": elapsed / a.steps * 1e3, "ms_per_frame_per_gpu": elapsed / a.steps * 1e3, "frames_per_second": frames / elapsed,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "rgba16_in_hbm": {"value": counted / el_hbm, "unit": "iterations/s", "ms_per_frame_per_gpu": el_hbm / a.steps * 1e3,
                                  "frames_per_second": frames / el_hbm,
                                  "note": "the same sweep with every frame left as RGBA16 in device memory (colorize, no conversion, "
                                          "no read-back): what SURVEY 8(d)'s metric ends with"},
                "config": {"workload": "BASELINE configs[4]: sequence --start 0 --end 360 --step 1 (the first steps*N frames), solar-sail, "
                                       "1e8 iterations per frame, 1800x2000, scale 1, frame k on rank k mod N; RGB16 conversion on the "
                                       "device + read-back included, PNG encoder excluded",
                           "jobs_per_frame": units * jpt, "iterations_per_job": per_job, "frames": frames,
                           "lanes_per_gpu": a.lanes, "frames_per_launch": {str(f): sizes.count(f) for f in sorted(set(sizes))},
                           "frames_per_launch_in_hbm": {str(f): sizes_hbm.count(f) for f in sorted(set(sizes_hbm))},
                           "launch": launch,
                           "counted_over_executed_iterations": round(per_job / (per_job + 1000.0), 4),
                           "parallelism": f"{world} replica(s), no collective"}}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    def run_config(config, steps, warmup, extras):
        result = None
        if config == "c4":
            # STRONG scaling: the frame (1e10 iterations, 524 288 jobs, 4096^2) is the same at every N; rank r renders the
            # contiguous job slice shard_jobs gives it (src/lib.rs:1056-1062 split)
            width = height = C4_SIZE
            total_jobs = a.jobs if jobs_given else C4_JOBS
            n = int(a.iters if a.iters != ITERS_PER_GPU else C4_ITERS) // total_jobs
            first_job, jobs = shard_jobs(total_jobs, world, rank)
        else:
            # WEAK scaling: world*jobs trajectories of n iterations; this rank owns jobs [rank*jobs, (rank+1)*jobs)
            width = height = WIDTH
            jobs = a.jobs
            n = int(a.iters) // jobs
            total_jobs = jobs * world
            first_job = rank * jobs
        iters_gpu = n * jobs
        cfg = S.Config.poisson_saturne(iterations=n * total_jobs, width=width, height=height,
                                       jobs_total=total_jobs, transparent=0, seed=1)
        starts = S.start_points(1, first_job, jobs)

        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            rt = S.Runtime(cfg, device=local_rank)
            rt.set_stream(stream.cuda_stream)
            rt.enable_timing(True)
            rt.set_tuning(block_threads=a.block, checkpoint_stride=a.stride, variant=a.variant)
            npix = width * height
            rgba = torch.empty(npix * 4, dtype=torch.int16, device="cuda")
            ex = None
            if world > 1 and a.exchange == "sliced":
                ex = SlicedExchange(S, cfg, rt, rank, world, "cuda")
            elif world > 1:
                key = torch.empty(npix, dtype=torch.int64, device="cuda")
                sums = torch.empty(3 * npix, dtype=torch.int32, device="cuda")
            # per timed step: render end / exchange end / colorize end (read after the closing fence, never inside the region)
            evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(max(steps, warmup, 1))] if world > 1 else []
            exch_ms = [0.0, 0.0]
            step_no = [0]

            # the inputs of the path — the start points of this rank's trajectories — are resident in HBM before the
            # timed region starts (with host start points each frame uploads 3 MiB: +0.1 ms)
            starts_dev = torch.from_numpy(np.ascontiguousarray(starts)).cuda()

            def step(more=True):
                rt.reset()
                if a.host_starts:
                    S.render_job_range(cfg, rt, n, starts)
                else:
                    S.render_job_range_device(cfg, rt, jobs, n, starts_dev.data_ptr())
                    # the frame loop knows its next frame (src/bin/main.rs:493-517): announce it, so that its 1000 uncounted
                    # warm-up iterations per job run under THIS frame's accumulate / fold / colorize. Every frame still does
                    # all of its work inside the timed region; the last one announces nothing.
                    if a.prefetch and more:
                        S.prefetch_device(cfg, rt, jobs, n, starts_dev.data_ptr())
                if world > 1:
                    ev = evs[step_no[0] % len(evs)]
                    step_no[0] += 1
                    ev[0].record()
                    if ex is not None:
                        # Runtime::merge folded in rank order, sliced: all-to-all of the image slices (16 B/px), the owner
                        # folds its slice, 4 scalars all-reduced, every rank colorizes its slice, RGBA16 gathered on rank 0
                        ex.merge(dist)
                        ev[1].record()
                        ex.colorize(dist, dst=0)
                    else:
                        # rooted: depth keys -> all-reduce MAX; counts + winner's steps -> reduce SUM; rank 0 colorizes
                        exchange_merge(rt, rank, dist, key, sums, dst=0)
                        ev[1].record()
                        if rank == 0:
                            S.colorize_device(cfg, rt, rgba.data_ptr())
                    ev[2].record()
                else:
                    S.colorize_device(cfg, rt, rgba.data_ptr())

            def fence():
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize()

            for k in range(warmup):
                step(more=k + 1 < warmup)  # the last untimed frame announces nothing: no work of the timed region runs before it
            fence()
            if world > 1 and a.check:
                # each rank's own (un-merged) count summed over ranks must equal the merged count on rank 0
                def reduce_sum(a_np):  # int64 SUM onto rank 0, through whatever the backend can move
                    t = torch.from_numpy(np.ascontiguousarray(a_np))
                    t = t.cuda() if a.backend == "nccl" else t
                    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
                    return t.cpu().numpy()

                rt.reset()
                S.render_job_range(cfg, rt, n, starts)
                own = reduce_sum(rt.count().ravel().astype(np.int64))
                if ex is not None:
                    ex.merge(dist)
                    torch.cuda.synchronize()
                    mine = np.zeros(npix, np.int64)  # every rank holds the merged frame inside its own slice: assemble them
                    mine[ex.first:ex.first + ex.count] = rt.count().ravel()[ex.first:ex.first + ex.count]
                    merged = reduce_sum(mine)
                else:
                    exchange_merge(rt, rank, dist, key, sums, dst=0)
                    torch.cuda.synchronize()
                    merged = rt.count().ravel().astype(np.int64)
                if rank == 0:
                    assert np.array_equal(merged, own % (1 << 32)), "merged count != sum of rank counts"
                    assert int(merged.sum()) == n * total_jobs
                    print(f"[check] merged count over {world} ranks == sum of per-rank counts == {n * total_jobs}", file=sys.stderr)
                fence()
            # HIP events around every launch of the timed region, recorded on the launch stream by the library and
            # summed until they are read after the closing fence (reading them synchronises, so not inside the region)
            rt.set_option("timing_accumulate", 1)
            t0 = time.perf_counter()
            step_no[0] = 0
            for k in range(steps):
                step(more=k + 1 < steps)
            fence()
            elapsed = time.perf_counter() - t0
            for ev in evs[:steps]:
                exch_ms[0] += ev[0].elapsed_time(ev[1])
                exch_ms[1] += ev[1].elapsed_time(ev[2])
            tm = rt.last_timing()
            iter_ms, fold_ms, launches = tm.iterate_ms, tm.resolve_ms, tm.iterate_launches
            col_ms = tm.colorize_ms
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())

        # Sustained rate (reported NEXT to `value`): the same step loop for a few seconds, every frame timed by HIP events on
        # the launch stream — min / median / max per frame, so that clock droop under a seconds-long fp64 load is on record.
        sustained = None
        if world == 1 and extras and a.sustained_seconds > 0:
            try:
                rt.enable_timing(False)  # no per-launch events in this loop: thousands of frames
                with torch.cuda.stream(stream):
                    per = elapsed / steps
                    frames = int(min(4000, max(steps, a.sustained_seconds / per)))
                    marks = [torch.cuda.Event(enable_timing=True) for _ in range(frames + 1)]
                    marks[0].record()
                    t0s = time.perf_counter()
                    for k in range(frames):
                        step(more=k + 1 < frames)
                        marks[k + 1].record()
                    torch.cuda.synchronize()
                    els = time.perf_counter() - t0s
                ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(frames))
                third = max(frames // 3, 1)
                order = [marks[k].elapsed_time(marks[k + 1]) for k in range(frames)]
                sustained = {"seconds": els, "frames": frames, "value": n * total_jobs * frames / els, "unit": "iterations/s",
                             "ms_per_frame": {"min": ms[0], "median": ms[frames // 2], "max": ms[-1],
                                              "mean_first_third": sum(order[:third]) / third, "mean_last_third": sum(order[-third:]) / third},
                             "note": "frame k's event-to-event time on the launch stream (the first frame runs its own warm-up, "
                                     "the others were announced)"}
            except Exception as e:
                sustained = {"error": repr(e)}

        # Pipelined throughput (reported NEXT to `value`, never as it): the same frames on two runtimes and two streams,
        # alternating, so that frame k's tail (accumulate, fold, colorize — memory-bound) and frame k+1's head (reset, warm-up
        # — no LDS, arithmetic-bound) may share the chip. `value` above is the one-stream number: a frame's latency.
        pipelined = None
        if world == 1 and a.pipeline and extras:
            try:
                streams = [torch.cuda.Stream(), torch.cuda.Stream()]
                rts, bufs = [], []
                for st in streams:
                    with torch.cuda.stream(st):
                        r2 = S.Runtime(cfg, device=local_rank)
                        r2.set_stream(st.cuda_stream)
                        r2.set_tuning(block_threads=a.block, checkpoint_stride=a.stride, variant=a.variant)
                        rts.append(r2)
                        bufs.append(torch.empty(npix * 4, dtype=torch.int16, device="cuda"))

                def frame(i):
                    r2, st = rts[i & 1], streams[i & 1]
                    with torch.cuda.stream(st):
                        r2.reset()
                        S.render_job_range_device(cfg, r2, jobs, n, starts_dev.data_ptr())
                        S.colorize_device(cfg, r2, bufs[i & 1].data_ptr())

                for i in range(max(warmup, 2)):
                    frame(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    frame(i)
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t0
                pipelined = {"value": n * total_jobs * steps / el2, "unit": "iterations/s", "ms_per_step": el2 / steps * 1e3,
                             "streams": 2, "note": "two runtimes on two streams, frames alternating; every frame does the full work"}
                for r2 in rts:
                    r2.close()
            except Exception as e:  # an optional extra: never lose the bench line over it
                pipelined = {"error": repr(e)}

        counted = n * total_jobs * steps
        value = counted / elapsed
        if rank == 0:
            kern_s = iter_ms * 1e-3 / max(launches, 1)             # average duration of one launch of the iterate kernel
            launch_desc = rt.describe_last_launch()                # what the library really launched (not a guess made here)
            iterate_kernel = launch_desc.split(" ")[0]
            traffic, traffic_src = None, None
            if config == "c2" and int(a.iters) == ITERS_PER_GPU and jobs == DEFAULT_JOBS and world == 1 and a.traffic:
                traffic, traffic_src = measure_traffic_live(iterate_kernel)
                if traffic is None:
                    why = traffic_src
                    traffic, traffic_src = pmc_traffic_bytes()
                    if traffic_src:
                        traffic_src = (f"{traffic_src}: PMC passes of this workload committed with the round ((2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                                       f"launch), NOT measured by this run ({why})")
            per_launch = n * jobs * steps / max(launches, 1)      # counted iterations one launch processes
            ach = ALG_BYTES_PER_ITER * per_launch / kern_s / 1e9
            out = {
                "metric": ("attractor iterations/sec at 1e9 iters, 2048x2048 buffer (poisson-saturne), per-GPU frame" if config == "c2"
                           else "attractor iterations/sec at 1e10 iters, 4096x4096 buffer (poisson-saturne), whole frame"),
                "value": value, "unit": "iterations/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
                "scaling": "weak" if config == "c2" else "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": ("BASELINE configs[1]: poisson-saturne, 1e9 iterations per GPU, 2048x2048, "
                                        "Gas colorize to RGBA16 in HBM") if config == "c2" else
                                       (f"BASELINE configs[3]: poisson-saturne, 1e10 iterations, 4096x4096, {total_jobs} jobs "
                                        "sharded over the GPUs, Gas colorize to RGBA16 in HBM"),
                           "jobs_per_gpu": jobs, "jobs_total": total_jobs,
                           "iterations_per_job": n, "counted_iterations_per_step": n * total_jobs,
                           "warmup_iterations_per_job_uncounted": 1000,
                           "counted_over_executed_iterations": round(n / (n + 1000.0), 4),
                           "next_frame_announced": bool(a.prefetch and not a.host_starts),
                           "start_points": "uploaded from host memory every step" if a.host_starts else "resident in HBM",
                           "parallelism": f"trajectories sharded over {world} GPU(s)"
                                          + ((f"; all-to-all of image slices (16 B/px) + merge in rank order + sharded "
                                              f"colorize + RGBA16 gather (8 B/px)" if a.exchange == "sliced" else
                                              f"; all-reduce MAX (depth keys) + reduce SUM (count, steps)")
                                             + f" over {'RCCL/xGMI' if a.backend == 'nccl' else a.backend + ' (ranks share GPUs: staged through host)'}"
                                             if world > 1 else "")},
                "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS,
                             "traffic": traffic,
                             "traffic_source": traffic_src,
                             "kernel": iterate_kernel, "launch": launch_desc, "kernel_ms": kern_s * 1e3,
                             "alg_bytes_per_iteration": ALG_BYTES_PER_ITER,
                             "launches_timed": launches,
                             "valu_frac": FP64_OPS_PER_ITER * per_launch / kern_s / FP64_PEAK_OPS,
                             "note": "judged roofline per SURVEY 8(d) is HBM with 12.07 algorithmic B/iteration; the "
                                     "kernel's binding resource is fp64 VALU issue (88 unfused ops/iteration, no FMA "
                                     "allowed): valu_frac = 88*it/s / 39.3e12 op/s. traffic = HBM-side bytes per launch from the "
                                     "PMC counters (traffic_source), below the algorithmic bytes because the scatter state lives in LDS/L2"},
                "sustained": sustained,
                "kernel_ms_per_step": {"warmup_and_pack": tm.warmup_ms / steps, "iterate": iter_ms / steps,
                                       "accumulate_fold_resolve": fold_ms / steps,
                                       "colorize_last": col_ms},
            }
            if pipelined is not None:
                out["pipelined"] = pipelined
            if world > 1:
                out["exchange_ms_per_step"] = {"merge": exch_ms[0] / steps, "colorize_and_gather": exch_ms[1] / steps,
                                               "form": a.exchange, "backend": a.backend}
                if ex is not None:
                    out["exchange_ms_per_step"]["bytes_on_the_wire_per_rank"] = ex.bytes_on_the_wire()
            if world == 1 and not a.no_cpu_baseline and config == "c2" and extras:
                out["cpu_baseline"] = cpu_baseline(a.cpu_seconds)
                out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            result = out
        if ex is not None:
            del ex
        rt.close()
        return result

    out = run_config(a.config, a.steps, a.warmup, True)
    if world > 1 and both_curves:
        # The extras must never cost the line itself. They have never run on more than one physical GPU, so every rank arms
        # the same watchdog once the configs[1] measurement is done: if the extras are not through in time (a hang in a
        # collective, a dead peer), rank 0 prints the line it has — with what went wrong — and every rank leaves.
        import threading
        state = {"printed": False}
        lock = threading.Lock()

        def emit():
            with lock:
                if rank == 0 and not state["printed"]:
                    print(json.dumps(out), flush=True)
                state["printed"] = True

        def give_up():
            if rank == 0:
                out.setdefault("strong_c4", {"error": f"not finished within {a.extras_seconds:.0f} s: given up"})
                out.setdefault("native", {"error": f"not reached within {a.extras_seconds:.0f} s"})
            emit()
            sys.stdout.flush()
            os._exit(0)

        # ... and the line as it stands goes to stderr at once (flushed), so that even a kill from outside during the extras
        # leaves the configs[1] measurement on record; stdout keeps its ONE line, printed last.
        if rank == 0:
            print("[bench] weak-scaling line before the extras: " + json.dumps(out), file=sys.stderr, flush=True)
        dog = threading.Timer(a.extras_seconds, give_up)
        dog.daemon = True
        dist.barrier()
        dog.start()
        try:
            # the strong-scaling frame on the same ranks: BASELINE configs[3], the same frame at every N
            c4 = run_config("c4", max(2, min(a.steps, 6)), 1, False)
            if rank == 0:
                ref, ref_file = None, None
                try:  # the committed N=1 line of the same workload (the newest round's)
                    import glob
                    ref_file = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_c4_n1.json")))[-1]
                    ref = json.loads(open(ref_file).read().strip().splitlines()[-1])
                    ref_file = os.path.relpath(ref_file, ROOT)
                except Exception:
                    pass
                out["strong_c4"] = {"value": c4["value"], "unit": c4["unit"], "ms_per_step": c4["ms_per_step"], "steps": c4["steps"],
                                    "scaling": "strong", "workload": c4["config"]["workload"], "jobs_total": c4["config"]["jobs_total"],
                                    "exchange_ms_per_step": c4.get("exchange_ms_per_step"), "kernel_ms_per_step": c4["kernel_ms_per_step"],
                                    "launch": c4["roofline"]["kernel"],
                                    "n1_profile": ({"file": ref_file, "value": ref["value"], "ms_per_step": ref["ms_per_step"]}
                                                   if ref else None),
                                    "speedup_vs_n1_profile": (c4["value"] / ref["value"]) if ref else None}
        except Exception as e:
            if rank == 0:
                out["strong_c4"] = {"error": repr(e)}
        try:
            # ... and both workloads through the C ABI alone (one process, rank 0, every GPU of the job; the other ranks wait)
            dist.barrier()
            if rank == 0:
                try:
                    devs = list(range(min(world, max(torch.cuda.device_count(), 1))))
                    devs = [devs[k % len(devs)] for k in range(world)]
                    out["native"] = {"c2": native_measure(S, torch, devs, "c2", max(2, min(a.steps, 6)), 1),
                                     "c4": native_measure(S, torch, devs, "c4", max(2, min(a.steps, 4)), 1)}
                except Exception as e:  # an extra: never lose the line over it
                    out["native"] = {"error": repr(e)}
            dist.barrier()
        except Exception as e:
            if rank == 0:
                out.setdefault("native", {"error": repr(e)})
        dog.cancel()
        emit()
    elif rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
