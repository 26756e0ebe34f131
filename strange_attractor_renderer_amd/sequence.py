"""The `sequence` subcommand's semantics (reference src/bin/main.rs:107-176, 457-517) for BASELINE configs[4]:
an angle sweep rendered one frame per GPU. SURVEY.md section 8(f)-1.

`angle_iter` restates `AngleIter` exactly, quirks included:
  * frames while `curr + step/2 < end` (so 0/360/1 gives 360 frames, 0/360/0.5 gives 720);
  * degrees are converted to radians ONLY in that branch; the single-image fallback (`end <= start`, i.e. the
    plain `--angle` path, which builds `AngleIter::new(angle, angle, 1., name)`) passes the value through
    unconverted — `-a 220` means 220 radians (main.rs:169-171);
  * frame files are `<stem><zero-padded index>` with `ceil(log10((end-start-step/2)/step))` digits, none when
    that count is <= 1 (main.rs:118-123); the extension is whatever the file name had.

`render_sequence` renders frame k on rank k % world with the semantics of one ParallelRenderer rendering the
frames in order: the runtime is reset per frame (src/lib.rs:950-951) and every frame draws fresh start points (the
reference's per-thread RNGs simply run on across frames). Frame k's points are the first `jobs` of the stream seeded
with `frame_seed(seed, k)`: a frame does not depend on the number of GPUs, and no rank has to skip through the
points of the frames it does not own (a continued stream costs 4e6 draws per frame at 8 GPUs: more than rendering).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterator

import numpy as np

from . import api


def angle_iter(start: float, end: float, step: float, file_name: str = "attractor") -> Iterator[tuple[float, str]]:
    count = (end - start - step / 2.0) / step
    # `count as usize`: saturating, NaN -> 0
    count_usize = 0 if not (count == count) or count <= 0 else int(min(count, 2.0**64 - 1))
    if count_usize <= 1:
        digits = 0
    else:
        digits = int(math.ceil(math.log10(count)))
    parent, base = os.path.split(file_name)
    stem, ext = os.path.splitext(base)
    if not stem:
        stem = "attractor"
    curr = start
    it = 0
    while True:
        if curr + step / 2.0 < end:
            v = curr
            curr += step
            name = stem + (f"{it:0>{digits}d}" if digits > 0 else "")
            path = os.path.join(parent, name + ext) if parent else name + ext
            it += 1
            yield v * math.pi / 180.0, path
        elif it == 0:
            it += 1
            yield curr, file_name
        else:
            return


def frames(start: float, end: float, step: float, file_name: str = "attractor") -> list[tuple[int, float, str]]:
    return [(k, a, f) for k, (a, f) in enumerate(angle_iter(start, end, step, file_name))]


def frame_seed(seed: int, k: int) -> int:
    """Seed of frame k's start-point stream (SplitMix64's increment keeps consecutive frames far apart)."""
    return (seed + 0x9E3779B97F4A7C15 * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def render_sequence(config: "api.Config", start: float, end: float, step: float, *, units: int = 0,
                    jobs_per_thread: int = 12, seed: int = 0, rank: int = 0, world: int = 1, device: int = 0,
                    file_name: str = "attractor", image_format: int | None = None,
                    sink: Callable[[int, str, np.ndarray], None] | None = None) -> list[tuple[int, str, np.ndarray]]:
    """Renders this rank's frames of the sweep (frame k belongs to rank k % world; no collective is needed).
    Returns [(frame index, file name, image)] unless `sink` consumes the frames. The image is RGBA16, or — with
    `image_format` (SAR_FMT_*) — the CLI's converted format, converted on the device before the read-back."""
    todo = [(k, a, f) for (k, a, f) in frames(start, end, step, file_name) if k % world == rank]
    out = []
    if not todo:
        return out
    from concurrent.futures import ThreadPoolExecutor
    renderer = api.ParallelRenderer(device=device, units=units, seed=seed)
    T = renderer.num_threads()
    total_jobs = T * jobs_per_thread
    per_job = config.iterations // T // jobs_per_thread          # src/lib.rs:1058
    rt = None
    # the next frame's start points (~1 ms of host time per 2e5 jobs: as long as a frame renders) are drawn on a helper
    # thread while the GPU works on the current frame (the ctypes call releases the GIL)
    pool = ThreadPoolExecutor(max_workers=1)
    draw = lambda k: api.start_points(frame_seed(seed, k), 0, total_jobs)  # noqa: E731
    pending = pool.submit(draw, todo[0][0])
    try:
        for n, (k, angle, name) in enumerate(todo):
            cfg = config.replace(angle=angle, jobs_total=total_jobs, iterations=per_job * total_jobs, seed=seed)
            if rt is None:
                rt = api.Runtime(cfg, device=device)
            rt.reset()                                            # :950-951
            starts = pending.result()
            if n + 1 < len(todo):
                pending = pool.submit(draw, todo[n + 1][0])
            api.render_jobs(cfg, rt, starts)
            img = api.colorize(cfg, rt) if image_format is None else api.colorize_format(cfg, rt, image_format)  # :1080
            if sink is not None:
                sink(k, name, img)
            else:
                out.append((k, name, img))
    finally:
        pool.shutdown(wait=True)
        if rt is not None:
            rt.close()
        renderer.shutdown()
    return out


def render_sequence_to_files(config: "api.Config", start: float, end: float, step: float, *, file_name: str = "attractor.png",
                             eight_bit: bool = False, pam: bool = False, bmp: bool = False, encoders: int = 4,
                             **kw) -> list[str]:
    """The `sequence` subcommand end to end for this rank's frames: render, convert by (config.transparent, 8bit) and
    encode, with the encoding of frame k overlapping the rendering of frame k+1 on `encoders` extra threads — what the
    reference CLI does with its writer threads (src/bin/main.rs:493-517). Returns the paths written, in frame order."""
    from concurrent.futures import ThreadPoolExecutor
    if (pam or bmp) and not eight_bit:
        raise ValueError("--pam / --bmp require --8bit (src/bin/main.rs:256-258)")
    kind = "pam" if pam else ("bmp" if bmp else "png")
    fmt = api.image_format(bool(config.c.transparent), eight_bit)
    pending = []

    def encode(image: np.ndarray, path: str) -> str:
        api.write_image(image, path, kind)
        return path

    with ThreadPoolExecutor(max_workers=max(1, encoders)) as pool:
        def sink(k: int, name: str, img: np.ndarray):  # img is already in the file's format (converted on the device)
            pending.append(pool.submit(encode, img, os.path.splitext(name)[0] + "." + kind))

        render_sequence(config, start, end, step, file_name=file_name, image_format=fmt, sink=sink, **kw)
        return [f.result() for f in pending]
