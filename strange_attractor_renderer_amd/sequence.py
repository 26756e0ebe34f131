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
frames in order: the runtime is reset per frame (src/lib.rs:950-951) and the start-point stream continues across
frames (the reference's per-thread RNGs persist), so a frame does not depend on the number of GPUs.
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


def render_sequence(config: "api.Config", start: float, end: float, step: float, *, units: int = 0,
                    jobs_per_thread: int = 12, seed: int = 0, rank: int = 0, world: int = 1, device: int = 0,
                    file_name: str = "attractor",
                    sink: Callable[[int, str, np.ndarray], None] | None = None) -> list[tuple[int, str, np.ndarray]]:
    """Renders this rank's frames of the sweep (frame k belongs to rank k % world; no collective is needed).
    Returns [(frame index, file name, RGBA16 image)] unless `sink` consumes the frames."""
    todo = [(k, a, f) for (k, a, f) in frames(start, end, step, file_name) if k % world == rank]
    out = []
    if not todo:
        return out
    renderer = api.ParallelRenderer(device=device, units=units, seed=seed)
    T = renderer.num_threads()
    total_jobs = T * jobs_per_thread
    per_job = config.iterations // T // jobs_per_thread          # src/lib.rs:1058
    rt = None
    try:
        for k, angle, name in todo:
            cfg = config.replace(angle=angle, jobs_total=total_jobs, iterations=per_job * total_jobs, seed=seed)
            if rt is None:
                rt = api.Runtime(cfg, device=device)
            rt.reset()                                            # :950-951
            starts = api.start_points(seed, k * total_jobs, total_jobs)   # the renderer's stream, frame k
            api.render_jobs(cfg, rt, starts)
            img = api.colorize(cfg, rt)                           # :1080
            if sink is not None:
                sink(k, name, img)
            else:
                out.append((k, name, img))
    finally:
        if rt is not None:
            rt.close()
        renderer.shutdown()
    return out


def render_sequence_to_files(config: "api.Config", start: float, end: float, step: float, *, file_name: str = "attractor.png",
                             eight_bit: bool = False, pam: bool = False, bmp: bool = False, encoders: int = 2,
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
        def sink(k: int, name: str, rgba16: np.ndarray):
            # the same conversions as write_image_matches (:52-57), on the host copy the frame loop already made
            if fmt == api._abi.SAR_FMT_RGBA16:
                img = rgba16
            elif fmt == api._abi.SAR_FMT_RGB16:
                img = np.ascontiguousarray(rgba16[..., :3])
            else:
                img8 = ((rgba16.astype(np.uint32) + 128) // 257).astype(np.uint8)
                img = img8 if fmt == api._abi.SAR_FMT_RGBA8 else np.ascontiguousarray(img8[..., :3])
            path = os.path.splitext(name)[0] + "." + kind
            pending.append(pool.submit(encode, img, path))

        render_sequence(config, start, end, step, file_name=file_name, sink=sink, **kw)
        return [f.result() for f in pending]
