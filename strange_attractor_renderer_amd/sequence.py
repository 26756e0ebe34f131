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
import time
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


# Stream synchronisations a lane's first frames are waited for with (SequenceRenderer.run). Rounds 3-5 needed ONE: the kernels of two
# freshly created streams did not overlap until one of them had been synchronised while the other was busy. With a lane's streams
# made together (a frame group, round 6) the sweep runs the same without it — 1440 frames 1070 / 830 ms (read-back / in HBM) either
# way — and a cold 360-frame sweep is 10-20 ms shorter (307-331 against 327-358 ms): 0.
PRETOUCH = True   # the ring's host images are mapped and zeroed ahead by the library's helper threads (sar_host_reserve)
SETTLE = 0


class SequenceRenderer:
    """What a `sequence` sweep keeps between frames: the job split of one ParallelRenderer, `lanes` groups of runtimes (each
    a frame group of the library: one stream, one allocation per runtime) used in turn, and a ring of page-locked host images the
    converted frames are read back into.
    A group renders a BATCH of consecutive frames per turn — `batch` frames through one set of launches
    (sar_render_jobs_batch): 0 = as many as fill the chip (asked of the library before every batch, at most `max_batch`),
    1 = a frame per launch. One object can render several sweeps (`run`); `close` frees the device and page-locked memory."""

    def __init__(self, config: "api.Config", *, units: int = 0, jobs_per_thread: int = 12, seed: int = 0, device: int = 0,
                 image_format: int | None = None, ring: int = 0, lanes: int = 0, batch: int = 0, max_batch: int = 16,
                 device_ring: list | None = None, options: dict | None = None, delivery: str = "batch"):
        """device_ring: device pointers of width*height*8-byte buffers — the frames are then left there as RGBA16 (colorize
        only, src/lib.rs:841: what SURVEY 8(d)'s metric ends with) instead of being converted and read back; sinks receive None."""
        if lanes < 0:
            raise ValueError("lanes must be at least 1 (0: automatic)")
        if batch < 0 or max_batch < 1:
            raise ValueError("batch must be >= 0 and max_batch >= 1")
        if ring and ring < (lanes or 1) + 1:
            raise ValueError("ring must be at least lanes + 1")
        # lanes, ring and the batch size are settled by the first run (_setup): whether frames of this shape share launches at all
        # is the library's to say
        if delivery not in ("frame", "batch"):
            raise ValueError("delivery must be 'frame' or 'batch'")
        # "batch": a batch's frames are delivered together, once the next batch is enqueued (ring: lanes + 1 batches of images).
        # "frame": each frame right before the read-back that takes its place in the ring (ring: lanes batches + 1 — half the
        # page-locked memory; measured: 1.7 instead of 1.4 ms per frame on the box of the A/B, the read-backs start later)
        self.delivery = delivery
        self.want_lanes, self.want_ring, self.want_max_batch = lanes, ring, batch if batch else max_batch
        self.lanes = self.ring = self.max_batch = 0
        self.batch = batch
        self.device_ring = list(device_ring) if device_ring else None
        self.resync = bool(int(os.environ.get("SAR_SEQ_RESYNC", "0")))   # experiment
        self.options = dict(options or {})                         # runtime options (sar_runtime_set_option) of every runtime made here
        self.config, self.seed, self.device = config, seed, device
        self.fmt = api._abi.SAR_FMT_RGBA16 if image_format is None else image_format
        renderer = api.ParallelRenderer(device=device, units=units, seed=seed)
        try:
            T = renderer.num_threads()
        finally:
            renderer.shutdown()
        self.total_jobs = T * jobs_per_thread
        self.per_job = config.iterations // T // jobs_per_thread   # src/lib.rs:1058
        self.groups: list = []                                     # per lane: its runtimes (one frame group: one stream)
        self.settle: dict = {}                                     # stream synchronisations done per group (see run)
        self.images: list = []
        self.busy: list = []                                       # per host image: what its last consumer returned
        self.free_slots = __import__("collections").deque()        # ring slots no frame occupies
        self.slots_made = 0                                        # device-ring slots handed out so far
        self.frames_per_launch: list = []                          # statistic: the batch sizes of the last run
        self.reserved = False                                      # host blocks announced to the library (sar_host_reserve)
        self.first_enqueued_at = None                              # statistic: time.perf_counter() when the last run's first batch was enqueued

    def _setup(self, cfg: "api.Config", n_frames: int = 0):
        """Lanes, ring and batch size, once the shape of a frame is known. Frames that cannot share launches (beyond 4 Mpx, jobs of
        several launch chunks: the library answers 1) go frame by frame on two lanes with a ring of lanes + 2 images — ONE runtime
        per lane, not a batch of them. Otherwise: ONE lane of batches when the frames are read back (the read-back runs on the lane's
        copy stream under the next batch anyway, and a second lane's streams only crowd the process's few hardware queues: 0.73
        against 0.79 ms per frame of configs[4]), two when they stay in device memory (0.60 against 0.65)."""
        if self.lanes:
            return
        batching = self.want_max_batch > 1 and (self.batch > 1 or api.batch_frames(cfg, None) > 1)
        if not batching:
            self.lanes = self.want_lanes or 2
            self.max_batch = 1
            self.ring = self.want_ring or self.lanes + 2
        else:
            self.lanes = self.want_lanes or (2 if self.device_ring else 1)
            self.max_batch = self.want_max_batch
            # A short sweep (BASELINE configs[4] on eight GPUs: 45 frames each) is set-up and pipeline fill as much as rendering:
            # batches of a quarter of it — fewer runtimes to build and images to page-lock before the GPU has work, a shorter wait
            # for the first frame — beat full ones (45 frames, cold: 62 ms in batches of 12 against 80 in batches of 16; a 360-frame
            # sweep renders the same either way, so it keeps the size that fills the chip's rounds best)
            if not self.batch and 0 < n_frames < 4 * self.max_batch:
                self.max_batch = max(min(8, self.max_batch), -(-n_frames // 4))
            # a frame keeps its host image from the enqueue of its read-back until it is delivered, `lanes` batches later — frame by
            # frame, each delivery right before the read-back that takes its place in the ring (page-locking an image costs 1-3 ms:
            # a ring of two batches per lane was 100 ms of a cold sweep, and half of it never held two frames at once)
            per_lane = self.lanes + (1 if self.delivery == "batch" else 0)
            self.ring = self.want_ring or per_lane * self.max_batch + 1
            self.max_batch = max(1, min(self.max_batch, (self.ring - 1) // per_lane))
        if self.ring < self.lanes + 1:
            raise ValueError("ring must be at least lanes + 1")
        if self.device_ring is None and PRETOUCH:
            # the ring's images are page-locked one by one as the read-backs need them (`_image`); helper threads of the library
            # map and zero their pages meanwhile — no HIP call on them — so that locking one is 0.1 instead of 1-3 ms
            want = min(self.ring, n_frames or self.ring) - len(self.images)
            if want > 0:
                api.host_reserve(api.image_bytes(self.fmt, cfg.c.width, cfg.c.height), want)
                self.reserved = True

    def frame_config(self, angle: float) -> "api.Config":
        return self.config.replace(angle=angle, jobs_total=self.total_jobs, iterations=self.per_job * self.total_jobs,
                                   seed=self.seed)

    def _group(self, g: int, n: int, cfg: "api.Config") -> list:
        """The first n runtimes of lane g. A lane of batches is ONE frame group (sar_runtime_new_group): one stream, one read-back
        stream, one device and one page-locked allocation for all its runtimes — built when the lane is first used, lane 1's
        while lane 0's first batch renders."""
        while len(self.groups) <= g:
            # (the lanes' streams are made in lane order: the HIP runtime deals streams to its few hardware queues in the order
            # they are first used, and two lanes that share a queue do not overlap at all — 0.8 instead of 0.6 ms per frame)
            if self.max_batch > 1:
                grp = api.Runtime.group(cfg, self.max_batch, self.device)
            else:
                grp = [api.Runtime(cfg, device=self.device)]
            for rt in grp:
                for name, value in self.options.items():
                    rt.set_option(name, value)
            self.groups.append(grp)
        return self.groups[g][:n]

    def _batch_size(self, g: int, cfg: "api.Config", left: int) -> int:
        f = self.batch
        if f == 0 and self.max_batch > 1:
            grp = self.groups[g] if len(self.groups) > g else None
            # the library counts the wave pairs the chip holds against the jobs that survived the last launch's warm-up
            f = api.batch_frames(cfg, grp[0] if grp else None)
        f = max(1, min(f or 1, self.max_batch))
        return min(f, left)                                       # (any number of frames is dealt to the XCDs: a sweep's tail too)

    def _image(self, slot: int, cfg: "api.Config"):
        """Host image `slot`, page-locked at its first use, on this thread: a helper thread that locks pages ahead contends with
        every other HIP call of the process (measured: the sweep got slower), and the GPU is busy with the batch's render meanwhile."""
        while len(self.images) <= slot:
            self.images.append(api.HostImage(cfg.c.width, cfg.c.height, self.fmt))
            self.busy.append(None)
        return self.images[slot]

    def run(self, todo: list[tuple[int, float, str]],
            sink: Callable[[int, str, np.ndarray], object] | None = None, zero_copy: bool = False) -> list[tuple[int, str, np.ndarray]]:
        """Renders the frames [(index, angle, name)] in order; returns them unless `sink` consumes them. A sink receives its
        own copy of the frame unless `zero_copy` asks for the view of the recycled page-locked image (see render_sequence)."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        out: list = []
        self.frames_per_launch = []
        if not todo:
            return out
        # the next batch's start points (~1 ms of host time per 2e5 jobs: half as long as a frame renders) are drawn on helper
        # threads while the GPU works on the current one (the ctypes call releases the GIL)
        pool = ThreadPoolExecutor(max_workers=3)

        class _Drawn:
            def __init__(self, ks):
                self.futures = [pool.submit(api.start_points, frame_seed(seed_, k), 0, jobs_) for k in ks]

            def result(self):
                return [f.result() for f in self.futures]

        seed_, jobs_ = self.seed, self.total_jobs
        draw = _Drawn

        def deliver(g: int, rt, slot: int, ticket: int, k: int, name: str, first_of_batch: bool = False):
            if self.settle.get(g, 0) < (SETTLE if self.max_batch > 1 else 1):   # (lanes of single runtimes: streams made one by one)
                # Measured on ROCm 7.2, in a process that uses nothing but this library: the kernels of two freshly created
                # streams do not overlap — as if they shared a hardware queue — until one of them has been synchronised
                # ONCE while the other was busy (1.7 ms per frame for the first ~70 frames per runtime, 1.2 after; an event
                # wait does not do it, a synchronise of the idle streams neither; with three streams two of them stay
                # coupled until ~70 frames per runtime whatever is synchronised). So the first frame of every lane is
                # waited for with a stream synchronise: one bubble of a frame or two at the start of a sweep.
                rt.synchronize()
                self.settle[g] = self.settle.get(g, 0) + 1
            if self.device_ring is not None:
                if self.resync and first_of_batch:
                    rt.synchronize()
                if sink is not None:
                    sink(k, name, None)
                return
            api.wait_image(rt, ticket)
            if sink is not None:
                busy[slot] = sink(k, name, images[slot].array if zero_copy else np.array(images[slot].array))
            else:
                out.append((k, name, np.array(images[slot].array)))

        try:
            waiting = deque()                                     # frames enqueued and not yet delivered: (batch, lane, runtime, slot, ticket, k, name, first)
            pos, turn = 0, 0
            cfg0 = self.frame_config(todo[0][1])
            self._setup(cfg0, len(todo))
            images, busy, ring, lanes = self.images, self.busy, self.ring, self.lanes
            free = self.free_slots                                # host images (device-ring slots) no frame occupies, oldest release first
            size = self._batch_size(0, cfg0, len(todo))
            pending = draw([k for k, _, _ in todo[:size]])

            def deliver_one():
                fr = waiting.popleft()
                deliver(*fr[1:7], first_of_batch=fr[7])
                free.append(fr[3])

            def take_slot(cfg):
                """A slot for the next read-back: one that is free; else a NEW image while the ring may grow (page-locking one costs
                1-3 ms: only when nothing is free); else the oldest frame is waited for and delivered."""
                if not free and self.device_ring is None and len(images) < ring:
                    self._image(len(images), cfg)
                    return len(images) - 1
                if not free and self.device_ring is not None and self.slots_made < ring:
                    self.slots_made += 1
                    return self.slots_made - 1
                while not free:
                    deliver_one()
                return free.popleft()

            while pos < len(todo):
                g = turn % lanes
                part = todo[pos:pos + size]
                pos += len(part)
                cfgs = [self.frame_config(angle) for _, angle, _ in part]
                rts = self._group(g, len(part), cfgs[0])
                starts = pending.result()
                if pos < len(todo):                               # the next batch: its size is the next lane's to say
                    size = self._batch_size((turn + 1) % lanes, cfg0, len(todo) - pos)
                    pending = draw([k for k, _, _ in todo[pos:pos + size]])
                api.reset_batch(rts)                              # :950-951, the batch's frames in one launch
                if len(part) > 1:
                    api.render_jobs_batch(cfgs, rts, starts)
                else:
                    api.render_jobs(cfgs[0], rts[0], starts[0])
                if not self.frames_per_launch:
                    self.first_enqueued_at = time.perf_counter()  # statistic: everything before this was set-up
                self.frames_per_launch.append(len(part))
                if self.device_ring is None:
                    # the whole batch's colorize + conversion first (device memory only): the launch stream then never waits for a
                    # host image — only the copies do, on their own stream
                    for cfg, rt in zip(cfgs, rts):
                        api.colorize_format_device(cfg, rt, self.fmt)  # :1080
                else:
                    # RGBA16 left in device memory: the batch's slots first, then ONE colorize launch for all of its frames
                    slots = [take_slot(cfg) for cfg in cfgs]
                    api.colorize_device_batch(cfgs, rts, [self.device_ring[slot % len(self.device_ring)] for slot in slots])  # :1080
                    for i, ((k, _, name), rt, slot) in enumerate(zip(part, rts, slots)):
                        waiting.append((turn, g, rt, slot, 0, k, name, i == 0))
                for i, ((k, _, name), cfg, rt) in enumerate(zip(part, cfgs, rts) if self.device_ring is None else ()):
                    # frames whose read-back has completed are delivered as we go (no wait): their images take this batch's frames,
                    # and the ring only grows while the copies cannot keep up
                    while waiting and waiting[0][0] < turn and api.image_done(waiting[0][2], waiting[0][4]):
                        deliver_one()
                    if self.delivery == "frame" and waiting and waiting[0][0] <= turn - lanes:
                        deliver_one()
                    slot = take_slot(cfg)
                    if hasattr(busy[slot], "result"):             # the consumer of the frame that last used this image
                        busy[slot].result()
                    busy[slot] = None
                    ticket = api.read_image_async(rt, images[slot])
                    waiting.append((turn, g, rt, slot, ticket, k, name, i == 0))
                while waiting and waiting[0][0] <= turn - lanes:  # the batch `lanes` back, while the GPU is busy with the later ones
                    deliver_one()
                turn += 1
            while waiting:
                deliver_one()
            for i, b in enumerate(busy):
                if hasattr(b, "result"):
                    b.result()
                busy[i] = None
        finally:
            pool.shutdown(wait=True)
            for grp in self.groups:                               # also after an error: nothing may still write the images
                for rt in grp[:1]:
                    rt.synchronize()
        return out

    def close(self):
        for grp in self.groups:
            for rt in grp[:1]:
                rt.synchronize()
        for im in self.images:
            im.close()
        if self.reserved:
            api.host_reserve(0, 0)                                # what the sweep did not take
            self.reserved = False
        for grp in self.groups:
            for rt in reversed(grp):
                rt.close()
        self.groups, self.images, self.busy, self.settle = [], [], [], {}
        self.free_slots.clear()
        self.slots_made = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def render_sequence(config: "api.Config", start: float, end: float, step: float, *, units: int = 0,
                    jobs_per_thread: int = 12, seed: int = 0, rank: int = 0, world: int = 1, device: int = 0,
                    file_name: str = "attractor", image_format: int | None = None,
                    sink: Callable[[int, str, np.ndarray], object] | None = None,
                    ring: int = 0, lanes: int = 0, zero_copy: bool = False, batch: int = 0,
                    max_batch: int = 16) -> list[tuple[int, str, np.ndarray]]:
    """Renders this rank's frames of the sweep (frame k belongs to rank k % world; no collective is needed).
    Returns [(frame index, file name, image)] unless `sink` consumes the frames. The image is RGBA16, or — with
    `image_format` (SAR_FMT_*) — the CLI's converted format, converted on the device before the read-back.

    The loop is the reference CLI's (src/bin/main.rs:493-517: frame k goes to the writer threads, the renderer goes on
    with frame k+1) moved one step down: everything of frame k is only ENQUEUED — reset, start points, iterate, colorize,
    conversion, the copy into one of `ring` page-locked host images — on the stream of one of `lanes` runtimes, used in
    turn, and the frame `lanes` back is handed to `sink` while the GPU works on the later ones. Frames are independent
    (each has its own Runtime state, :950-951 resets it), so with two lanes the GPU fills one frame's latency-bound parts
    (the 1000-iteration warm-up of a few waves per SIMD, the kernel tails, the read-back on the copy engine) with the other
    frame's arithmetic. A sink receives its own copy of the frame (as every frame was a fresh array before the images were
    recycled). With `zero_copy` it receives a VIEW of one of the page-locked images instead, valid until the sink returns — or, when
    the sink returns an object with `.result()` (a Future of its consumer), until that has returned: the loop waits for it before
    it reuses the image. `batch` consecutive frames go through ONE set of launches (sar_render_jobs_batch; 0 = as many as fill the
    chip, at most `max_batch`; 1 = a frame per launch) — a frame of 65 536 jobs fills a third of an MI355X — and a lane's turn is
    then a batch. `ring` bounds the page-locked images (0 = (lanes + 1) * max_batch + 1; a frame per launch: lanes + 2): an image is
    page-locked only when no delivered frame's image is free, so a sweep whose read-backs keep up holds about one batch of them."""
    todo = [(k, a, f) for (k, a, f) in frames(start, end, step, file_name) if k % world == rank]
    if not todo:
        return []
    with SequenceRenderer(config, units=units, jobs_per_thread=jobs_per_thread, seed=seed, device=device,
                          image_format=image_format, ring=ring, lanes=lanes, batch=batch, max_batch=max_batch) as seq:
        return seq.run(todo, sink, zero_copy)


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
            f = pool.submit(encode, img, os.path.splitext(name)[0] + "." + kind)
            pending.append(f)
            return f                                  # the page-locked image is reused only after its file is written

        lanes_ = kw.get("lanes", 0) or 1
        kw.setdefault("ring", max(1, encoders) + (lanes_ + 1) * (kw.get("batch", 0) or kw.get("max_batch", 16)) + 1)
        render_sequence(config, start, end, step, file_name=file_name, image_format=fmt, sink=sink, zero_copy=True, **kw)  # the encoder's Future guards the view
        return [f.result() for f in pending]
