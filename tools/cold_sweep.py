"""BASELINE configs[4] at the shape BASELINE states it: ONE cold `sequence --start 0 --end 360 --step 1` sweep (reference
src/bin/main.rs:107-176 frames, :493-517 loop), 360 / N frames per GPU — wall time from the construction of the SequenceRenderer
to the last delivered frame, with the host-side time of every kind of call on the way (where a cold sweep's setup goes).

python tools/cold_sweep.py [--frames 360] [--world 1] [--mode readback|hbm] [--reps 2] [--json out.json]

Each repetition builds everything anew (runtimes, page-locked images, streams); the process itself is warm after the first
(HIP context, code objects), which is why the first repetition is reported separately as `process_cold`."""
import argparse
import json
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=360)
    ap.add_argument("--world", type=int, default=1, help="frames / world frames are rendered (rank 0's share)")
    ap.add_argument("--mode", default="readback", choices=["readback", "hbm"])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lanes", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--max-batch", type=int, default=16)
    ap.add_argument("--jobs", type=int, default=65536)
    ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--skip", default="", help="TIMING EXPERIMENT ONLY (wrong images): comma list of per-frame calls to leave out: reset, colorize")
    ap.add_argument("--reuse-images", action="store_true", help="experiment: the host images of the first repetition serve the later ones "
                    "(runtimes are still built anew): is a cold sweep slower because its page-locked memory is new?")
    ap.add_argument("--reuse-runtimes", action="store_true", help="experiment: the frame groups of the first repetition serve the later ones")
    ap.add_argument("--pause", type=float, default=0.0, help="seconds between a repetition's close() and the next one's clock (the driver wipes freed "
                    "device memory in the background, on the copy engines the next sweep's read-backs use)")
    ap.add_argument("--share-copy-stream", action="store_true", help="experiment: every lane's read-backs on the FIRST lane's copy stream")
    ap.add_argument("--gpu-marks", action="store_true", help="HIP events per batch: start of its kernels, end of its conversions (launch stream), "
                    "start of its read-backs (copy stream) - printed for the middle of the last repetition")
    ap.add_argument("--no-pretouch", action="store_true", help="A/B: no sar_host_reserve (every image mapped, zeroed and locked at its first use)")
    ap.add_argument("--per-frame", action="store_true", help="A/B: reset and colorize as one call per frame instead of one per batch")
    ap.add_argument("--copy-cus", type=int, default=0, help="experiment: the lanes' read-back streams on this many CUs of their own (a multiple "
                    "of 8: bit i of a CU mask is CU i / 8 of XCD i %% 8), the launch streams on the others")
    ap.add_argument("--settle", type=int, default=-1, help="override sequence.SETTLE (stream synchronisations a lane's first frames are waited for with)")
    ap.add_argument("--keep", action="store_true", help="keep every repetition's renderer alive until the end (no reuse of just-freed memory)")
    ap.add_argument("--delivery", default="batch", choices=["frame", "batch"])
    ap.add_argument("--json", default="")
    ap.add_argument("--timeline", type=int, default=0, help="record the first N host calls of every repetition (start ms, duration ms, call)")
    a = ap.parse_args()

    import torch
    import strange_attractor_renderer_amd as S
    from strange_attractor_renderer_amd import api, sequence
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames as sequence_frames

    if a.no_pretouch:
        sequence.PRETOUCH = False
    if a.settle >= 0:
        sequence.SETTLE = a.settle
    acc: dict = {}
    timeline: list = []
    t_origin = [0.0]

    def timed(owner, name, label):
        fn = getattr(owner, name)

        def wrapper(*args, **kw):
            t = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                e = time.perf_counter()
                d = acc.setdefault(label, [0.0, 0])
                d[0] += e - t
                d[1] += 1
                if len(timeline) < a.timeline:
                    timeline.append([round((t - t_origin[0]) * 1e3, 3), round((e - t) * 1e3, 3), label])
        setattr(owner, name, wrapper)

    timed(api.Runtime, "__init__", "Runtime()")
    if hasattr(api.Runtime, "group"):
        inner_group = api.Runtime.group.__func__

        def timed_group(cls, *args, **kw):
            t = time.perf_counter()
            try:
                return inner_group(cls, *args, **kw)
            finally:
                e = time.perf_counter()
                d = acc.setdefault("Runtime.group()", [0.0, 0])
                d[0] += e - t
                d[1] += 1
                if len(timeline) < a.timeline:
                    timeline.append([round((t - t_origin[0]) * 1e3, 3), round((e - t) * 1e3, 3), "Runtime.group()"])
        api.Runtime.group = classmethod(timed_group)
    if a.copy_cus:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
        masked_streams = []

        def masked(bits_on):
            words = (C.c_uint32 * 8)(*([0] * 8))
            for b in bits_on:
                words[b // 32] |= 1 << (b % 32)
            st = C.c_void_p()
            assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words) == 0
            masked_streams.append(st)
            return st.value

        plain_group = api.Runtime.group.__func__

        def group_on_masked_streams(cls, *args, **kw):
            rts = plain_group(cls, *args, **kw)
            launch, copy = masked(range(a.copy_cus, 256)), masked(range(a.copy_cus))
            for rt in rts:
                rt.set_stream(launch)
                rt.set_copy_stream(copy)
            return rts
        api.Runtime.group = classmethod(group_on_masked_streams)
    if a.share_copy_stream:
        inner = api.Runtime.group.__func__
        first_copy = []

        def group_sharing(cls, *args, **kw):
            rts = inner(cls, *args, **kw)
            if not first_copy:
                first_copy.append(rts[0].copy_stream())
            else:
                for rt in rts:
                    rt.set_copy_stream(first_copy[0])
            return rts
        api.Runtime.group = classmethod(group_sharing)
    if "reset" in a.skip:
        api.reset_batch = lambda rts: None
    if "colorize" in a.skip:
        api.colorize_device_batch = lambda *args, **kw: None
    if a.per_frame:                   # the A/B of the batched reset / colorize launches: one call per frame as before
        def reset_each(rts):
            for rt in rts:
                rt.reset()

        def colorize_each(cfgs, rts, outs):
            for c, rt, o in zip(cfgs, rts, outs):
                api.colorize_device(c, rt, o)
        api.reset_batch, api.colorize_device_batch = reset_each, colorize_each
    marks: list = []
    if a.gpu_marks:
        ext = {}

        def ev_on(stream_ptr):
            st = ext.setdefault(stream_ptr, torch.cuda.ExternalStream(stream_ptr))
            e = torch.cuda.Event(enable_timing=True)
            e.record(st)
            return e
        inner_reset, inner_read = api.reset_batch, api.read_image_async

        def reset_marked(rts):
            marks.append({"lane_stream": rts[0].stream(), "start": ev_on(rts[0].stream()), "frames": len(rts)})
            inner_reset(rts)

        def read_marked(rt, image):
            m = marks[-1]
            if "kernels_done" not in m:
                m["kernels_done"] = ev_on(rt.stream())
                m["copies_start"] = ev_on(rt.copy_stream())
                m["copy_stream"] = rt.copy_stream()
            t = inner_read(rt, image)
            m["n_read"] = m.get("n_read", 0) + 1
            if m["n_read"] == m["frames"]:
                m["copies_enqueued_end"] = ev_on(rt.copy_stream())
            return t
        api.reset_batch, api.read_image_async = reset_marked, read_marked
    timed(api.Runtime, "reset", "reset")
    timed(api.Runtime, "synchronize", "synchronize")
    timed(api.Runtime, "close", "Runtime.close")
    timed(api.HostImage, "__init__", "HostImage()")
    timed(api.HostImage, "close", "HostImage.close")
    timed(api, "render_jobs_batch", "render_jobs_batch")
    timed(api, "render_jobs", "render_jobs")
    timed(api, "colorize_format_async", "colorize_format_async")
    for name in ("colorize_format_device", "read_image_async", "image_done"):
        if hasattr(api, name):
            timed(api, name, name)
    timed(api, "colorize_device", "colorize_device")
    timed(api, "colorize_device_batch", "colorize_device_batch")
    timed(api, "reset_batch", "reset_batch")
    timed(api, "wait_image", "wait_image")
    timed(api, "batch_frames", "batch_frames")
    timed(api.ParallelRenderer, "__init__", "ParallelRenderer()")

    units, jpt = a.jobs // 4, 4
    scfg = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, transparent=0)
    todo = [f for f in sequence_frames(0.0, float(a.frames), 1.0) if f[0] % a.world == 0]
    done = [0]

    def sink(k, name, img):
        done[0] += 1

    torch.cuda.init()
    torch.cuda.synchronize()
    reps = []
    kept = []
    stolen: dict = {}
    for rep in range(a.reps):
        acc.clear()
        del timeline[:]
        if a.share_copy_stream:
            del first_copy[:]
        done[0] = 0
        hbm = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_origin[0] = t0
        kw = dict(units=units, jobs_per_thread=jpt, seed=4, device=0, lanes=a.lanes, batch=a.batch, max_batch=a.max_batch, delivery=a.delivery)
        if a.mode == "hbm":
            slots = ((a.lanes or 2) + 1) * max(a.batch, a.max_batch) + 1
            hbm = [torch.empty(1800 * 2000 * 4, dtype=torch.int16, device="cuda") for _ in range(slots)]
            seq = SequenceRenderer(scfg, device_ring=[t.data_ptr() for t in hbm], ring=slots, **kw)
        else:
            seq = SequenceRenderer(scfg, image_format=S.SAR_FMT_RGB16, ring=a.ring, **kw)
        if a.reuse_images and stolen.get("images"):
            seq.images, seq.busy = stolen["images"], [None] * len(stolen["images"])
            seq.free_slots.extend(range(len(seq.images)))
        if a.reuse_runtimes and stolen.get("groups"):
            seq.groups = stolen["groups"]
        t1 = time.perf_counter()
        seq.run(todo, sink, zero_copy=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if a.gpu_marks and rep == a.reps - 1:
            base = marks[0]["start"]
            ms = lambda e: round(base.elapsed_time(e), 2)
            print("batch lane  kernels: start   end  (dur)   copies: start   end  (dur)   | next batch's kernels start")
            lanes_seen = sorted({m["lane_stream"] for m in marks})
            for i, m in enumerate(marks):
                if "copies_enqueued_end" not in m or not (len(marks) // 2 - 8 <= i < len(marks) // 2 + 8):
                    continue
                ks, ke, cs, ce = ms(m["start"]), ms(m["kernels_done"]), ms(m["copies_start"]), ms(m["copies_enqueued_end"])
                print(f"{i:5d} {lanes_seen.index(m['lane_stream']):4d}   {ks:9.2f} {ke:9.2f} ({ke - ks:6.2f})   {max(cs, ke):9.2f} {ce:9.2f} ({ce - max(cs, ke):6.2f})")
        del marks[:]
        sizes = list(seq.frames_per_launch)
        launch = seq.groups[0][0].describe_last_launch() if seq.groups and seq.groups[0] else ""
        n_rt = sum(len(g) for g in seq.groups)
        if a.reuse_images and a.mode != "hbm":
            stolen["images"], seq.images, seq.busy = seq.images, [], []
            seq.free_slots.clear()
        if a.reuse_runtimes:
            stolen["groups"], seq.groups = seq.groups, []
        if a.keep:
            kept.append((seq, hbm))   # freed memory that is handed out again has to be scrubbed by the driver first (0.2 s for 10 GB
        else:                         # on one box): with --keep every repetition gets memory nobody has used in this process
            seq.close()
        t3 = time.perf_counter()
        if a.pause:
            time.sleep(a.pause)
        assert done[0] == len(todo)
        reps.append({"rep": rep, "frames": len(todo), "construct_ms": (t1 - t0) * 1e3, "sweep_ms": (t2 - t1) * 1e3,
                     "cold_ms": (t2 - t0) * 1e3, "cold_ms_per_frame": (t2 - t0) * 1e3 / len(todo), "close_ms": (t3 - t2) * 1e3,
                     "frames_per_launch": sizes, "runtimes": n_rt, "launch": launch,
                     "host_ms": {k: [round(v[0] * 1e3, 3), v[1]] for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])}})
        if a.timeline:
            reps[-1]["timeline"] = list(timeline)
        print(json.dumps({k: v for k, v in reps[-1].items() if k != "timeline"}), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"args": vars(a), "reps": reps}, f, indent=1)


if __name__ == "__main__":
    main()
