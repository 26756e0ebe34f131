"""The `sequence` angle iterator (reference src/bin/main.rs:107-176) and the frame-per-GPU driver."""
import math
import os

import numpy as np
import pytest

from strange_attractor_renderer_amd.sequence import angle_iter, frame_seed, frames


def test_default_sweeps_frame_counts_and_names():
    f1 = frames(0.0, 360.0, 1.0, "attractor")
    assert len(f1) == 360                                   # curr + 0.5 < 360  ->  curr = 0..359
    assert f1[0] == (0, 0.0, "attractor000") and f1[-1][2] == "attractor359"
    assert f1[90][1] == 90.0 * math.pi / 180.0              # degrees -> radians only in the sweep branch
    f2 = frames(0.0, 360.0, 0.5, "out/solar.png")           # the subcommand's default step
    assert len(f2) == 720 and f2[1][2] == "out/solar001.png" and f2[-1][2] == "out/solar719.png"
    assert f2[-1][1] == 359.5 * math.pi / 180.0
    f3 = frames(10.0, 13.0, 1.0, "a")                        # count = 2.5 -> 1 digit
    assert [n for _, _, n in f3] == ["a0", "a1", "a2"] and len(f3) == 3
    f4 = frames(0.0, 2.4, 1.0, "a")                          # count = 1.9 -> `as usize` 1 -> no digits: both frames "a"
    assert [n for _, _, n in f4] == ["a", "a"]
    assert len(frames(0.0, 1.5, 1.0, "a")) == 1                # 1 + 0.5 < 1.5 is false


def test_single_image_fallback_passes_angle_unconverted():
    # the plain --angle path builds AngleIter::new(angle, angle, 1., name): one frame, value NOT converted
    assert list(angle_iter(220.0, 220.0, 1.0, "attractor")) == [(220.0, "attractor")]
    assert list(angle_iter(5.0, 1.0, 1.0, "x")) == [(5.0, "x")]


def test_float_accumulation_matches_rust_loop():
    got = [a for a, _ in angle_iter(0.0, 1.0, 0.1, "f")]
    curr, want = 0.0, []
    while curr + 0.05 < 1.0:
        want.append(curr * math.pi / 180.0)
        curr += 0.1
    assert got == want and len(got) == 10


@pytest.mark.gpu
def test_sequence_frames_match_oracle_and_do_not_depend_on_world_size(sar, oracle, gpu):
    from strange_attractor_renderer_amd.sequence import render_sequence
    cfg = sar.Config.solar_sail(iterations=400_000, width=180, height=200, scale=1.0, transparent=0,
                                render_kind=sar.SAR_RENDER_DEPTH)
    units, jpt, seed = 128, 2, 9
    whole = render_sequence(cfg, 0.0, 4.0, 1.0, units=units, jobs_per_thread=jpt, seed=seed)
    assert [k for k, _, _ in whole] == [0, 1, 2, 3]
    parts = {}
    for r in range(2):                                       # two "GPUs": frames 0,2 and 1,3
        for k, name, img in render_sequence(cfg, 0.0, 4.0, 1.0, units=units, jobs_per_thread=jpt, seed=seed,
                                            rank=r, world=2):
            parts[k] = img
    n = 400_000 // units // jpt
    for k, name, img in whole:
        assert name == f"attractor{k}"
        np.testing.assert_array_equal(img, parts[k])
        c = cfg.replace(angle=k * math.pi / 180.0)
        ort = oracle.Runtime(180, 200)
        oracle.render_jobs(c.c, ort, oracle.start_points(frame_seed(seed, k), 0, units * jpt), n)
        np.testing.assert_array_equal(img, oracle.colorize(c.c, ort))


@pytest.mark.gpu
@pytest.mark.parametrize("lanes,ring", [(1, 2), (1, 3), (2, 3), (2, 0), (3, 4)])
def test_sequence_sink_receives_every_frame_while_the_next_one_renders(sar, gpu, lanes, ring):
    """The read-back of frame k overlaps frame k+1 (sar_colorize_format_async into `ring` page-locked images): a sink sees
    the frames in order, each equal to the frame of the list-returning call, also when it holds the image back through a
    Future (the loop must not reuse a page-locked image before its consumer is done), and whatever the number of runtimes
    (`lanes`) the frames are rendered on in turn."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    from strange_attractor_renderer_amd.sequence import render_sequence
    cfg = sar.Config.solar_sail(iterations=300_000, width=160, height=120, scale=1.0, transparent=0)
    kw = dict(units=64, jobs_per_thread=2, seed=3, image_format=sar.SAR_FMT_RGB8)
    want = render_sequence(cfg, 0.0, 7.0, 1.0, lanes=1, **kw)
    assert len(want) == 7 and want[0][2].shape == (120, 160, 3) and want[0][2].dtype == np.uint8
    assert any(not np.array_equal(want[0][2], w[2]) for w in want[1:])
    got = []

    def slow_copy(k, img):
        time.sleep(0.02)                                     # the image must still be frame k's after the loop moved on
        got.append((k, np.array(img)))

    with ThreadPoolExecutor(max_workers=1) as pool:
        render_sequence(cfg, 0.0, 7.0, 1.0, sink=lambda k, name, img: pool.submit(slow_copy, k, img), ring=ring, lanes=lanes,
                        **kw)
    assert [k for k, _ in got] == list(range(7))
    for (k, img), (_, _, w) in zip(got, want):
        np.testing.assert_array_equal(img, w)
    with pytest.raises(ValueError):
        render_sequence(cfg, 0.0, 2.0, 1.0, ring=lanes, lanes=lanes, **kw)
    # one SequenceRenderer, several sweeps (what bench.py --config c5 does): the runtimes and host images are reused
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames
    with SequenceRenderer(cfg, units=64, jobs_per_thread=2, seed=3, image_format=sar.SAR_FMT_RGB8, lanes=lanes, ring=ring) as seq:
        todo = frames(0.0, 7.0, 1.0)
        first = seq.run(todo[:3])
        again = seq.run(todo)
    for (k, _, img), (_, _, w) in zip(first + again, want[:3] + want):
        np.testing.assert_array_equal(img, w)


@pytest.mark.gpu
def test_sequence_to_files_overlapped_encoding(sar, oracle, gpu, tmp_path):
    """render_sequence_to_files: frames rendered, converted and encoded on writer threads; every file decodes to the
    oracle's frame (start points continue across frames, runtime reset per frame)."""
    import image_decode as D
    from strange_attractor_renderer_amd.sequence import frames, render_sequence_to_files
    jobs_per_thread, units, n_total = 3, 64, 192 * 400
    cfg = sar.Config.solar_sail(iterations=n_total, width=96, height=80, scale=1.0, transparent=0)
    paths = render_sequence_to_files(cfg, 0.0, 40.0, 10.0, file_name=str(tmp_path / "f.png"), eight_bit=True, units=units,
                                     jobs_per_thread=jobs_per_thread, seed=9)
    fl = frames(0.0, 40.0, 10.0, str(tmp_path / "f.png"))
    assert [os.path.basename(p) for p in paths] == [os.path.basename(f) for (_, _, f) in fl] and len(paths) == 4
    total_jobs = units * jobs_per_thread
    per_job = n_total // units // jobs_per_thread
    for (k, angle, _), path in zip(fl, paths):
        c = cfg.replace(angle=angle, jobs_total=total_jobs, iterations=per_job * total_jobs)
        ort = oracle.Runtime(96, 80)
        oracle.render_jobs(c.c, ort, oracle.start_points(frame_seed(9, k), 0, total_jobs), per_job)
        want = oracle.convert(3, oracle.colorize(c.c, ort))
        np.testing.assert_array_equal(D.decode_png(path), want)
