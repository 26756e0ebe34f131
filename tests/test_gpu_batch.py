"""GPU: F frames of a sweep through ONE set of launches (sar_render_jobs_batch, BASELINE configs[4]).

The reference's frame loop (src/bin/main.rs:493-517) resets (src/lib.rs:950-951) and renders frame after frame; the batched
launch lets F consecutive frames share the chip. Its contract is "frame i == sar_render_jobs(cfgs[i], rts[i], starts[i])":
every test here holds the batched frames to the per-frame renders AND to the CPU oracle, bit for bit (count, max, zbuf,
steps, image).
"""
import math

import numpy as np
import pytest

from strange_attractor_renderer_amd.sequence import frame_seed

# SAR_FUZZ_BASE=<k>: the seeded random tests of this file draw OTHER cases (tools/soak.sh runs a range of k on the GPU box)
_FUZZ_BASE = 1_000_003 * int(__import__("os").environ.get("SAR_FUZZ_BASE", "0"))

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def _frames(sar, preset, kind, n_frames, width, height, jobs, n, seed, first=0):
    cfgs, starts = [], []
    for k in range(first, first + n_frames):
        cfgs.append(getattr(sar.Config, preset)(iterations=jobs * n, width=width, height=height, jobs_total=jobs, render_kind=kind,
                                                scale=1.0, transparent=0, angle=k * math.pi / 180.0 * 7.0, seed=seed))
        starts.append(sar.start_points(frame_seed(seed, k), 0, jobs))
    return cfgs, starts


def _state(sar, cfg, rt):
    return rt.count(), rt.max(), rt.zbuf(), rt.steps(), sar.colorize(cfg, rt)


def _assert_same(a, b, what):
    assert np.array_equal(a[0], b[0]), f"{what}: count differs"
    assert a[1] == b[1], f"{what}: max differs"
    assert np.array_equal(_bits(a[2]), _bits(b[2])), f"{what}: zbuf differs"
    assert np.array_equal(_bits(a[3]), _bits(b[3])), f"{what}: steps differs"
    assert np.array_equal(a[4], b[4]), f"{what}: image differs"


def _oracle_state(oracle, cfg, starts, n, ort=None):
    ort = ort or oracle.Runtime(cfg.c.width, cfg.c.height)
    oracle.render_jobs(cfg.c, ort, starts, n)   # (copies: the arrays are views of the oracle runtime's memory)
    return ort, (ort.count.copy(), ort.max, ort.zbuf.copy(), ort.steps.copy(), oracle.colorize(cfg.c, ort))


@pytest.mark.parametrize("preset,kind", [("solar_sail", 0), ("solar_sail", 1), ("poisson_saturne", 0)])
@pytest.mark.parametrize("shared_stream,F,options", [(False, 6, {}), (True, 6, {"batch_warm": 2}), (True, 8, {"batch_warm": 2, "hint_bits": 16}),
                                                      (True, 4, {"batch_starts": 1}), (False, 2, {"batch_starts": 2, "batch_xcd": 1}),
                                                      (True, 16, {}), (True, 24, {"batch_warm": 2}), (True, 13, {}), (False, 3, {}),
                                                      (True, 11, {"batch_warm": 2, "hint_bits": 16})])
def test_batched_sweep_equals_per_frame_renders_and_oracle(sar, oracle, gpu, preset, kind, shared_stream, F, options):
    """F frames in one set of launches — frames dealt to the XCDs (2, 4, 8: XCDs per frame; 16, 24: frames per XCD; 3, 6, 11, 13:
    eight equal runs of wave pairs, what a sweep's tail is) or every frame everywhere (batch_xcd 1), the warm-up in one or two
    phases, the start points read in place or copied, the runtimes on one stream or on their own."""
    W, H, jobs, n = 600, 500, 4096, 300
    cfgs, starts = _frames(sar, preset, kind, F, W, H, jobs, n, seed=11)
    rts = [sar.Runtime(c) for c in cfgs]
    if shared_stream:
        for rt in rts[1:]:
            rt.share_streams(rts[0])
    for k, v in options.items():
        rts[0].set_option(k, v)
    sar.render_jobs_batch(cfgs, rts, starts)
    assert f"batch of {F} frames" in rts[0].describe_last_launch() and "k_iterate_split" in rts[F - 1].describe_last_launch()
    want_map = 0 if options.get("batch_xcd") == 1 else (1 if F in (2, 4, 8) else (2 if F % 8 == 0 else 3))
    assert f"(xcd map {want_map})" in rts[0].describe_last_launch(), rts[0].describe_last_launch()
    differ = 0
    for i, (cfg, rt, st) in enumerate(zip(cfgs, rts, starts)):
        got = _state(sar, cfg, rt)
        one = sar.Runtime(cfg)
        sar.render_jobs(cfg, one, st)
        assert "batch" not in one.describe_last_launch()
        _assert_same(got, _state(sar, cfg, one), f"frame {i} vs its own render call")
        one.close()
        _, want = _oracle_state(oracle, cfg, st, n)
        _assert_same(got, want, f"frame {i} vs the oracle")
        differ += int(i > 0 and not np.array_equal(got[0], first[0]))
        first = got if i == 0 else first
    assert differ == F - 1                                   # every frame has its own view and start points
    for rt in reversed(rts):
        rt.close()


def test_batches_accumulate_on_unreset_runtimes_and_after_reset_with_narrow_hints(sar, oracle, gpu):
    """`render` continues an un-reset runtime (src/lib.rs:742-744): a second batch on the same runtimes adds to the first; a
    reset starts over. With 16-bit depth hints the first warm-up after a reset measures the depth range (per frame)."""
    F, W, H, jobs, n = 3, 512, 512, 2048, 400
    cfgs, starts = _frames(sar, "poisson_saturne", 0, F, W, H, jobs, n, seed=5)
    cfgs2, starts2 = _frames(sar, "poisson_saturne", 0, F, W, H, jobs, n, seed=6, first=10)
    rts = [sar.Runtime(c) for c in cfgs]
    rts[0].set_option("hint_bits", 16)                      # the launch options are the leader's
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "hints=q16" in rts[0].describe_last_launch() and "batch of 3" in rts[2].describe_last_launch()
    sar.render_jobs_batch(cfgs2, rts, starts2)
    for i in range(F):
        ort, _ = _oracle_state(oracle, cfgs[i], starts[i], n)
        _, want = _oracle_state(oracle, cfgs2[i], starts2[i], n, ort)
        _assert_same(_state(sar, cfgs2[i], rts[i]), want, f"frame {i}, two batches on an un-reset runtime")
    for rt in rts:
        rt.reset()
    sar.render_jobs_batch(cfgs2, rts, starts2)
    for i in range(F):
        _, want = _oracle_state(oracle, cfgs2[i], starts2[i], n)
        _assert_same(_state(sar, cfgs2[i], rts[i]), want, f"frame {i} after a reset")
    for rt in reversed(rts):
        rt.close()


def test_frames_that_cannot_share_a_launch_run_one_after_the_other(sar, oracle, gpu):
    """Another job count, one frame, more frames than a table holds, a frame that needs several launch chunks: same results."""
    W, H, n = 256, 192, 200
    cfgs, starts = _frames(sar, "solar_sail", 0, 3, W, H, 1024, n, seed=2)
    odd = cfgs[1].replace(jobs_total=512, iterations=512 * n)
    cfgs[1], starts[1] = odd, starts[1][:512]
    rts = [sar.Runtime(c) for c in cfgs]
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch" not in rts[0].describe_last_launch()
    for i in range(3):
        _, want = _oracle_state(oracle, cfgs[i], starts[i], n)
        _assert_same(_state(sar, cfgs[i], rts[i]), want, f"mixed frame {i}")
    rts[0].reset()
    sar.render_jobs_batch(cfgs[:1], rts[:1], starts[:1])
    _assert_same(_state(sar, cfgs[0], rts[0]), _oracle_state(oracle, cfgs[0], starts[0], n)[1], "a batch of one")
    # a frame cut into launch chunks (test hook) is not batched
    for rt in rts:
        rt.reset()
    cfgs, starts = _frames(sar, "solar_sail", 0, 3, W, H, 1024, n, seed=2)
    rts[0].set_option("debug_chunk_jobs", 256)
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch" not in rts[0].describe_last_launch()
    for i in range(3):
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], n)[1], f"chunked frame {i}")
    for rt in rts:
        rt.close()
    # 35 frames: a table of 32 and a table of 3
    cfgs, starts = _frames(sar, "poisson_saturne", 0, 35, 128, 128, 256, 100, seed=8)
    rts = [sar.Runtime(c) for c in cfgs]
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch of 32 frames" in rts[0].describe_last_launch() and "batch of 3 frames" in rts[34].describe_last_launch()
    for i in (0, 7, 8, 15, 16, 31, 32, 34):
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], 100)[1], f"frame {i} of 35")
    with pytest.raises(sar.SarError):
        sar.render_jobs_batch(cfgs[:2], [rts[0], sar.Runtime(cfgs[0].replace(width=64, height=64))], starts[:2])
    for rt in rts:
        rt.close()


def test_drawn_start_points_follow_each_runtimes_stream(sar, oracle, gpu):
    """starts == None: frame i draws from rts[i]'s own stream, as sar_render_jobs does (src/lib.rs:748)."""
    F, jobs, n = 3, 512, 150
    cfgs = [sar.Config.poisson_saturne(iterations=jobs * n, width=200, height=160, jobs_total=jobs, seed=40 + i, angle=0.1 * i,
                                       transparent=0) for i in range(F)]
    rts = [sar.Runtime(c) for c in cfgs]
    sar.render_jobs_batch(cfgs, rts, None)
    for i in range(F):
        _, want = _oracle_state(oracle, cfgs[i], oracle.start_points(40 + i, 0, jobs), n)
        _assert_same(_state(sar, cfgs[i], rts[i]), want, f"frame {i}, drawn points")
    for rt in rts:
        rt.close()


def test_c5_frames_36_to_38_batched_at_full_size(sar, oracle, gpu):
    """BASELINE configs[4] at full size through the batched path: frames 36..38 of the solar-sail sweep in one set of launches;
    frame 37 is the committed `c5_frame37` case — oracle (threaded, identical bits) and frozen checksums."""
    import json
    import os
    import fullsize_cases as FC
    ocfg, starts37, n = FC.build_case("c5_frame37", oracle)
    jobs = starts37.shape[0]
    cfgs, starts = [], []
    for k in (36, 37, 38):
        c = sar.Config(oracle.copy_config(ocfg)).replace(angle=k * math.pi / 180.0)
        cfgs.append(c)
        starts.append(sar.start_points(frame_seed(0, k), 0, jobs))
    assert np.array_equal(starts[1], starts37)
    rts = [sar.Runtime(c) for c in cfgs]
    for rt in rts[1:]:
        rt.set_stream(rts[0].stream())
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch of 3 frames" in rts[1].describe_last_launch()
    cnt, mx, z, st, img = _state(sar, cfgs[1], rts[1])
    ort = oracle.Runtime(ocfg.width, ocfg.height)
    oracle.render_jobs_mt(ocfg, ort, starts37, n)
    _assert_same((cnt, mx, z, st, img), (ort.count, ort.max, ort.zbuf, ort.steps, oracle.colorize(ocfg, ort)), "c5 frame 37, batched")
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_checksums.json")))["c5_frame37"]
    got = {"max": mx, "count_sum": int(cnt.sum(dtype=np.uint64)), "touched": int((cnt > 0).sum()),
           "count_fnv": f"{oracle.fnv1a64(cnt):016x}", "zbuf_fnv": f"{oracle.fnv1a64(z):016x}",
           "steps_fnv": f"{oracle.fnv1a64(st):016x}", "rgba_fnv": f"{oracle.fnv1a64(img):016x}"}
    assert got == {k: g[k] for k in got}
    # the neighbours are other frames, and the library's advice for such a frame is more than one
    assert not np.array_equal(rts[0].count(), cnt) and not np.array_equal(rts[2].count(), cnt)
    assert 1 <= sar.batch_frames(cfgs[0], rts[0]) <= 16
    for rt in reversed(rts):
        rt.close()


@pytest.mark.parametrize("batch,lanes", [(0, 2), (3, 1), (2, 2), (1, 2)])
def test_sequence_sweep_in_batches_equals_frame_per_launch(sar, oracle, gpu, batch, lanes):
    """render_sequence with batches of frames per lane turn == the frame-per-launch sweep == the oracle, frame by frame."""
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames, render_sequence
    cfg = sar.Config.solar_sail(iterations=600_000, width=320, height=240, scale=1.0, transparent=0)
    kw = dict(units=256, jobs_per_thread=4, seed=21)
    want = render_sequence(cfg, 0.0, 11.0, 1.0, batch=1, lanes=1, **kw)
    got = render_sequence(cfg, 0.0, 11.0, 1.0, batch=batch, lanes=lanes, **kw)
    assert [k for k, _, _ in got] == list(range(11))
    for (k, name, img), (_, wname, w) in zip(got, want):
        assert name == wname
        np.testing.assert_array_equal(img, w)
    n = 600_000 // 256 // 4
    for k in (0, 5, 10):
        c = cfg.replace(angle=k * math.pi / 180.0, jobs_total=1024, iterations=n * 1024)
        _, o = _oracle_state(oracle, c, oracle.start_points(frame_seed(21, k), 0, 1024), n)
        np.testing.assert_array_equal(got[k][2], o[4])
    with SequenceRenderer(cfg, lanes=lanes, batch=batch, **kw) as seq:
        seq.run(frames(0.0, 11.0, 1.0))
        sizes = seq.frames_per_launch
    assert sum(sizes) == 11 and (batch == 0 or max(sizes) == batch)


@pytest.mark.parametrize("seed", range(10))
def test_seeded_random_batches_equal_the_oracle(sar, oracle, gpu, seed):
    """Ten seeded random batches away from the round numbers: odd image sizes, job counts that leave a wave partly filled, odd
    iteration counts (the last n % 2 iterations run as a short phase of the wave pairs), 2..12 frames, both presets and render
    kinds, views turned and scaled per batch, random A/B options — every frame bit for bit the oracle's."""
    rng = np.random.default_rng(1000 + seed + _FUZZ_BASE)
    preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
    kind = int(rng.integers(2))
    F = int(rng.integers(2, 13))
    W, H = int(rng.integers(97, 700)), int(rng.integers(83, 600))
    jobs = int(rng.integers(65, 3000))
    n = int(rng.integers(33, 500)) | int(rng.integers(2))          # odd half of the time
    scale = float(rng.choice([0.6, 1.0, 1.0, 1.7]))
    options = {}
    if rng.integers(2):
        options["batch_warm"] = int(rng.integers(1, 3))
    if rng.integers(3) == 0:
        options["hint_bits"] = int(rng.choice([16, 32]))
    if rng.integers(4) == 0:
        options["batch_xcd"] = 1
    if rng.integers(4) == 0:
        options["batch_starts"] = int(rng.integers(1, 4))
    cfgs, starts = [], []
    for k in range(F):
        cfgs.append(getattr(sar.Config, preset)(iterations=jobs * n, width=W, height=H, jobs_total=jobs, render_kind=kind, scale=scale,
                                                transparent=int(rng.integers(2)), angle=float(rng.uniform(0, 6.28)), seed=seed))
        starts.append(sar.start_points(frame_seed(seed, k), 0, jobs))
    rts = [sar.Runtime(c) for c in cfgs]
    if rng.integers(2):
        for rt in rts[1:]:
            rt.share_streams(rts[0])
    for k, v in options.items():
        rts[0].set_option(k, v)
    sar.render_jobs_batch(cfgs, rts, starts)
    assert f"batch of {F} frames" in rts[0].describe_last_launch(), (rts[0].describe_last_launch(), W, H, jobs, n)
    for i in range(F):
        _, want = _oracle_state(oracle, cfgs[i], starts[i], n)
        _assert_same(_state(sar, cfgs[i], rts[i]), want, f"seed {seed}: frame {i} of {F} ({preset}, {W}x{H}, {jobs} jobs x {n}, {options})")
    for rt in reversed(rts):
        rt.close()


def test_batch_after_an_announced_frame_nobody_rendered(sar, oracle, gpu):
    """An announced warm-up (sar_runtime_prefetch_device) writes the runtime's second set of warm-up buffers on its side stream; a
    two-phase batch uses that set for its first phase. The announcement is simply dropped, the batch is right, and the runtime
    still renders announced frames afterwards."""
    import torch
    F, W, H, jobs, n = 4, 400, 300, 3000, 201
    cfgs, starts = _frames(sar, "solar_sail", 0, F, W, H, jobs, n, seed=31)
    rts = [sar.Runtime(c) for c in cfgs]
    for rt in rts[1:]:
        rt.share_streams(rts[0])
    rts[0].set_option("batch_warm", 2)
    dev = [torch.from_numpy(np.ascontiguousarray(s)).cuda() for s in starts]
    torch.cuda.synchronize()
    for rt, c, d in zip(rts, cfgs, dev):
        sar.prefetch_device(c, rt, jobs, n, d.data_ptr())           # announced ... and never rendered
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch of 4 frames" in rts[0].describe_last_launch()
    for i in range(F):
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], n)[1], f"frame {i} after a dropped announcement")
    rt = rts[2]
    rt.reset()
    sar.prefetch_device(cfgs[2], rt, jobs, n, dev[2].data_ptr())
    sar.render_job_range_device(cfgs[2], rt, jobs, n, dev[2].data_ptr())
    assert "warmup_ahead=1" in rt.describe_last_launch()
    _assert_same(_state(sar, cfgs[2], rt), _oracle_state(oracle, cfgs[2], starts[2], n)[1], "an announced frame after the batch")
    for rt in reversed(rts):
        rt.close()


def test_a_45_frame_sweep_in_batches_of_16_16_13_equals_per_frame_renders_and_oracle(sar, oracle, gpu):
    """BASELINE configs[4] at 8 GPUs is 45 frames per GPU (src/bin/main.rs:107-176: 360 frames; frame k -> GPU k mod 8): batches of 16,
    16 and a tail of 13, the tail dealt to the XCDs like the full batches (xcd map 3), on the runtimes of ONE frame group — every
    frame bit for bit the frame-per-launch sweep's, and the oracle's."""
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames, render_sequence
    cfg = sar.Config.solar_sail(iterations=700_000, width=360, height=280, scale=1.0, transparent=0)
    kw = dict(units=320, jobs_per_thread=4, seed=33)
    todo = [f for f in frames(0.0, 360.0, 1.0) if f[0] % 8 == 3]                    # rank 3 of 8: frames 3, 11, ..., 355
    assert len(todo) == 45
    with SequenceRenderer(cfg, batch=16, **kw) as seq:
        got = seq.run(todo)
        assert seq.frames_per_launch == [16, 16, 13]
        assert len(seq.groups) == 1 and len(seq.groups[0]) == 16
        assert "batch of 13 frames (xcd map 3)" in seq.groups[0][0].describe_last_launch()
    with SequenceRenderer(cfg, batch=1, lanes=1, **kw) as seq:
        want = seq.run(todo)
        assert seq.frames_per_launch == [1] * 45 and seq.ring == 3
    assert [k for k, _, _ in got] == [k for k, _, _ in todo]
    for (k, name, img), (wk, wname, w) in zip(got, want):
        assert (k, name) == (wk, wname)
        np.testing.assert_array_equal(img, w)
    n, jobs = 700_000 // 320 // 4, 1280
    for i in (0, 15, 16, 31, 32, 38, 44):                                          # the edges of the three batches
        k = todo[i][0]
        c = cfg.replace(angle=k * math.pi / 180.0, jobs_total=jobs, iterations=n * jobs)
        _, o = _oracle_state(oracle, c, oracle.start_points(frame_seed(33, k), 0, jobs), n)
        np.testing.assert_array_equal(got[i][2], o[4])


def test_runtimes_of_a_frame_group_are_ordinary_runtimes(sar, oracle, gpu):
    """sar_runtime_new_group: one stream, one slab — and every member renders alone, is resized, reset and freed like any runtime
    (in any order; the group's allocations go with the last one)."""
    W, H, jobs, n = 300, 200, 700, 257
    cfgs, starts = _frames(sar, "solar_sail", 1, 5, W, H, jobs, n, seed=3)
    rts = sar.Runtime.group(cfgs[0], 5)
    assert len({rt.stream() for rt in rts}) == 1 and len({rt.copy_stream() for rt in rts}) == 1
    for i in (4, 0, 2):                                                           # alone, frame per launch
        sar.render_jobs(cfgs[i], rts[i], starts[i])
        assert "batch" not in rts[i].describe_last_launch()
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], n)[1], f"group member {i} alone")
    for rt in rts:
        rt.reset()
    sar.render_jobs_batch(cfgs, rts, starts)                                      # together
    assert "batch of 5 frames (xcd map 3)" in rts[0].describe_last_launch() and "/chip" in rts[0].describe_last_launch()
    for i in range(5):
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], n)[1], f"group member {i} in a batch")
    rts[1].close()                                                                # a member leaves; the others go on
    big = cfgs[3].replace(width=700, height=500)                                  # a member outgrows its share of the slab
    rts[3].set_width_height(700, 500)
    sar.render_jobs(big, rts[3], starts[3])
    _assert_same(_state(sar, big, rts[3]), _oracle_state(oracle, big, starts[3], n)[1], "group member 3, resized")
    rts[0].reset()
    sar.render_jobs(cfgs[0], rts[0], starts[0])
    _assert_same(_state(sar, cfgs[0], rts[0]), _oracle_state(oracle, cfgs[0], starts[0], n)[1], "group member 0 after a neighbour left")
    for i in (0, 3, 4, 2):
        rts[i].close()
    one = sar.Runtime(cfgs[2])                                                    # and the device still works afterwards
    sar.render_jobs(cfgs[2], one, starts[2])
    _assert_same(_state(sar, cfgs[2], one), _oracle_state(oracle, cfgs[2], starts[2], n)[1], "after the group")
    one.close()


def test_frames_that_cannot_share_launches_keep_one_runtime_per_lane(sar, oracle, gpu):
    """beyond 4 Mpx (bins of 65 536 pixels) frames do not share launches — sar_runtime_batch_frames answers 1, and the sweep
    keeps ONE runtime per lane, two lanes and a ring of four images (not sixteen full-size runtimes per lane rendered one after the
    other on one stream); the frames are the oracle's."""
    from strange_attractor_renderer_amd.sequence import SequenceRenderer, frames
    cfg = sar.Config.poisson_saturne(iterations=64 * 300, width=4096, height=1536, transparent=0)      # 6.3 Mpx
    kw = dict(units=16, jobs_per_thread=4, seed=9)
    assert sar.batch_frames(cfg.replace(jobs_total=64)) == 1
    with SequenceRenderer(cfg, image_format=sar.SAR_FMT_RGB8, **kw) as seq:
        got = seq.run(frames(0.0, 5.0, 1.0))
        assert (seq.lanes, seq.ring, seq.max_batch) == (2, 4, 1)
        assert [len(g) for g in seq.groups] == [1, 1] and seq.frames_per_launch == [1] * 5 and len(seq.images) <= 4
        assert "batch" not in seq.groups[0][0].describe_last_launch()
    n = 64 * 300 // 16 // 4
    for k in (0, 4):
        c = cfg.replace(angle=k * math.pi / 180.0, jobs_total=64, iterations=n * 64)
        _, o = _oracle_state(oracle, c, oracle.start_points(frame_seed(9, k), 0, 64), n)
        np.testing.assert_array_equal(got[k][2], oracle.convert(3, o[4]))


def test_a_member_with_another_hint_layout_renders_on_its_own(sar, oracle, gpu):
    """a batched frame's depth hints are laid out by the LEADER's options. A member whose own hints may have been written
    in another layout (its hint_tile option differs) is not batched — the frames run one after the other, same results."""
    F, W, H, jobs, n = 3, 512, 256, 2048, 300
    cfg_kw = dict(seed=23)
    cfgs, starts = _frames(sar, "poisson_saturne", 0, F, W, H, jobs, n, **cfg_kw)
    rts = [sar.Runtime(c) for c in cfgs]
    for rt in rts:
        rt.set_option("hint_bits", 16)
    rts[1].set_option("hint_tile", 1)                              # row-major hints on ONE member
    sar.render_jobs(cfgs[1], rts[1], starts[1])                    # ... which it has written in that layout
    sar.render_jobs_batch(cfgs, rts, starts)                       # (un-reset: frame 1 accumulates on top)
    assert "batch" not in rts[0].describe_last_launch() and "batch" not in rts[1].describe_last_launch()
    for i in range(F):
        ort, want = _oracle_state(oracle, cfgs[i], starts[i], n)
        if i == 1:
            _, want = _oracle_state(oracle, cfgs[i], starts[i], n, ort)
        _assert_same(_state(sar, cfgs[i], rts[i]), want, f"frame {i}, mixed hint layouts")
    rts[1].set_option("hint_tile", 0)                              # the same layout again: batched
    for rt in rts:
        rt.reset()
    sar.render_jobs_batch(cfgs, rts, starts)
    assert "batch of 3 frames" in rts[1].describe_last_launch()
    for i in range(F):
        _assert_same(_state(sar, cfgs[i], rts[i]), _oracle_state(oracle, cfgs[i], starts[i], n)[1], f"frame {i}, one layout")
    for rt in reversed(rts):
        rt.close()


@pytest.mark.parametrize("kind,options", [(0, {}), (0, {"hint_bits": 16}), (1, {})])
def test_reset_and_colorize_of_a_batch_in_one_launch_each(sar, oracle, gpu, kind, options):
    """sar_runtime_reset_batch / sar_colorize_device_batch: frame i is Runtime::reset (:684-695) / colorize (:1080) of runtime i —
    one launch for the members of a frame group, frame after frame for whoever does not fit (another stream, Depth, another palette)."""
    import torch
    W, H, jobs, n = 320, 240, 900, 301
    cfgs, starts = _frames(sar, "solar_sail", kind, 7, W, H, jobs, n, seed=11)
    cfgs[5] = cfgs[5].replace(brightness_factor=cfgs[5].c.brightness_factor * 1.5)   # a frame of other colours: a launch of its own
    rts = sar.Runtime.group(cfgs[0], 6) + [sar.Runtime(cfgs[6])]                  # six of a group and one on a stream of its own
    for rt in rts:
        for k, v in options.items():
            rt.set_option(k, v)
    outs = [torch.zeros(W * H * 4, dtype=torch.int16, device="cuda") for _ in range(7)]

    def images():
        sar.colorize_device_batch(cfgs, rts, [o.data_ptr() for o in outs])
        for rt in rts[5:]:
            rt.synchronize()
        return [o.cpu().numpy().view(np.uint16).reshape(H, W, 4) for o in outs]

    for round_ in range(2):                                                       # the second round renders on what reset_batch left
        sar.render_jobs_batch(cfgs[:6], rts[:6], starts[:6])
        sar.render_jobs(cfgs[6], rts[6], starts[6])
        got = images()
        for i in range(7):
            want = _oracle_state(oracle, cfgs[i], starts[i], n)[1]
            assert np.array_equal(got[i], want[4].reshape(H, W, 4)), f"round {round_}: image {i} of the batched colorize"
            _assert_same(_state(sar, cfgs[i], rts[i]), want, f"round {round_}: frame {i}")
        sar.reset_batch(rts)
        for i, rt in enumerate(rts):
            assert not rt.count().any() and rt.max() == 0 and not rt.steps().any(), f"round {round_}: runtime {i} after reset_batch"
            assert np.array_equal(_bits(rt.zbuf()), np.full(W * H, np.float32(-1.0)).view(np.uint32).reshape(rt.zbuf().shape))
    blank = images()                                                              # unvisited pixels only: the short path of every frame
    ort = oracle.Runtime(W, H)
    for i in range(7):
        assert np.array_equal(blank[i], oracle.colorize(cfgs[i].c, ort).reshape(H, W, 4)), f"blank image {i}"
    sar.reset_batch([])
    sar.colorize_device_batch([], [], [])
    with pytest.raises(sar.SarError):
        sar.colorize_device_batch(cfgs[:2], rts[:2], [outs[0].data_ptr(), 0])
    for rt in rts:
        rt.close()
