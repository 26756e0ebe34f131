"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against the
CPU oracle on the same explicit start points.

Bar: bit-exact for count / max (integer), bit-exact for zbuf and steps (the device executes the
reference's fp64 op sequence, no FMA), bit-exact RGBA16 for colorize wherever ln() comes from the
host-libm table (count+1 <= 2^20), <= 1 LSB beyond it (device log, <= 1 ulp)."""
import numpy as np
import pytest

# SAR_FUZZ_BASE=<k>: the seeded random tests of this file draw OTHER cases (tools/soak.sh runs a range of k on the GPU box)
_FUZZ_BASE = 1_000_003 * int(__import__("os").environ.get("SAR_FUZZ_BASE", "0"))

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def assert_state_equal(rt, ort, what=""):
    cnt = rt.count()
    np.testing.assert_array_equal(cnt, ort.count, err_msg=f"count {what}")
    assert rt.max() == ort.max, f"max {what}"
    np.testing.assert_array_equal(_bits(rt.zbuf()), _bits(ort.zbuf), err_msg=f"zbuf {what}")
    np.testing.assert_array_equal(_bits(rt.steps()), _bits(ort.steps), err_msg=f"steps {what}")


def _cfg(sar, preset, **kw):
    return getattr(sar.Config, preset)(**kw)


def test_c1_single_trajectory_matches_golden_and_oracle(sar, oracle, gpu):
    """BASELINE config 1: poisson-saturne, 1e7 iterations, 512x512, one trajectory (render semantics)."""
    import json, os
    cfg = _cfg(sar, "poisson_saturne", iterations=10_000_000, width=512, height=512, jobs_total=1)
    p0 = np.array([[0.05, 0.031, 0.077]])
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt, p0)
    ort = oracle.Runtime(512, 512)
    oracle.render(cfg.c, ort, p0[0], 10_000_000)
    assert_state_equal(rt, ort, "C1")
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))["c1_512"]
    assert f"{oracle.fnv1a64(rt.count()):016x}" == golden["count_fnv"]
    assert f"{oracle.fnv1a64(rt.zbuf()):016x}" == golden["zbuf_fnv"]
    assert f"{oracle.fnv1a64(rt.steps()):016x}" == golden["steps_fnv"]
    assert rt.max() == golden["max"]
    # colorize (gas, both alpha modes) — exact
    for transparent in (0, 1):
        c2 = cfg.replace(transparent=transparent)
        np.testing.assert_array_equal(sar.colorize(c2, rt), oracle.colorize(c2.c, ort))


@pytest.mark.parametrize("variant", [0, 1, 3])
@pytest.mark.parametrize("block", [64, 256])
def test_many_jobs_bit_exact(sar, oracle, gpu, variant, block):
    """Thousands of short trajectories: exercises depth ties, the checkpoint resolve and both bin layouts."""
    jobs, n = 4096, 1500
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=384, height=320, jobs_total=jobs, seed=11)
    starts = sar.start_points(11, 0, jobs)
    rt = sar.Runtime(cfg)
    rt.set_tuning(block_threads=block, checkpoint_stride=64, variant=variant)
    sar.render_jobs(cfg, rt, starts)
    ort = oracle.Runtime(384, 320)
    oracle.render_jobs(cfg.c, ort, starts, n)
    assert_state_equal(rt, ort, f"variant={variant} block={block}")
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))


@pytest.mark.parametrize("stride", [1, 7, 64, 100000])
def test_checkpoint_stride_does_not_change_results(sar, oracle, gpu, stride):
    jobs, n = 512, 700
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=200, height=160, jobs_total=jobs)
    starts = sar.start_points(3, 0, jobs)
    rt = sar.Runtime(cfg)
    rt.set_tuning(checkpoint_stride=stride)
    sar.render_jobs(cfg, rt, starts)
    ort = oracle.Runtime(200, 160)
    oracle.render_jobs(cfg.c, ort, starts, n)
    assert_state_equal(rt, ort, f"stride={stride}")


def test_solar_sail_divergent_jobs_and_depth(sar, oracle, gpu):
    """~38 % of solar-sail start points blow up to NaN: those jobs only feed count[0] (and only that)."""
    jobs, n = 2048, 1200
    cfg = _cfg(sar, "solar_sail", iterations=jobs * n, width=360, height=400, jobs_total=jobs,
               render_kind=sar.SAR_RENDER_DEPTH, scale=1.0)
    starts = sar.start_points(5, 0, jobs)
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt, starts)
    ort = oracle.Runtime(360, 400)
    oracle.render_jobs(cfg.c, ort, starts, n)
    assert ort.count[0, 0] > 100 * n, "expected many divergent jobs in this sample"
    assert_state_equal(rt, ort, "solar-sail")
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))      # depth image
    gas = cfg.replace(render_kind=sar.SAR_RENDER_GAS)
    np.testing.assert_array_equal(sar.colorize(gas, rt), oracle.colorize(gas.c, ort))      # gas image
    lib17 = cfg.replace(scale=1.7, angle=0.7)                                                # library scale, rotated
    rt.reset(); ort.reset()
    sar.render_jobs(lib17, rt, starts)
    oracle.render_jobs(lib17.c, ort, starts, n)
    assert_state_equal(rt, ort, "solar-sail scale 1.7 angle 0.7")


def test_render_accumulates_across_calls_and_reset(sar, oracle, gpu):
    """render on a non-reset Runtime continues the image (src/lib.rs:742-744); earlier calls win depth ties."""
    jobs, n = 256, 900
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=160, height=160, jobs_total=jobs)
    s1, s2 = sar.start_points(1, 0, jobs), sar.start_points(2, 0, jobs)
    rt = sar.Runtime(cfg)
    ort = oracle.Runtime(160, 160)
    sar.render_jobs(cfg, rt, s1); oracle.render_jobs(cfg.c, ort, s1, n)
    sar.render_jobs(cfg, rt, s2); oracle.render_jobs(cfg.c, ort, s2, n)
    assert_state_equal(rt, ort, "two calls")
    # the same 2*jobs trajectories in ONE call are the same sequential render
    rt2 = sar.Runtime(cfg)
    both = cfg.replace(iterations=2 * jobs * n, jobs_total=2 * jobs)
    sar.render_jobs(both, rt2, np.concatenate([s1, s2]))
    assert_state_equal(rt2, ort, "one call")
    rt.reset(); ort.reset()
    assert_state_equal(rt, ort, "after reset")
    assert rt.max() == 0


def test_launch_chunking_is_invisible(sar, oracle, gpu):
    """Forcing many launch chunks (test hook) must not change a bit: chunk boundaries fall on whole jobs and
    later chunks only win with a strictly greater depth."""
    jobs, n = 1000, 600
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=128, height=128, jobs_total=jobs)
    starts = sar.start_points(9, 0, jobs)
    ort = oracle.Runtime(128, 128)
    oracle.render_jobs(cfg.c, ort, starts, n)
    for variant in (1, 3):
        for cap in (1, 64, 333):
            rt = sar.Runtime(cfg)
            rt.set_tuning(block_threads=64, variant=variant | (cap << 8))
            sar.render_jobs(cfg, rt, starts)
            assert_state_equal(rt, ort, f"variant {variant} chunk cap {cap}")


def test_merge_matches_reference_semantics(sar, oracle, gpu):
    jobs, n = 300, 800
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=96, height=96, jobs_total=jobs)
    sa, sb = sar.start_points(21, 0, jobs), sar.start_points(22, 0, jobs)
    ra, rb = sar.Runtime(cfg), sar.Runtime(cfg)
    oa, ob = oracle.Runtime(96, 96), oracle.Runtime(96, 96)
    sar.render_jobs(cfg, ra, sa); sar.render_jobs(cfg, rb, sb)
    oracle.render_jobs(cfg.c, oa, sa, n); oracle.render_jobs(cfg.c, ob, sb, n)
    ra.merge(rb)
    assert oracle.merge(oa, ob) == 0
    assert_state_equal(ra, oa, "merge")
    # merge == rendering both job lists in sequence for count/zbuf; steps may differ only on depth ties
    # (merge keeps self on ties, like the reference) — here checked against the oracle's merge above.
    other = sar.Runtime(cfg.replace(width=64, height=64))
    with pytest.raises(sar.SarError) as e:
        ra.merge(other)
    assert e.value.status == 2  # SAR_ERR_DIM_MISMATCH (assert_eq! in the reference)


def test_load_roundtrip_and_wrapping_merge(sar, oracle, gpu):
    """u32 counts wrap like the release build (src/lib.rs:719); max follows the merged values only."""
    w = h = 8
    cfg = _cfg(sar, "poisson_saturne", width=w, height=h)
    rng = np.random.default_rng(0)
    ca = rng.integers(0, 2**32, size=(h, w), dtype=np.uint64).astype(np.uint32)
    cb = rng.integers(0, 2**32, size=(h, w), dtype=np.uint64).astype(np.uint32)
    za = rng.uniform(-1, 1, size=(h, w)).astype(np.float32); za[0, :3] = -1.0
    zb = rng.uniform(-1, 1, size=(h, w)).astype(np.float32); zb[0, 1:4] = -1.0; zb[1, 0] = za[1, 0]
    sa, sb = rng.uniform(-1, 2, size=(h, w)), rng.uniform(-1, 2, size=(h, w))
    ra, rb = sar.Runtime(cfg), sar.Runtime(cfg)
    ra.load(ca, sa, za, 17); rb.load(cb, sb, zb, 99)
    np.testing.assert_array_equal(ra.count(), ca)
    np.testing.assert_array_equal(_bits(ra.zbuf()), _bits(za))
    np.testing.assert_array_equal(_bits(ra.steps()), _bits(sa))
    assert ra.max() == 17
    oa, ob = oracle.Runtime(w, h), oracle.Runtime(w, h)
    oa.count[:] = ca; oa.steps[:] = sa; oa.zbuf[:] = za; oa.set_max(17)
    ob.count[:] = cb; ob.steps[:] = sb; ob.zbuf[:] = zb; ob.set_max(99)
    ra.merge(rb); oracle.merge(oa, ob)
    assert_state_equal(ra, oa, "wrapping merge")


def test_render_parallel_job_split(sar, oracle, gpu):
    """render_parallel: n = N / units / jobs_per_unit (two floor divisions), units*jobs_per_unit jobs,
    start points from the renderer's stream, reset before, colorize after (src/lib.rs:1051-1082)."""
    units, jpu = 192, 3
    cfg = _cfg(sar, "poisson_saturne", iterations=1_000_003, width=256, height=192, transparent=0)
    r = sar.ParallelRenderer(units=units, seed=77)
    assert r.num_threads() == units
    img = sar.render_parallel(r, cfg, jpu)
    n = 1_000_003 // units // jpu
    starts = oracle.start_points(77, 0, units * jpu)
    ort = oracle.Runtime(256, 192)
    oracle.render_jobs(cfg.c, ort, starts, n)
    np.testing.assert_array_equal(img, oracle.colorize(cfg.c, ort))
    assert_state_equal(r.runtime(), ort, "render_parallel")
    # second frame: the runtime is reset, the stream continues (the reference's RNGs persist across frames)
    img2 = sar.render_parallel(r, cfg.replace(angle=0.5), jpu)
    starts2 = oracle.start_points(77, units * jpu, units * jpu)
    ort.reset()
    c2 = cfg.replace(angle=0.5)
    oracle.render_jobs(c2.c, ort, starts2, n)
    np.testing.assert_array_equal(img2, oracle.colorize(c2.c, ort))
    r.shutdown()
    assert sar.ParallelRenderer().num_threads() % 64 == 0  # default: 64 units per CU


def test_render_single_draws_from_seeded_stream(sar, oracle, gpu):
    cfg = _cfg(sar, "poisson_saturne", iterations=200_000, width=128, height=128, seed=5)
    rt = sar.Runtime(cfg)
    sar.render(cfg, rt)
    sar.render(cfg, rt)
    starts = oracle.start_points(5, 0, 2)
    ort = oracle.Runtime(128, 128)
    oracle.render(cfg.c, ort, starts[0], 200_000)
    oracle.render(cfg.c, ort, starts[1], 200_000)
    assert_state_equal(rt, ort, "render x2")


def test_edge_cases(sar, oracle, gpu):
    cfg = _cfg(sar, "poisson_saturne", iterations=0, width=33, height=17, jobs_total=5)
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt)                      # zero iterations: nothing changes
    assert rt.count().sum() == 0 and rt.max() == 0
    assert np.all(rt.zbuf() == -1.0) and np.all(rt.steps() == 0.0)
    img = sar.colorize(cfg, rt)                   # max == 0: ln(1)/ln(1) = NaN -> 0 (src/lib.rs:860-869)
    ort = oracle.Runtime(33, 17)
    np.testing.assert_array_equal(img, oracle.colorize(cfg.c, ort))
    d = cfg.replace(render_kind=sar.SAR_RENDER_DEPTH)
    np.testing.assert_array_equal(sar.colorize(d, rt), oracle.colorize(d.c, ort))
    # iterations < jobs: per-job count floors to zero
    few = cfg.replace(iterations=4, jobs_total=5)
    sar.render_jobs(few, rt)
    assert rt.count().sum() == 0
    # 1x1 image, ragged job count (not a multiple of the wave size)
    tiny = _cfg(sar, "poisson_saturne", iterations=77 * 130, width=1, height=1, jobs_total=77)
    rt1 = sar.Runtime(tiny)
    st = sar.start_points(1, 0, 77)
    sar.render_jobs(tiny, rt1, st)
    o1 = oracle.Runtime(1, 1)
    oracle.render_jobs(tiny.c, o1, st, 130)
    assert_state_equal(rt1, o1, "1x1")
    # config / runtime size mismatch is an error, not UB
    with pytest.raises(sar.SarError):
        sar.render(cfg.replace(width=34), rt)
    with pytest.raises(sar.SarError):
        sar.Runtime(cfg.replace(palette_len=0))
    # zoomed / rotated view with most points out of bounds (previous_point must still advance, :793)
    zoom = _cfg(sar, "poisson_saturne", iterations=300 * 500, width=80, height=60, jobs_total=300,
                scale=6.0, angle=2.2)
    rz, oz = sar.Runtime(zoom), oracle.Runtime(80, 60)
    sz = sar.start_points(4, 0, 300)
    sar.render_jobs(zoom, rz, sz); oracle.render_jobs(zoom.c, oz, sz, 500)
    assert 0 < oz.count.sum() < 300 * 500
    assert_state_equal(rz, oz, "zoom")
    np.testing.assert_array_equal(sar.colorize(zoom, rz), oracle.colorize(zoom.c, oz))


def test_custom_palette_and_brightness(sar, oracle, gpu):
    pal = np.array([[0.1, 0.9, 0.3], [0.8, 0.2, 0.6], [0.4, 0.4, 1.0]])
    cfg = _cfg(sar, "poisson_saturne", iterations=64 * 2000, width=96, height=64, jobs_total=64,
               palette_rgb=pal, brightness_offset=-0.25, brightness_factor=2.0, transparent=1)
    st = sar.start_points(8, 0, 64)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(96, 64)
    sar.render_jobs(cfg, rt, st); oracle.render_jobs(cfg.c, ort, st, 2000)
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))


@pytest.mark.parametrize("transparent", [0, 1])
@pytest.mark.parametrize("case", ["plain", "empty_frame", "wrapped_max", "negative_zero_palette", "inf_palette", "nan_palette",
                                  "negative_palette", "zero_offset", "negative_zero_offset"])
def test_colorize_of_unvisited_pixels_takes_no_shortcut_it_cannot_prove(sar, oracle, gpu, case, transparent):
    """k_colorize_gas skips the palette, the square roots and the division for a pixel nobody visited (factor = ln 1 / ln(max + 1) =
    +0, -0 or NaN; r * factor is then the factor whatever r) — where r is provably finite. Unvisited pixels with every kind of
    `steps` (the reset 0.0, NaN, infinities, out of range), under palettes with -0.0 / inf / NaN / negative entries, an empty
    frame (max 0: 0 / 0), a wrapped max (ln 0 = -inf: factor -0.0) and zero brightness offsets: bit for bit the oracle's image."""
    w, h = 64, 32
    pal = np.array([[0.1, 0.9, 0.3], [0.8, 0.0, 0.6], [0.4, 0.4, 1.0]])
    kw = dict(brightness_offset=-0.15, brightness_factor=5.0 / 3.0)
    if case == "negative_zero_palette":
        pal[0] = [-0.0, -0.0, 0.5]; pal[1] = [-0.0, 0.2, -0.0]
    elif case == "inf_palette":
        pal[0, 1] = np.inf
    elif case == "nan_palette":
        pal[1, 2] = np.nan
    elif case == "negative_palette":
        pal[0, 0] = -0.3
    elif case == "zero_offset":
        kw["brightness_offset"] = 0.0
    elif case == "negative_zero_offset":
        kw["brightness_offset"] = -0.0
    cfg = _cfg(sar, "poisson_saturne", width=w, height=h, transparent=transparent, palette_rgb=pal, **kw)
    rng = np.random.default_rng(17)
    cnt = np.zeros((h, w), np.uint32)
    steps = np.zeros((h, w))
    visited = rng.random((h, w)) < 0.25
    if case != "empty_frame":
        cnt[visited] = rng.integers(1, 5000, size=int(visited.sum()))
        steps[visited] = rng.uniform(0, 1, size=int(visited.sum()))
    odd = [np.nan, np.inf, -np.inf, -0.0, 1.0, 0.999999, 1.5, -2.0, 5e-324, 0.3333]
    free = np.argwhere(cnt == 0)
    for i, v in enumerate(odd * 6):                           # unvisited pixels whose steps are not the reset state
        steps[tuple(free[i * 7 % len(free)])] = v
    mx = 0xFFFFFFFF if case == "wrapped_max" else int(cnt.max())
    z = np.full((h, w), -1.0, np.float32)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    rt.load(cnt, steps, z, mx)
    ort.count[:] = cnt; ort.steps[:] = steps; ort.zbuf[:] = z; ort.set_max(mx)
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))
    rt.close()


def test_colorize_beyond_ln_table_within_one_lsb(sar, oracle, gpu):
    """count+1 > 2^20 uses the device log (<= 1 ulp): RGBA16 may differ by at most 1 LSB there."""
    w = h = 16
    cfg = _cfg(sar, "poisson_saturne", width=w, height=h, transparent=1)
    rng = np.random.default_rng(3)
    cnt = rng.integers(2**20, 2**31, size=(h, w), dtype=np.uint64).astype(np.uint32)
    cnt[0, 0] = 2**32 - 1                         # count+1 wraps to 0 -> ln(0) = -inf
    steps = rng.uniform(-0.2, 1.2, size=(h, w)); steps[0, 1] = np.nan
    z = rng.uniform(-1, 1, size=(h, w)).astype(np.float32)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    mx = int(cnt[1:].max())
    rt.load(cnt, steps, z, mx)
    ort.count[:] = cnt; ort.steps[:] = steps; ort.zbuf[:] = z; ort.set_max(mx)
    a, b = sar.colorize(cfg, rt).astype(np.int64), oracle.colorize(cfg.c, ort).astype(np.int64)
    assert np.abs(a - b).max() <= 1


def test_device_sqrt_div_are_correctly_rounded(sar, oracle, gpu):
    """The colour transform's sqrt and division must be IEEE-exact on the device: drive them through the
    depth-winner path with an image so coarse that almost every visit is a winner candidate."""
    jobs, n = 1024, 400
    for preset in ("poisson_saturne", "solar_sail"):
        cfg = _cfg(sar, preset, iterations=jobs * n, width=1024, height=1024, jobs_total=jobs, scale=1.0)
        st = sar.start_points(31, 0, jobs)
        rt, ort = sar.Runtime(cfg), oracle.Runtime(1024, 1024)
        sar.render_jobs(cfg, rt, st); oracle.render_jobs(cfg.c, ort, st, n)
        assert (ort.zbuf != -1).sum() > 50_000
        assert_state_equal(rt, ort, preset)


@pytest.mark.parametrize("interleave", [0, 1, 2])
@pytest.mark.parametrize("size", [(1920, 1080), (3000, 2500), (3072, 3072), (4096, 4096), (8192, 6000)])
def test_bin_geometries(sar, oracle, gpu, size, interleave):
    """Image sizes that exercise every bin geometry of the LDS-binned path (ragged last bin, 512+ bins, bin counts that
    are and are not powers of two, bins of 65536 pixels counted in two halves at 4096^2 and at 49 Mpx), under both pixel -> (bin, record) maps:
    bins of consecutive pixels (1) and bins dealt round-robin in 2048-pixel segments (2); 0 = the host's choice."""
    w, h = size
    jobs, n = 2048, 300
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    st = sar.start_points(13, 0, jobs)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    rt.set_option("bin_interleave", interleave)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, f"{w}x{h} bin_interleave={interleave}")


@pytest.mark.parametrize("bin_shift", [12, 13, 14, 15, 16])
@pytest.mark.parametrize("interleave", [1, 2])
@pytest.mark.parametrize("size", [(700, 500), (64, 48), (1, 1), (2048, 3)])
def test_bin_maps_small_and_odd_shapes(sar, oracle, gpu, size, interleave, bin_shift):
    """Both bin maps with every bin size on images of one bin, of a few bins, of one pixel and of three very long rows
    (segments of the interleaved map that straddle rows), two render calls into one runtime (the segment flags of the
    second launch must not hide the first launch's pixels)."""
    w, h = size
    jobs, n = 700, 400
    cfg = _cfg(sar, "solar_sail", iterations=jobs * n, width=w, height=h, jobs_total=jobs, scale=0.9)
    st = sar.start_points(29, 0, 2 * jobs)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    rt.set_tuning(variant=3, bin_shift=bin_shift, bin_interleave=interleave, chunk_records=60 if bin_shift >= 15 else 0)
    for part in (st[:jobs], st[jobs:]):
        sar.render_jobs(cfg, rt, part)
        oracle.render_jobs(cfg.c, ort, part, n)
    assert_state_equal(rt, ort, f"{w}x{h} bin_shift={bin_shift} bin_interleave={interleave}")


@pytest.mark.parametrize("records", [12, 20, 28, 60])
@pytest.mark.parametrize("acc_lists", [1, 4])
def test_packed_counters_with_every_chunk_size(sar, oracle, gpu, records, acc_lists):
    """Bins of 65536 pixels — k_bin_accumulate counts them with packed 16-bit counters — under every chunk size, on an image
    of 5 such bins whose last one is ragged, interleaved (8 bins) and not."""
    w, h = 640, 480
    jobs, n = 1500, 500
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    st = sar.start_points(37, 0, jobs)
    for interleave in (1, 2):
        rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
        rt.set_tuning(variant=3, bin_shift=16, chunk_records=records, acc_lists=acc_lists, bin_interleave=interleave)
        sar.render_jobs(cfg, rt, st)
        oracle.render_jobs(cfg.c, ort, st, n)
        assert_state_equal(rt, ort, f"records={records} acc_lists={acc_lists} bin_interleave={interleave}")


def _fixed_point_config(sar, **kw):
    """A contrived coefficient set: every coordinate's polynomial is a constant, so every trajectory sits on ONE point from
    its first iteration on and every visit of every job lands on ONE pixel (a point of the poisson-saturne attractor, so the
    preset's view shows it)."""
    pt = (float.fromhex("0x1.d37397ce5279dp-3"), float.fromhex("0x1.494519191dfdbp-3"), float.fromhex("-0x1.ff8befd61a1b4p-3"))
    coef = lambda c: [c] + [0.0] * 9
    return _cfg(sar, "poisson_saturne", coeff_x=coef(pt[0]), coeff_y=coef(pt[1]), coeff_z=coef(pt[2]), **kw)


def _expect_hot_pixel(sar, oracle, cfg, w, h, total):
    """What `total` visits of the one pixel leave: depth and payload of the FIRST visit (later ones tie: strict `>`, :821),
    count = total mod 2^32 (:811), max = u32::MAX once the count has wrapped (:813-815)."""
    small = cfg.replace(iterations=64 * 8, jobs_total=64)
    st = sar.start_points(5, 0, 64)
    ort = oracle.Runtime(w, h)
    oracle.render_jobs(small.c, ort, st, 8)
    (ys, xs) = np.nonzero(ort.count)
    assert len(ys) == 1 and ort.count[ys[0], xs[0]] == 64 * 8
    ort.count[ys[0], xs[0]] = total & 0xFFFFFFFF
    ort.set_max(0xFFFFFFFF if total >> 32 else total)
    return ort, (int(ys[0]), int(xs[0]))


def test_hot_pixel_through_the_packed_counters(sar, oracle, gpu):
    """3e8 visits of ONE pixel through bins of 65536 pixels: with one workgroup per bin the packed 16-bit counter of that
    pixel overflows ~9000 times — more events than the workgroup's list holds, so both the event list and the
    straight-to-memory path of k_bin_accumulate's PACKED mode carry hits."""
    w = h = 1024
    jobs, n = 65536, 4578
    cfg = _fixed_point_config(sar, iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    rt = sar.Runtime(cfg)
    rt.set_tuning(variant=3, bin_shift=16, splits=1)
    sar.render_jobs(cfg, rt, sar.start_points(5, 0, jobs))
    ort, _ = _expect_hot_pixel(sar, oracle, cfg, w, h, jobs * n)
    assert_state_equal(rt, ort, "hot pixel")


def test_hot_pixel_past_u32_through_the_binned_path(sar, oracle, gpu):
    """One real pixel driven past 2^32 hits by the default path (reference src/lib.rs:811-815, 860): the fold's wrap flag
    makes `max` read u32::MAX, the count is the total mod 2^32, and colorize takes ln(max + 1) = ln(0)."""
    w = h = 2048
    jobs, n = 131072, 32769            # 4 295 098 368 = 2^32 + 131072 visits
    cfg = _fixed_point_config(sar, iterations=jobs * n, width=w, height=h, jobs_total=jobs, transparent=1)
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt, sar.start_points(5, 0, jobs))
    ort, (y, x) = _expect_hot_pixel(sar, oracle, cfg, w, h, jobs * n)
    assert ort.count[y, x] == 131072
    assert_state_equal(rt, ort, "hot pixel past 2^32")
    assert rt.max() == 0xFFFFFFFF
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))


@pytest.mark.parametrize("preset", ["poisson_saturne", "solar_sail"])
def test_ragged_job_count_both_presets_both_kinds(sar, oracle, gpu, preset):
    """A job count that leaves a ragged last workgroup, NaN-absorbing trajectories (solar_sail) and both render
    kinds, at a non-square size."""
    jobs, n = 1000 + 37, 900
    for kind in (0, 1):
        cfg = _cfg(sar, preset, iterations=jobs * n, width=640, height=480, jobs_total=jobs, render_kind=kind)
        st = sar.start_points(21, 0, jobs)
        rt, ort = sar.Runtime(cfg), oracle.Runtime(640, 480)
        sar.render_jobs(cfg, rt, st)
        oracle.render_jobs(cfg.c, ort, st, n)
        assert_state_equal(rt, ort, f"{preset} kind={kind}")
        np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))


def test_deterministic_across_runs(sar, gpu):
    jobs, n = 8192, 500
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=512, height=512, jobs_total=jobs)
    st = sar.start_points(1, 0, jobs)
    outs = []
    for _ in range(2):
        rt = sar.Runtime(cfg)
        sar.render_jobs(cfg, rt, st)
        outs.append((rt.count(), rt.zbuf(), rt.steps(), rt.max()))
    for a, b in zip(outs[0][:3], outs[1][:3]):
        np.testing.assert_array_equal(_bits(a), _bits(b))
    assert outs[0][3] == outs[1][3]


def test_exchange_kernels_reproduce_merge_in_rank_order(sar, oracle, gpu):
    """The device-side pack/select/import of the multi-GPU merge (sar_runtime_exchange_*), with the two
    collectives (all-reduce MAX, reduce SUM) replaced by elementwise torch ops over three 'ranks' on one GPU."""
    import torch
    w, h, jobs, n = 200, 150, 600, 700
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    world = 3
    rts, orts = [], []
    for r in range(world):
        st = sar.start_points(40 + r, 0, jobs)
        rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
        sar.render_jobs(cfg, rt, st)
        oracle.render_jobs(cfg.c, ort, st, n)
        rts.append(rt); orts.append(ort)
    # make rank 1 and rank 2 tie with rank 0 on some pixels (lowest rank must win the tie)
    z0, c0, s0 = rts[0].zbuf(), rts[0].count(), rts[0].steps()
    z1, c1, s1 = rts[1].zbuf(), rts[1].count(), rts[1].steps()
    tie = (z0 != -1.0) & (z1 != -1.0)
    z1[tie] = z0[tie]
    rts[1].load(c1, s1, z1, rts[1].max())
    orts[1].zbuf[:] = z1
    npix = w * h
    keys = [torch.empty(npix, dtype=torch.int64, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    exs = [sar.Exchange(rts[r], world, r) for r in range(world)]
    for r in range(world):
        exs[r].rooted(0, keys[r].data_ptr())
        rts[r].synchronize()
    red = torch.stack(keys).max(dim=0).values.contiguous()          # all-reduce MAX
    torch.cuda.synchronize()  # torch's stream before the runtimes' streams
    sums = [torch.empty(3 * npix, dtype=torch.int32, device="cuda") for _ in range(world)]
    for r in range(world):
        exs[r].rooted(1, red.data_ptr(), sums[r].data_ptr())
        rts[r].synchronize()
    total = torch.stack(sums).sum(dim=0, dtype=torch.int32).contiguous()   # reduce SUM (wrapping int32)
    torch.cuda.synchronize()
    exs[0].rooted(2, red.data_ptr(), total.data_ptr())
    rts[0].synchronize()
    for ex in exs:
        ex.close()
    acc = orts[0]
    for other in orts[1:]:
        assert oracle.merge(acc, other) == 0
    assert_state_equal(rts[0], acc, "exchange == merge folded in rank order")
    assert tie.sum() > 1000


@pytest.mark.parametrize("preset", ["poisson_saturne", "solar_sail"])
def test_attractor_extent_bit_exact(sar, oracle, gpu, preset):
    """sar_runtime_extent (the reference's TODO first pass, src/lib.rs:326-333) against the oracle: min/max are exact
    and order-free, so all 12 bounds must match bit for bit — with a ragged job count and NaN-ending jobs."""
    jobs, n = 1000 + 13, 3000
    cfg = _cfg(sar, preset, iterations=jobs * n, width=64, height=64, jobs_total=jobs)
    st = sar.start_points(17, 0, jobs)
    rt = sar.Runtime(cfg)
    got = sar.attractor_extent(cfg, rt, jobs, n, st)
    want = oracle.extent(cfg.c, st, n)
    np.testing.assert_array_equal(_bits(got), _bits(want))
    assert np.all(np.isfinite(got))
    # drawing the start points from the runtime's stream gives the same bounds as passing that stream's points
    rt.seed(17)
    np.testing.assert_array_equal(_bits(sar.attractor_extent(cfg, rt, jobs, n)), _bits(want))


@pytest.mark.parametrize("records", [12, 20, 28, 60])
@pytest.mark.parametrize("splits,acc_threads,hint_bits,interleave,acc_lists,split_waves",
                         [(0, 0, 0, 0, 0, 0), (1, 256, 16, 1, 1, 1), (5, 512, 16, 2, 4, 2), (16, 1024, 32, 1, 1, 2),
                          (3, 1024, 32, 2, 4, 1), (2, 1024, 32, 2, 4, 2)])
def test_chunk_sizes_and_accumulate_shapes_bit_exact(sar, oracle, gpu, records, splits, acc_threads, hint_bits, interleave, acc_lists,
                                                     split_waves):
    """Every chunk size of the binned path (32 / 48-on-64 / 64 / 128-byte chunks: different lane-group shapes in
    k_bin_accumulate), the iterate kernel whole (split_waves 1) and as producer / consumer wave pairs (2: 64- and 128-byte
    chunks), both hint types, with several accumulate grids, against the oracle; enough records per (bin, wave) list to chain
    many chunks and to overflow staging buffers within one slot request (all trajectories start close together)."""
    jobs, n = 2048 + 64, 1201  # odd: the depth pipeline's pass of 2 leaves a last single iteration
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=256, height=192, jobs_total=jobs)
    st = sar.start_points(23, 0, jobs)
    st[:512] = st[0] + np.arange(512)[:, None] * 1e-13  # near-identical trajectories: many lanes hit one bin at once
    rt, ort = sar.Runtime(cfg), oracle.Runtime(256, 192)
    rt.set_tuning(variant=3, chunk_records=records, splits=splits, acc_threads=acc_threads, hint_bits=hint_bits,
                  bin_interleave=interleave, acc_lists=acc_lists, split_waves=split_waves)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, f"records={records} splits={splits} acc_threads={acc_threads} hint_bits={hint_bits} "
                               f"bin_interleave={interleave} acc_lists={acc_lists} split_waves={split_waves}")


@pytest.mark.parametrize("size", [(256, 64), (4096, 8), (64, 24), (1024, 1024)])
def test_tiled_narrow_hints_on_power_of_two_widths(sar, oracle, gpu, size):
    """16-bit depth hints of images whose width is a power of two (and whose height is a multiple of eight) live in 8 x 8
    tiles (HintTile); "hint_tile" 1 keeps them row-major. Both layouts, two render calls into one runtime (the hints of the
    first call filter the second), against the oracle; switching the layout in between clears the hints."""
    w, h = size
    jobs, n = 1024, 600
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs, scale=0.8)
    st = sar.start_points(61, 0, 2 * jobs)
    for tile in (0, 1):
        rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
        rt.set_tuning(variant=3, hint_bits=16, hint_tile=tile)
        for part, flip in ((st[:jobs], False), (st[jobs:], tile == 0)):
            if flip:
                rt.set_option("hint_tile", 1)   # another layout mid-way: the old hints must not be read under it
            sar.render_jobs(cfg, rt, part)
            oracle.render_jobs(cfg.c, ort, part, n)
        assert_state_equal(rt, ort, f"{w}x{h} hint_tile={tile}")
        assert "q16" in rt.describe_last_launch()


@pytest.mark.parametrize("seed", range(32))
def test_random_configurations_bit_exact(sar, oracle, gpu, seed):
    """Seeded random shapes (tiny to ragged images, 1..3000 jobs, 1..900 iterations, both presets and render kinds,
    view angle / scale / brightness / transparency) against the oracle, including the converted export formats."""
    rng = np.random.default_rng(1000 + seed + _FUZZ_BASE)
    preset = ["poisson_saturne", "solar_sail"][int(rng.integers(2))]
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 500))
    jobs, n = int(rng.integers(1, 3000)), int(rng.integers(1, 900))
    kw = dict(iterations=jobs * n, width=w, height=h, jobs_total=jobs, render_kind=int(rng.integers(2)),
              transparent=int(rng.integers(2)), angle=float(rng.uniform(0, 6.3)), scale=float(rng.uniform(0.4, 2.5)),
              brightness_offset=float(rng.uniform(-0.4, 0.1)))
    cfg = _cfg(sar, preset, **kw)
    st = sar.start_points(int(rng.integers(1 << 30)), 0, jobs)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, f"seed {seed}: {preset} {w}x{h} jobs={jobs} n={n} {kw}")
    want = oracle.colorize(cfg.c, ort)
    np.testing.assert_array_equal(sar.colorize(cfg, rt), want)
    fmt = int(rng.integers(4))
    np.testing.assert_array_equal(sar.colorize_format(cfg, rt, fmt), oracle.convert(fmt, want))


@pytest.mark.parametrize("seed", range(16))
def test_custom_attractors_views_and_transforms_bit_exact(sar, oracle, gpu, seed):
    """The reference's Config is generic data, not two presets (src/lib.rs:253-308): any 30 coefficients
    (PolynomialSprott2Degree, :575-580), any View (center_camera, un-normalised axis, rotation, scale: :291-308), either
    colour transform with its own offset / factor (:507-516) or the other preset's, a palette of 1..8 entries (:406-473).
    Perturbed coefficients leave the attractor bounded, let it collapse to a point or blow up to inf / NaN within the
    frame — every one of those has to give the oracle's bits, dead jobs and all."""
    rng = np.random.default_rng(77_000 + seed + _FUZZ_BASE)
    preset = ["poisson_saturne", "solar_sail"][seed % 2]
    base = _cfg(sar, preset)
    rel = [0.0, 1e-6, 1e-3, 3e-2, 0.3][int(rng.integers(5))]       # how far from the preset's map
    co = {k: np.array(getattr(base.c, k)[:]) * (1.0 + rel * rng.standard_normal(10)) for k in ("coeff_x", "coeff_y", "coeff_z")}
    if seed % 5 == 0:
        co["coeff_y"][int(rng.integers(10))] = -0.0                  # a signed zero among the coefficients
    w, h = int(rng.integers(8, 900)), int(rng.integers(8, 700))
    jobs, n = int(rng.integers(1, 1500)), int(rng.integers(1, 1200))
    kw = dict(iterations=jobs * n, width=w, height=h, jobs_total=jobs, render_kind=int(rng.integers(2)),
              transparent=int(rng.integers(2)), angle=float(rng.uniform(-7, 7)), scale=float(rng.uniform(0.2, 3.0)),
              center_camera=rng.uniform(-0.6, 0.6, 3), rotation_axis=rng.uniform(-1.5, 1.5, 3),
              rotation_angle=float(rng.uniform(-4, 4)), color_transform=int(rng.integers(2)),
              ct_offset=float(rng.uniform(-1, 1)), ct_factor=float(rng.uniform(-2, 2)),
              palette_rgb=rng.uniform(0, 1, (int(rng.integers(1, 9)), 3)), brightness_offset=float(rng.uniform(-0.5, 0.2)),
              brightness_factor=float(rng.uniform(0.5, 3.0)), **co)
    cfg = base.replace(**kw)
    cfg.validate()
    st = sar.start_points(int(rng.integers(1 << 30)), 0, jobs)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, f"seed {seed}: {preset} rel={rel} {w}x{h} jobs={jobs} n={n}")
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))
    # the same configuration through the job split of render_parallel (:1051-1082)
    units, jpt = 8, 3
    pr = sar.ParallelRenderer(units=units, seed=5)
    c2 = cfg.replace(iterations=units * jpt * 257)
    img = sar.render_parallel(pr, c2, jpt)
    ort2 = oracle.Runtime(w, h)
    oracle.render_jobs(c2.replace(jobs_total=units * jpt).c, ort2, sar.start_points(5, 0, units * jpt), 257)
    np.testing.assert_array_equal(img, oracle.colorize(c2.c, ort2))
    pr.shutdown()


def test_device_resident_start_points_give_the_same_bits(sar, oracle, gpu):
    """sar_render_job_range_device: start points handed over in device memory (across several launch chunks)."""
    import torch
    jobs, n = 1500, 500
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=300, height=200, jobs_total=jobs)
    st = sar.start_points(31, 0, jobs)
    ort = oracle.Runtime(300, 200)
    oracle.render_jobs(cfg.c, ort, st, n)
    dev = torch.from_numpy(st).cuda()
    torch.cuda.synchronize()
    for cap in (0, 333):
        rt = sar.Runtime(cfg)
        rt.set_tuning(variant=cap << 8)
        sar.render_job_range_device(cfg, rt, jobs, n, dev.data_ptr())
        assert_state_equal(rt, ort, f"device starts, chunk cap {cap}")


@pytest.mark.parametrize("preset", ["poisson_saturne", "solar_sail"])
def test_announced_frames_whose_warm_up_ran_ahead(sar, oracle, gpu, preset):
    """sar_runtime_prefetch_device: the warm-up of the announced call runs on a second stream under the previous frame's
    tail. Three frames on one runtime (reset in between) — announced and matched, announced with other start points (the
    announcement is dropped), not announced — all equal the oracle; solar-sail's dead jobs reach count[0] exactly once."""
    import torch
    jobs, n = 4096, 300
    w, h = 640, 480
    cfg = _cfg(sar, preset, iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    sts = [sar.start_points(40 + k, 0, jobs) for k in range(3)]
    devs = [torch.from_numpy(s).cuda() for s in sts]
    torch.cuda.synchronize()
    rt = sar.Runtime(cfg)
    sar.prefetch_device(cfg, rt, jobs, n, devs[0].data_ptr())            # nothing in flight: runs at once
    for k, announce in enumerate([1, 0, None]):                           # frame 0 announces 1; frame 1 announces frame 0's points
        rt.reset()
        sar.render_job_range_device(cfg, rt, jobs, n, devs[k].data_ptr())
        if announce is not None:
            sar.prefetch_device(cfg, rt, jobs, n, devs[announce].data_ptr())
        ort = oracle.Runtime(w, h)
        oracle.render_jobs(cfg.c, ort, sts[k], n)
        assert_state_equal(rt, ort, f"{preset} frame {k}")
    # an announced call on an un-reset runtime continues the image (src/lib.rs:742-744)
    sar.prefetch_device(cfg, rt, jobs, n, devs[0].data_ptr())
    sar.render_job_range_device(cfg, rt, jobs, n, devs[0].data_ptr())
    oracle.render_jobs(cfg.c, ort, sts[0], n)
    assert_state_equal(rt, ort, f"{preset} continued")


def test_announced_frames_with_narrow_hints_and_launch_chunks(sar, oracle, gpu):
    """The announced warm-up also measures the depth range the 16-bit hints quantise (and hands it over with the buffers);
    an announced call of several launch chunks finds its FIRST chunk's warm-up done; an announcement does not survive
    another render call; sar_runtime_describe_last_launch says what ran."""
    import torch
    jobs, n = 6144, 250   # a multiple of the workgroup size: one launch chunk unless capped
    w, h = 700, 500
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    st = sar.start_points(77, 0, jobs)
    dev = torch.from_numpy(st).cuda()
    other = torch.from_numpy(sar.start_points(78, 0, jobs)).cuda()
    torch.cuda.synchronize()
    ort = oracle.Runtime(w, h)
    oracle.render_jobs(cfg.c, ort, st, n)
    for chunk_cap in (0, 2500):
        rt = sar.Runtime(cfg)
        rt.set_tuning(variant=3 | (chunk_cap << 8), hint_bits=16)
        sar.prefetch_device(cfg, rt, jobs, n, dev.data_ptr())
        sar.render_job_range_device(cfg, rt, jobs, n, dev.data_ptr())
        assert_state_equal(rt, ort, f"announced, narrow hints, chunk cap {chunk_cap}")
        d = rt.describe_last_launch()
        # warm-ups found done: the announced first chunk, and every further launch chunk's (a call of several chunks runs its
        # next chunk's warm-up under the current chunk's accumulate / fold by itself)
        assert "hints=q16" in d and f"chunks={1 if chunk_cap == 0 else 3} warmup_ahead={1 if chunk_cap == 0 else 3}" in d, d
        # an announcement is spent by the next render call, whatever that call renders
        rt.reset()
        sar.prefetch_device(cfg, rt, jobs, n, dev.data_ptr())
        sar.render_job_range_device(cfg, rt, jobs, n, other.data_ptr())
        rt.reset()
        sar.render_job_range_device(cfg, rt, jobs, n, dev.data_ptr())
        assert_state_equal(rt, ort, "after a dropped announcement")
        assert f"warmup_ahead={1 if chunk_cap == 0 else 7}" in rt.describe_last_launch()  # 3 + 2 + 2: no first chunk found any more
        rt.close()


@pytest.mark.parametrize("hint_bits", [32, 16])
def test_an_announcement_stands_for_another_view_but_not_another_map(sar, oracle, gpu, hint_bits):
    """The warm-up is the map alone (src/lib.rs:750-752): a frame announced under one view is found done by a call that
    renders the same jobs under ANOTHER angle / scale / kind (a sweep's next frame; render_parallel announces its next
    frame under the current view) — with the narrow hints too, whose quantiser then spans a depth range measured under the
    announcing view —, and is dropped when a coefficient differs."""
    import torch
    jobs, n, w, h = 6144, 260, 520, 410
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
    st = sar.start_points(91, 0, jobs)
    dev = torch.from_numpy(st).cuda()
    torch.cuda.synchronize()
    rt = sar.Runtime(cfg)
    rt.set_option("hint_bits", hint_bits)
    turned = cfg.replace(angle=2.2, scale=1.7, render_kind=sar.SAR_RENDER_DEPTH)
    other_map = cfg.replace(coeff_z=np.array(cfg.c.coeff_z[:]) * (1 + 1e-9))
    for announced, rendered, found in ((cfg, turned, 1), (cfg, other_map, 1), (turned, cfg, 2)):  # warm-ups found done so far
        rt.reset()
        sar.prefetch_device(announced, rt, jobs, n, dev.data_ptr())
        sar.render_job_range_device(rendered, rt, jobs, n, dev.data_ptr())
        ort = oracle.Runtime(w, h)
        oracle.render_jobs(rendered.c, ort, st, n)
        assert_state_equal(rt, ort, f"hint_bits {hint_bits}, announcements found so far {found}")
        assert f"warmup_ahead={found}" in rt.describe_last_launch(), rt.describe_last_launch()
        np.testing.assert_array_equal(sar.colorize(rendered, rt), oracle.colorize(rendered.c, ort))


def test_announced_calls_of_growing_and_shrinking_size(sar, oracle, gpu):
    """The two sets of warm-up buffers swap when an announced call is consumed, capacities included, while the converted
    start points of an announcement live in a buffer of their own: a small announced call followed by larger ones (and a
    call of several launch chunks, which announces its own next chunk) must not write past any of them."""
    import torch
    w, h, n = 300, 200, 120
    rt = None
    for k, jobs in enumerate((300, 9000, 700, 20000, 20000, 64)):
        cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=w, height=h, jobs_total=jobs)
        st = sar.start_points(500 + k, 0, jobs)
        dev = torch.from_numpy(st).cuda()
        torch.cuda.synchronize()
        if rt is None:
            rt = sar.Runtime(cfg)
            rt.set_option("debug_chunk_jobs", 6000)              # 9000 and 20000 jobs run as 2 and 4 launch chunks
        rt.reset()
        sar.prefetch_device(cfg, rt, jobs, n, dev.data_ptr())
        sar.render_job_range_device(cfg, rt, jobs, n, dev.data_ptr())
        ort = oracle.Runtime(w, h)
        oracle.render_jobs(cfg.c, ort, st, n)
        assert_state_equal(rt, ort, f"call {k}: {jobs} jobs")


def test_every_job_diverging_in_the_warm_up(sar, oracle, gpu):
    """No trajectory survives the warm-up (start points far outside the basin): the hot kernel has nothing to do, every
    counted iteration lands on pixel (0,0) (reference src/lib.rs:789, 800-802) and the depth buffer stays empty."""
    jobs, n = 300, 250
    cfg = _cfg(sar, "poisson_saturne", iterations=jobs * n, width=64, height=48, jobs_total=jobs)
    st = np.full((jobs, 3), 50.0) + np.arange(jobs)[:, None]
    rt, ort = sar.Runtime(cfg), oracle.Runtime(64, 48)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, "all jobs NaN")
    assert rt.count()[0, 0] == jobs * n and rt.count().sum() == jobs * n
    # a mix: the same diverging jobs interleaved with ordinary ones
    st2 = sar.start_points(3, 0, jobs)
    st2[::3] = st[::3]
    rt2, ort2 = sar.Runtime(cfg), oracle.Runtime(64, 48)
    sar.render_jobs(cfg, rt2, st2)
    oracle.render_jobs(cfg.c, ort2, st2, n)
    assert_state_equal(rt2, ort2, "every third job NaN")


@pytest.mark.parametrize("preset,kind", [("poisson_saturne", 0), ("solar_sail", 1)])
@pytest.mark.parametrize("variant", [1, 3])
def test_jobs_longer_than_one_launch_can_order_run_as_segments(sar, oracle, gpu, preset, kind, variant):
    """Config::iterations is a usize (src/lib.rs:267); a launch orders its visits with 32 bits. A job with more iterations
    runs as successive launches that hand the trajectory state on (no second warm-up) — forced here at a small size
    through the debug_max_ordinals hook: 3 segments per job (the last one shorter), one job per launch chunk, on the
    binned path and on the one-atomic-per-visit path, with jobs that diverge in the warm-up and later (solar-sail)."""
    jobs, n = 9, 12345
    cfg = _cfg(sar, preset, iterations=jobs * n, width=200, height=160, jobs_total=jobs, render_kind=kind)
    st = sar.start_points(77, 0, jobs)
    rt, ort = sar.Runtime(cfg), oracle.Runtime(200, 160)
    rt.set_tuning(variant=variant)
    rt.set_option("debug_max_ordinals", 5000)
    sar.render_jobs(cfg, rt, st)
    oracle.render_jobs(cfg.c, ort, st, n)
    assert_state_equal(rt, ort, f"segments {preset} variant={variant}")
    np.testing.assert_array_equal(sar.colorize(cfg, rt), oracle.colorize(cfg.c, ort))
