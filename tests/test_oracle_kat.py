"""CPU: the oracle against the committed known-answer vectors (tests/golden/kat.json)."""
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat.json")))


def hexes(v):
    return [float.hex(float(x)) for x in v]


def test_map_iterates(oracle):
    ps, ss = oracle.poisson_saturne(), oracle.solar_sail()
    p0 = np.array([0.05, 0.031, 0.077])
    for n, want in GOLD["poisson_iter"].items():
        assert hexes(oracle.iterate(ps, p0, int(n))) == want
    q0 = np.array([0.025, 0.0155, 0.0385])
    for n, want in GOLD["solar_iter"].items():
        assert hexes(oracle.iterate(ss, q0, int(n))) == want


def test_rotation_matrix_release_semantics(oracle):
    assert hexes(oracle.rotation_matrix(oracle.poisson_saturne()).ravel()) == GOLD["poisson_matrix"]
    m = oracle.rotation_matrix(oracle.solar_sail())
    assert hexes(m.ravel()) == GOLD["solar_matrix"]
    # solar-sail's axis is not unit length and is NOT normalised in release builds (src/lib.rs:181-183):
    # the "rotation" is not orthonormal
    assert abs(np.linalg.det(m) - 1.0) > 1e-3


def test_c1_single_trajectory(oracle):
    g = GOLD["c1_512"]
    c = oracle.poisson_saturne()
    c.width = c.height = 512
    rt = oracle.Runtime(512, 512)
    oracle.render(c, rt, np.array([float.fromhex(h) for h in g["p0"]]), g["iterations"])
    assert int(rt.count.sum()) == g["in_bounds"] and int((rt.count > 0).sum()) == g["touched"]
    assert rt.max == g["max"] == int(rt.count.max())
    assert f"{oracle.fnv1a64(rt.count):016x}" == g["count_fnv"]
    assert f"{oracle.fnv1a64(rt.zbuf):016x}" == g["zbuf_fnv"]
    assert f"{oracle.fnv1a64(rt.steps):016x}" == g["steps_fnv"]
    c.transparent = 0
    assert f"{oracle.fnv1a64(oracle.colorize(c, rt)):016x}" == g["rgba_fnv"]
    c.transparent = 1
    assert f"{oracle.fnv1a64(oracle.colorize(c, rt)):016x}" == g["rgba_transparent_fnv"]
    # 4 jobs on one un-reset runtime
    g4 = GOLD["c1_512_4jobs"]
    rt.reset()
    p0 = np.array([0.05, 0.031, 0.077])
    oracle.render_jobs(c, rt, np.stack([p0 * (k + 1) / 4 for k in range(4)]), 2_500_000)
    assert rt.max == g4["max"] and f"{oracle.fnv1a64(rt.count):016x}" == g4["count_fnv"]
    assert f"{oracle.fnv1a64(rt.zbuf):016x}" == g4["zbuf_fnv"] and f"{oracle.fnv1a64(rt.steps):016x}" == g4["steps_fnv"]


def test_solar_sail_depth_with_divergent_jobs(oracle):
    g = GOLD["solar_450x500_64jobs"]
    s = oracle.solar_sail()
    s.width, s.height, s.scale, s.render_kind = 450, 500, 1.0, oracle.SAR_RENDER_DEPTH
    st = oracle.start_points(g["seed"], 0, g["jobs"])
    rt = oracle.Runtime(450, 500)
    oracle.render_jobs(s, rt, st, g["iters_per_job"])
    assert int(rt.count[0, 0]) == g["count00"] and rt.max == g["max"]
    assert g["count00"] % 1 == 0 and g["count00"] >= g["iters_per_job"]  # whole diverged jobs land on pixel (0,0)
    assert f"{oracle.fnv1a64(rt.count):016x}" == g["count_fnv"]
    assert f"{oracle.fnv1a64(rt.zbuf):016x}" == g["zbuf_fnv"]
    assert f"{oracle.fnv1a64(rt.steps):016x}" == g["steps_fnv"]
    assert f"{oracle.fnv1a64(oracle.colorize(s, rt)):016x}" == g["depth_rgba_fnv"]
    # AdjustedVelocity(offset .8, factor -.2) is always negative -> palette entry 0 (monochrome)
    assert np.all(rt.steps[rt.zbuf != -1] < 0)


def test_start_point_stream(oracle):
    assert [hexes(r) for r in oracle.start_points(1, 0, 3)] == GOLD["start_points_seed1"]
    assert [hexes(r) for r in oracle.start_points(1, 5, 2)] == GOLD["start_points_seed1_skip5"]
    a = oracle.start_points(9, 0, 100)
    assert np.array_equal(a[40:60], oracle.start_points(9, 40, 20))
    assert a.min() >= 0.0 and a.max() < 0.1


def test_palette(oracle):
    import ctypes as C
    ps = oracle.poisson_saturne()
    for v, want in GOLD["palette_default"].items():
        rgb = np.empty(3)
        oracle.lib().sar_oracle_palette(C.byref(ps), float(v), rgb.ctypes.data_as(C.POINTER(C.c_double)))
        assert hexes(rgb) == want
    # clamps: < 0 -> entry 0; >= 1 -> 0.999999 (src/lib.rs:443-449)
    assert GOLD["palette_default"]["-0.5"] == GOLD["palette_default"]["0.0"]
    assert GOLD["palette_default"]["1.0"] == GOLD["palette_default"]["7.0"]


def test_merge_semantics(oracle):
    a, b = oracle.Runtime(4, 2), oracle.Runtime(4, 2)
    a.count[:] = [[1, 2, 0xFFFFFFFF, 0], [5, 6, 7, 8]]
    b.count[:] = [[10, 0, 2, 0], [1, 1, 1, 1]]
    a.zbuf[:] = [[0.5, -1.0, 0.25, -1.0], [0.1, 0.2, 0.3, 0.4]]
    b.zbuf[:] = [[0.5, 0.0, 0.1, -1.0], [0.2, 0.1, 0.3, 0.5]]
    a.steps[:] = 1.0
    b.steps[:] = 2.0
    a.set_max(3)
    assert oracle.merge(a, b) == 0
    assert a.count.tolist() == [[11, 2, 1, 0], [6, 7, 8, 9]]            # wrapping add
    assert a.max == 11                                                  # from merged counts only
    assert a.steps.tolist() == [[1.0, 2.0, 1.0, 1.0], [2.0, 1.0, 1.0, 2.0]]  # strict >, self wins ties
    assert oracle.merge(a, oracle.Runtime(3, 3)) != 0                   # dimension mismatch


def test_attractor_extent_matches_the_numbers_in_the_reference_source(oracle):
    """The only numbers the reference's source holds for this path: the comment at src/lib.rs:329-333 lists the extent
    of poisson-saturne ("xmin = -0.327770, xmax = 0.335278, ymin = -0.012949, ymax = 0.492107, zmin = -0.628829,
    zmax = 0.103010") — measured by the author in SCREEN space (raw coordinates are nowhere near). They pin the map
    coefficients, the un-normalised rotation matrix and `screen_space` of the oracle: 6.4e7 iterations reproduce every
    bound to better than 5e-5, from the inside (the author's run was longer; an extreme only ever grows)."""
    ref = np.array([-0.327770, 0.335278, -0.012949, 0.492107, -0.628829, 0.103010])
    cfg = oracle.poisson_saturne()
    e = oracle.extent(cfg, oracle.start_points(1, 0, 64), 1_000_000)
    sign = np.array([-1, 1, -1, 1, -1, 1])
    assert np.all(np.abs(e[:6] - ref) < 5e-5), e[:6]
    assert np.all((ref - e[:6]) * sign > -2e-6), "a bound lies outside the reference's"
    # the raw extent is a different box: the comment's numbers are not raw coordinates
    assert np.abs(e[6:] - ref).max() > 0.1


def test_threaded_oracle_is_the_sequential_oracle_bit_for_bit(oracle):
    """tests/oracle_lib.render_jobs_mt (what lets the full-size GPU frames meet the oracle in seconds): contiguous job
    slices on private runtimes folded with Runtime::merge in slice order == `jobs` sequential render calls."""
    for preset in (oracle.poisson_saturne, oracle.solar_sail):
        c = preset()
        c.width, c.height, c.scale = 160, 120, 1.0
        st = oracle.start_points(21, 0, 37)
        a, b = oracle.Runtime(160, 120), oracle.Runtime(160, 120)
        oracle.render_jobs(c, a, st, 3000)
        oracle.render_jobs_mt(c, b, st, 3000, 5)
        assert np.array_equal(a.count, b.count) and a.max == b.max
        assert np.array_equal(a.zbuf.view(np.uint32), b.zbuf.view(np.uint32))
        assert np.array_equal(a.steps.view(np.uint64), b.steps.view(np.uint64))


def test_fullsize_checksum_fixture_lists_every_case():
    import json
    import os
    import fullsize_cases as F
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fullsize_checksums.json")))
    assert set(g) == set(F.CASES)
    assert g["c2_131072"]["count_sum"] == 131072 * 7629 and g["c4_rank5_share"]["count_sum"] == 65536 * 19073
