"""Generates the committed fixtures under tests/golden/ (run in the build container; never on the GPU box).

  kat.json            known-answer vectors of the CPU oracle. Nothing in the reference's own test-suite
                      pins this path (its only test is a doc-test that constructs a Config), so these are
                      oracle-emitted — after the oracle has been cross-checked, bit for bit, against the
                      committed independent restatement in pure-Python floats (second_restatement.py: >= 1e5
                      iterations of next_point per preset, the rotation matrices, both colour transforms,
                      Palette::interpolate and whole small renders). The values listed in SURVEY.md §8c (from
                      the survey session's throw-away restatement, gcc AND AMD-clang) are asserted as well.
  ref_png_stats.json  statistics of the one artefact the reference ships, media/poisson-saturne.png
                      (README.md:72-73: `-i1000000000 -b -0.25`, 1920x1080, OS-random seed): bounding box,
                      non-zero fraction and a 96x54 block-mean thumbnail per channel. Data derived from a
                      data file; used for the statistical pin of the oracle.

    python tests/golden/make_golden.py [--png /root/reference/media/poisson-saturne.png]
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import oracle_lib as O  # noqa: E402


def hexes(v):
    return [float.hex(float(x)) for x in v]


def cross_check_against_second_restatement(iterations=120_000):
    """The KATs below are oracle-emitted; before they are frozen the oracle must agree, bit for bit, with the independent
    pure-Python restatement (tests/golden/second_restatement.py: no compiler, no contraction) over >= 1e5 iterations of
    next_point per preset, the projection (through a render), both colour transforms and Palette::interpolate."""
    import second_restatement as R
    rng = np.random.default_rng(11)
    for cfg, pre, p0 in ((O.poisson_saturne(), R.POISSON, (0.05, 0.031, 0.077)), (O.solar_sail(), R.SOLAR, (0.025, 0.0155, 0.0385))):
        q = p0
        for k in range(iterations):
            q = R.next_point(pre, q)
            if k + 1 in (1, 1000, 10_000, iterations):
                got = O.iterate(cfg, np.array(p0), k + 1)
                assert got.view(np.uint64).tolist() == np.array(q).view(np.uint64).tolist(), (k, got, q)
        assert np.array_equal(O.rotation_matrix(cfg).ravel().view(np.uint64), np.array(R.rotation_matrix(pre)).ravel().view(np.uint64))
        for _ in range(5000):
            d, s = rng.uniform(-0.6, 0.6, 3), rng.uniform(-0.8, 0.8, 3)
            got = O.lib().sar_oracle_color_transform(O.C.byref(cfg), O._dptr(d), O._dptr(s))
            assert np.float64(got).view(np.uint64) == np.float64(R.color_transform(pre, tuple(d), tuple(s))).view(np.uint64)
        # projection + bounds + depth test + payload: a whole (small) render
        cfg.width, cfg.height, cfg.scale = 64, 48, 1.0
        a, b = O.Runtime(64, 48), R.Runtime(64, 48)
        O.render(cfg, a, np.array(p0), 20_000)
        R.render(pre, b, p0, 20_000, angle=0.0, scale=1.0)
        assert np.array_equal(a.count.ravel(), np.array(b.count, dtype=np.uint32)) and a.max == b.max
        assert np.array_equal(a.zbuf.ravel().view(np.uint32), np.array(b.zbuf, dtype=np.float32).view(np.uint32))
        assert np.array_equal(a.steps.ravel().view(np.uint64), np.array(b.steps).view(np.uint64))
    cfg = O.poisson_saturne()
    for v in rng.uniform(-0.3, 1.3, 5000):
        rgb = np.empty(3)
        O.lib().sar_oracle_palette(O.C.byref(cfg), float(v), O._dptr(rgb))
        assert rgb.view(np.uint64).tolist() == np.array(R.palette_interpolate(R.DEFAULT_PALETTE, float(v))).view(np.uint64).tolist()


def kat():
    cross_check_against_second_restatement()
    out = {}
    ps, ss = O.poisson_saturne(), O.solar_sail()
    p0 = np.array([0.05, 0.031, 0.077])
    out["poisson_iter"] = {str(n): hexes(O.iterate(ps, p0, n)) for n in (1, 1000, 1001000)}
    q0 = np.array([0.025, 0.0155, 0.0385])
    out["solar_iter"] = {str(n): hexes(O.iterate(ss, q0, n)) for n in (1, 1000)}
    out["poisson_matrix"] = hexes(O.rotation_matrix(ps).ravel())
    out["solar_matrix"] = hexes(O.rotation_matrix(ss).ravel())
    # SURVEY.md §8c session values (independent restatement) — must agree
    assert out["poisson_iter"]["1"] == ["-0x1.4d66b288f62a8p-6", "0x1.2fdb737366f52p-3", "-0x1.007e66211707fp-1"]
    assert out["poisson_iter"]["1001000"] == ["0x1.3f5a9c613ab36p-2", "0x1.d454faad0974fp-3", "-0x1.071b380ea3f9bp-3"]
    assert out["solar_iter"]["1000"] == ["0x1.09450fdb8bfb7p-4", "-0x1.2b75c30ac8523p-3", "0x1.4ce662c2471fcp-3"]
    assert out["poisson_matrix"][0] == "-0x1.926334438a872p-4" and out["poisson_matrix"][8] == "0x1.80efcaa45292bp-3"
    assert out["solar_matrix"][0] == "-0x1.34d3ccaa576adp-1"

    c = O.poisson_saturne()
    c.width = c.height = 512
    rt = O.Runtime(512, 512)
    O.render(c, rt, p0, 10_000_000)
    out["c1_512"] = {
        "p0": hexes(p0), "iterations": 10_000_000,
        "in_bounds": int(rt.count.sum()), "touched": int((rt.count > 0).sum()), "max": rt.max,
        "depth_set": int((rt.zbuf != -1).sum()),
        "count_fnv": f"{O.fnv1a64(rt.count):016x}", "zbuf_fnv": f"{O.fnv1a64(rt.zbuf):016x}",
        "steps_fnv": f"{O.fnv1a64(rt.steps):016x}",
    }
    assert out["c1_512"]["count_fnv"] == "52a35a7e05fd92db" and out["c1_512"]["max"] == 5673
    assert out["c1_512"]["zbuf_fnv"] == "b69ddb65be9314f5" and out["c1_512"]["steps_fnv"] == "21fff5b92fa0204d"
    c.transparent = 0
    out["c1_512"]["rgba_fnv"] = f"{O.fnv1a64(O.colorize(c, rt)):016x}"
    c.transparent = 1
    out["c1_512"]["rgba_transparent_fnv"] = f"{O.fnv1a64(O.colorize(c, rt)):016x}"
    rt.reset()
    starts = np.stack([p0 * (k + 1) / 4 for k in range(4)])
    O.render_jobs(c, rt, starts, 2_500_000)
    out["c1_512_4jobs"] = {"touched": int((rt.count > 0).sum()), "max": rt.max,
                           "count_fnv": f"{O.fnv1a64(rt.count):016x}", "zbuf_fnv": f"{O.fnv1a64(rt.zbuf):016x}",
                           "steps_fnv": f"{O.fnv1a64(rt.steps):016x}"}
    assert out["c1_512_4jobs"]["count_fnv"] == "ae9cc71e6ceb683a" and out["c1_512_4jobs"]["max"] == 5743

    # solar-sail, depth path, with diverging jobs (seeded stream)
    s = O.solar_sail()
    s.width, s.height, s.scale, s.render_kind = 450, 500, 1.0, O.SAR_RENDER_DEPTH
    st = O.start_points(2024, 0, 64)
    rs = O.Runtime(450, 500)
    O.render_jobs(s, rs, st, 20000)
    out["solar_450x500_64jobs"] = {
        "seed": 2024, "jobs": 64, "iters_per_job": 20000, "count00": int(rs.count[0, 0]), "max": rs.max,
        "count_fnv": f"{O.fnv1a64(rs.count):016x}", "zbuf_fnv": f"{O.fnv1a64(rs.zbuf):016x}",
        "steps_fnv": f"{O.fnv1a64(rs.steps):016x}", "depth_rgba_fnv": f"{O.fnv1a64(O.colorize(s, rs)):016x}",
    }
    # start-point stream
    out["start_points_seed1"] = [hexes(r) for r in O.start_points(1, 0, 3)]
    out["start_points_seed1_skip5"] = [hexes(r) for r in O.start_points(1, 5, 2)]
    # palette / colour transform spot values
    pal = {}
    for v in (-0.5, 0.0, 0.123456789, 0.5, 0.999999, 1.0, 7.0):
        rgb = np.empty(3)
        O.lib().sar_oracle_palette(O.C.byref(ps), v, rgb.ctypes.data_as(O.C.POINTER(O.C.c_double)))
        pal[repr(v)] = hexes(rgb)
    out["palette_default"] = pal
    return out


def decode_png16(path):
    """PIL flattens 16-bit RGB to 8: use the tests' own minimal decoder."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from image_decode import decode_png
    return decode_png(path)


def png_stats(path):
    im = decode_png16(path)
    h, w = im.shape[:2]
    rgb = im[..., :3].astype(np.float64)
    nz = rgb.sum(axis=2) > 0
    ys, xs = np.where(nz)
    bh, bw = 20, 20
    thumb = rgb.reshape(h // bh, bh, w // bw, bw, 3).mean(axis=(1, 3))
    return {
        "source": "reference media/poisson-saturne.png (README.md:72-73: -i1000000000 -b -0.25)",
        "width": int(w), "height": int(h), "mode": str(im.dtype),
        "bbox_x": [int(xs.min()), int(xs.max())], "bbox_y": [int(ys.min()), int(ys.max())],
        "nonzero_fraction": float(nz.mean()),
        "channel_mean": [float(v) for v in rgb.mean(axis=(0, 1))],
        "thumb_block": [bh, bw],
        "thumb": np.round(thumb).astype(int).tolist(),
    }


if __name__ == "__main__":
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat(), f, indent=1)
    png = "/root/reference/media/poisson-saturne.png"
    if "--png" in sys.argv:
        png = sys.argv[sys.argv.index("--png") + 1]
    if os.path.exists(png):
        with open(os.path.join(HERE, "ref_png_stats.json"), "w") as f:
            json.dump(png_stats(png), f)
    print("wrote fixtures to", HERE)
