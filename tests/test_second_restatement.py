"""The C oracle against tests/golden/second_restatement.py — an independent restatement of the reference in pure-Python
floats (no compiler: no FMA contraction, no re-association). Every comparison is bit for bit; the map is chaotic, so a
1-ulp disagreement anywhere shows up within a few dozen iterations."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import second_restatement as R  # noqa: E402


def _bits(v):
    return np.asarray(v, dtype=np.float64).view(np.uint64)


def _presets(oracle):
    return [(oracle.poisson_saturne(), R.POISSON), (oracle.solar_sail(), R.SOLAR)]


def test_next_point_agrees_over_long_trajectories(oracle):
    for cfg, pre in _presets(oracle):
        for p0 in ((0.05, 0.031, 0.077), (0.025, 0.0155, 0.0385)):
            p = p0
            for n in (1, 10, 1000, 20000):
                q = p0
                for _ in range(n):
                    q = R.next_point(pre, q)
                got = oracle.iterate(cfg, np.array(p0), n)
                assert np.array_equal(_bits(got), _bits(q)) or (np.isnan(got).all() and all(v != v for v in q)), (n, p0)
            del p


def test_rotation_matrix_and_transforms_agree(oracle):
    rng = np.random.default_rng(7)
    for cfg, pre in _presets(oracle):
        assert np.array_equal(_bits(oracle.rotation_matrix(cfg).ravel()), _bits(np.array(R.rotation_matrix(pre)).ravel()))
        for _ in range(2000):
            d = rng.uniform(-0.6, 0.6, 3)
            s = rng.uniform(-0.8, 0.8, 3)
            got = oracle.lib().sar_oracle_color_transform(oracle.C.byref(cfg), oracle._dptr(d), oracle._dptr(s))
            assert _bits([got])[0] == _bits([R.color_transform(pre, tuple(d), tuple(s))])[0]
    cfg = oracle.poisson_saturne()
    for v in list(rng.uniform(-0.3, 1.3, 2000)) + [0.0, 1.0, 0.999999, 1.0 / 6.0, 0.5]:
        rgb = np.empty(3)
        oracle.lib().sar_oracle_palette(oracle.C.byref(cfg), float(v), oracle._dptr(rgb))
        assert np.array_equal(_bits(rgb), _bits(R.palette_interpolate(R.DEFAULT_PALETTE, float(v))))


@pytest.mark.parametrize("which", [0, 1])
def test_whole_render_and_colorize_agree(oracle, which):
    """render (bounds test incl. NaN, count, strict depth test, payload, previous_point on skips), merge and both
    colorize kinds on a small frame — with a diverging start point for solar-sail."""
    cfg, pre = _presets(oracle)[which]
    W, H, n = 72, 56, 2500
    cfg.width, cfg.height, cfg.scale, cfg.angle = W, H, 1.0, 0.7
    starts = oracle.start_points(33, 0, 3)
    a = oracle.Runtime(W, H)
    b = R.Runtime(W, H)
    for p0 in starts:
        oracle.render(cfg, a, p0, n)
        R.render(pre, b, tuple(float(v) for v in p0), n, angle=0.7, scale=1.0)
    assert np.array_equal(a.count.ravel(), np.array(b.count, dtype=np.uint32)) and a.max == b.max
    assert np.array_equal(a.zbuf.ravel().view(np.uint32), np.array(b.zbuf, dtype=np.float32).view(np.uint32))
    assert np.array_equal(_bits(a.steps.ravel()), _bits(b.steps))
    for transparent in (0, 1):
        cfg.render_kind, cfg.transparent = oracle.SAR_RENDER_GAS, transparent
        assert np.array_equal(oracle.colorize(cfg, a).reshape(-1, 4), np.array(R.colorize_gas(b, transparent=bool(transparent)), dtype=np.uint16))
    cfg.render_kind = oracle.SAR_RENDER_DEPTH
    assert np.array_equal(oracle.colorize(cfg, a).reshape(-1, 4), np.array(R.colorize_depth(b), dtype=np.uint16))
    # merge: two halves rendered apart, folded with Runtime::merge
    a1, a2, b1, b2 = oracle.Runtime(W, H), oracle.Runtime(W, H), R.Runtime(W, H), R.Runtime(W, H)
    oracle.render(cfg, a1, starts[0], n); oracle.render(cfg, a2, starts[1], n)  # noqa: E702
    R.render(pre, b1, tuple(float(v) for v in starts[0]), n, angle=0.7, scale=1.0)
    R.render(pre, b2, tuple(float(v) for v in starts[1]), n, angle=0.7, scale=1.0)
    assert oracle.merge(a1, a2) == 0
    b1.merge(b2)
    assert np.array_equal(a1.count.ravel(), np.array(b1.count, dtype=np.uint32)) and a1.max == b1.max
    assert np.array_equal(_bits(a1.steps.ravel()), _bits(b1.steps))
