"""A SECOND, independent restatement of the reference's arithmetic for this path — test infrastructure, like oracle/.

Written from the reference source alone (Icelk/strange-attractor-renderer `src/lib.rs`, cited per function), NOT from
oracle/sar_oracle.c: pure-Python floats are IEEE binary64 with one rounding per operation — no compiler, no
contraction, no re-association, no vectoriser — so an oracle that was FMA-contracted or re-associated by its C compiler
disagrees with this file after a handful of iterations (the map is chaotic). tests/golden/make_golden.py cross-checks
the C oracle against it over >= 1e5 iterations before it freezes the known-answer vectors, and
tests/test_second_restatement.py repeats a shorter cross-check in every CPU run.

Deliberately naive: lists and loops, one statement per reference statement. numpy is used for exactly one thing, the
f64 -> f32 rounding of `z2 as f32`.
"""
import math

import numpy as np

# ---- presets: reference src/lib.rs:310-387 (values copied as data) --------------------------------------------------
POISSON = dict(
    x=[0.021, 1.182, -1.183, 0.128, -1.12, -0.641, -1.152, -0.834, -0.97, 0.722],
    y=[0.243038, -0.825, -1.2, -0.835443, -0.835443, -0.364557, 0.458, 0.622785, -0.394937, -1.032911],
    z=[-0.455696, 0.673, 0.915, -0.258228, -0.495, -0.264, -0.432, -0.416, -0.877, -0.3],
    center_camera=(-0.005, 0.262, -0.366 + 0.12),
    axis=(0.304289493528802, 0.760492682863655, 0.573636455813981), rotation=1.78268191887446, scale=1.0,
    transform=("poisson_saturne",),
)
SOLAR = dict(
    x=[0.744304, -0.546835, 0.121519, -0.653165, 0.399, 0.379, 0.44, 1.014, -0.805063, 0.377],
    y=[-0.683, 0.531646, -0.04557, -1.2, -0.546835, 0.091139, 0.744304, -0.273418, -0.349367, -0.531646],
    z=[0.712, 0.744304, -0.577215, 0.966, 0.04557, 1.063291, 0.01519, -0.425316, 0.212658, -0.01519],
    center_camera=(0.28, -0.12, 0.22), axis=(0.02466, 0.4618, -0.54789), rotation=2.2195, scale=1.7,
    transform=("adjusted_velocity", 0.8, -0.2),          # offset, factor (:381-384)
)
DEFAULT_PALETTE = [(1.0, 1.0, 0.5), (0.5, 1.0, 0.5), (1.0, 0.5, 0.5), (0.5, 1.0, 1.0), (0.5, 0.5, 1.0), (1.0, 0.5, 1.0)]
BRIGHTNESS = (-0.15, 5.0 / 3.0)                           # BrighnessConstants::default (:397-404)


# ---- PolynomialSprott2Degree::next_point, :583-621 -------------------------------------------------------------------
def sum_coefficients(polynomials, coefficients):          # :588-600
    s = 0.0
    for i in range(10):
        s += polynomials[i] * coefficients[i]
    return s


def next_point(preset, p):
    x, y, z = p
    monoms = [1.0, x, x * x, x * y, x * z, y, y * y, y * z, z, z * z]   # :602-613 (`square` is `self * self`, :92-94)
    return (sum_coefficients(monoms, preset["x"]), sum_coefficients(monoms, preset["y"]),
            sum_coefficients(monoms, preset["z"]))


# ---- EulerAxisRotation::to_rotation_matrix (release: axis NOT normalised), :176-196; mul_right, :205-216 ------------
def rotation_matrix(preset):
    x, y, z = preset["axis"]
    rotation = preset["rotation"]
    c = math.cos(rotation)
    c1 = 1.0 - c
    s = math.sin(rotation)
    return [[c + x * x * c1, x * y * c1 - z * s, x * z * c1 + y * s],
            [y * x * c1 + z * s, c + y * y * c1, y * z * c1 - x * s],
            [z * x * c1 - y * s, z * y * c1 + x * s, c + z * z * c1]]


def mul_right(m, v):
    return (m[0][0] * v[0] + m[0][1] * v[1] + m[0][2] * v[2],
            m[1][0] * v[0] + m[1][1] * v[1] + m[1][2] * v[2],
            m[2][0] * v[0] + m[2][1] * v[1] + m[2][2] * v[2])


# ---- colour transforms, :507-516 and :520-558 -------------------------------------------------------------------------
def magnitude(v):                                          # :129-131
    return math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def color_transform(preset, delta, screen_space):
    t = preset["transform"]
    if t[0] == "adjusted_velocity":
        return (magnitude(delta) + t[1]) * t[2]            # :514
    COS = 0.7009092642998508981833083453238941729068756103515625      # :529-531
    SIN = 0.7132504491541815649924274111981503665447235107421875      # :535-537
    cc = preset["center_camera"]
    p = screen_space
    x2 = (p[0] + cc[0]) * COS + (p[2] + cc[1]) * SIN       # :538-539
    if x2 < -0.0839 or 10.55 * x2 + p[1] < 0.46 - 1.0941 or 1.0426 * x2 + p[1] < 0.179 - 0.1576 \
            or 0.5139 * x2 - p[1] > -0.04 - 0.04092:       # :542-546
        part = 0.0
    else:
        part = 1.0
    color = (part + magnitude(delta)) / 2.0                # :556
    return (color - 0.1) / 0.9                             # :557


# ---- Rust `as` casts ----------------------------------------------------------------------------------------------------
def as_u32(v):
    if v != v or v <= 0.0:
        return 0
    return 4294967295 if v >= 4294967295.0 else int(v)


def as_u16(v):
    if v != v or v <= 0.0:
        return 0
    return 65535 if v >= 65535.0 else int(v)


def as_f32(v):
    with np.errstate(over="ignore", invalid="ignore"):
        return np.float32(v)


# ---- Runtime + render, :631-838 -----------------------------------------------------------------------------------------
class Runtime:
    def __init__(self, width, height):
        self.width, self.height = width, height
        self.reset()

    def reset(self):                                       # :682-699
        n = self.width * self.height
        self.count = [0] * n
        self.steps = [0.0] * n
        self.zbuf = [np.float32(-1.0)] * n
        self.max = 0

    def merge(self, other):                                # :708-738
        assert (self.width, self.height) == (other.width, other.height)
        for k in range(self.width * self.height):
            self.count[k] = (self.count[k] + other.count[k]) & 0xFFFFFFFF
            if self.count[k] > self.max:
                self.max = self.count[k]
            if other.zbuf[k] > self.zbuf[k]:
                self.steps[k] = other.steps[k]
                self.zbuf[k] = other.zbuf[k]


def render(preset, rt, p0, iterations, angle=0.0, scale=None):
    scale = preset["scale"] if scale is None else scale
    initial_point = p0                                     # `rng.random::<Vec3>() * 0.1` is the caller's (:748)
    for _ in range(1000):                                  # :750-752
        initial_point = next_point(preset, initial_point)
    m = rotation_matrix(preset)                            # :755
    sin_v = math.sin(angle)
    cos_v = math.cos(angle)
    cc = preset["center_camera"]
    width = float(rt.width)
    height = float(rt.height)
    width_scaled = width * scale
    scale_adjusted_mid = 0.5 / scale
    previous_point = initial_point
    current_point = initial_point
    for _ in range(iterations):                            # :769
        current_point = next_point(preset, current_point)
        ss = mul_right(m, current_point)                   # :773
        x2 = (ss[0] + cc[0]) * cos_v + (ss[2] + cc[1]) * sin_v      # :776-777
        z2 = (ss[0] + cc[0]) * sin_v - (ss[2] + cc[1]) * cos_v      # :778-779
        i = (scale_adjusted_mid - x2) * width_scaled       # :783
        j = height / 2.0 - (ss[1] + cc[2]) * width_scaled  # :786
        if i >= width or j >= height or i < 0.0 or j < 0.0:          # :789 (NaN passes)
            previous_point = current_point                 # :793
            continue
        iu, ju = as_u32(i), as_u32(j)                      # :800-802
        idx = ju * rt.width + iu
        rt.count[idx] = (rt.count[idx] + 1) & 0xFFFFFFFF   # :811 (wrapping in release builds)
        if rt.count[idx] > rt.max:
            rt.max = rt.count[idx]
        zf = as_f32(z2)
        if zf > rt.zbuf[idx]:                              # :821
            delta = (current_point[0] - previous_point[0], current_point[1] - previous_point[1],
                     current_point[2] - previous_point[2])
            rt.steps[idx] = color_transform(preset, delta, ss)
            rt.zbuf[idx] = zf
        previous_point = current_point                     # :836


# ---- Palette::interpolate, :442-472; colorize, :841-904 --------------------------------------------------------------------
def palette_interpolate(entries, value):
    lst = list(entries) + [entries[-1]]                    # Palette::new duplicates the last entry (:416-418)
    count_f64 = float(len(lst) - 1)
    if value < 0.0:
        value = 0.0
    elif value >= 1.0:
        value = 0.999999
    value = value * count_f64
    n = as_u32(math.floor(value))
    sub_n_offset = math.fmod(value, 1.0)
    sub_n_offset_1 = 1.0 - sub_n_offset
    r1, g1, b1 = lst[n]
    r2, g2, b2 = lst[n + 1]
    return (math.sqrt(r2 * sub_n_offset + r1 * sub_n_offset_1), math.sqrt(g2 * sub_n_offset + g1 * sub_n_offset_1),
            math.sqrt(b2 * sub_n_offset + b1 * sub_n_offset_1))


def _ln(v):
    return math.log(v) if v > 0.0 else (float("-inf") if v == 0.0 else float("nan"))


def _div(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        return float("nan") if (a == 0.0 or a != a) else math.copysign(float("inf"), a) * math.copysign(1.0, b)


def colorize_gas(rt, entries=DEFAULT_PALETTE, brightness=BRIGHTNESS, transparent=True):
    out = []
    offset, factor_b = brightness
    for steps, count in zip(rt.steps, rt.count):
        r, g, b = palette_interpolate(entries, steps)
        factor = _div(_ln(float(count + 1)), _ln(float(rt.max + 1)))     # f64::log(self, base) = ln / ln (:860)
        out.append((as_u16((r * factor + offset) * factor_b * 65535.0), as_u16((g * factor + offset) * factor_b * 65535.0),
                    as_u16((b * factor + offset) * factor_b * 65535.0), as_u16(factor * 65535.0) if transparent else 65535))
    return out


def colorize_depth(rt):
    mx, mn = np.float32(0.0), np.float32(np.finfo(np.float32).max)        # fold seeds (:877-882)
    for z in rt.zbuf:
        if z != np.float32(-1.0):
            mx = max(mx, z)
            mn = min(mn, z)
    with np.errstate(all="ignore"):
        diff = np.float32(mx - mn)
        out = []
        for z in rt.zbuf:
            zz = np.float32(0.0) if z == np.float32(-1.0) else np.float32(np.float32(z - mn) / diff)
            v = as_u16(float(np.float32(zz * np.float32(65535.0))))
            out.append((v, v, v, 65535))
    return out
