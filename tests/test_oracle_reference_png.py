"""CPU: the statistical pin of the oracle against the ONE artefact the reference ships.

media/poisson-saturne.png was rendered by the reference CLI with `-i1000000000 -b -0.25` at 1920x1080
(README.md:72-73) from an OS-random seed, so only statistics can be compared: the support of the image,
its orientation and its per-channel block means (tests/golden/ref_png_stats.json, derived from the PNG by
tests/golden/make_golden.py). The CLI writes RGB16 with `transparent=false` and scale 1."""
import json
import os

import numpy as np

STATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_png_stats.json")))


def test_oracle_image_matches_reference_png_statistics(oracle):
    c = oracle.poisson_saturne()
    c.width, c.height = STATS["width"], STATS["height"]
    c.iterations = 1_000_000_000
    c.brightness_offset = -0.25      # CLI -b replaces only the offset (src/bin/main.rs:423-431)
    c.transparent = 0
    threads = min(8, len(os.sched_getaffinity(0)))
    secs, done, img = oracle.render_parallel(c, threads, 12, 20240928)
    assert done > 0.999e9
    rgb = img[..., :3].astype(np.float64)
    nz = rgb.sum(axis=2) > 0
    ys, xs = np.where(nz)
    assert abs(int(xs.min()) - STATS["bbox_x"][0]) <= 3 and abs(int(xs.max()) - STATS["bbox_x"][1]) <= 3
    assert abs(int(ys.min()) - STATS["bbox_y"][0]) <= 3 and abs(int(ys.max()) - STATS["bbox_y"][1]) <= 3
    assert abs(float(nz.mean()) - STATS["nonzero_fraction"]) < 2e-3
    bh, bw = STATS["thumb_block"]
    h, w = rgb.shape[:2]
    thumb = rgb.reshape(h // bh, bh, w // bw, bw, 3).mean(axis=(1, 3))
    ref = np.asarray(STATS["thumb"], dtype=np.float64)
    for ch in range(3):
        corr = np.corrcoef(thumb[..., ch].ravel(), ref[..., ch].ravel())[0, 1]
        assert corr > 0.998, (ch, corr)
        # orientation: mirrored variants must NOT match
        assert np.corrcoef(thumb[:, ::-1, ch].ravel(), ref[..., ch].ravel())[0, 1] < 0.6
        assert np.corrcoef(thumb[::-1, :, ch].ravel(), ref[..., ch].ravel())[0, 1] < 0.6
        assert abs(rgb[..., ch].mean() / STATS["channel_mean"][ch] - 1.0) < 0.01
