"""Image export row (reference src/bin/main.rs:40-100, write_image_matches): format choice, the
RGBA16 -> RGB16 / RGBA8 / RGB8 conversion and the PNG / BMP / PAM encoders.

The conversions and encoders of the reference live in the `image` crate (0.25, unpinned, not vendored): parity is
UNPINNED beyond "the file decodes to the same samples" — which is what these tests check, with the tests' own decoders
and with PIL where PIL can represent the image. CPU tests exercise the host-only encoders and the oracle; GPU tests hold
the device conversion to the oracle bit for bit."""
import os

import numpy as np
import pytest

import image_decode as D

FMT = {"rgba16": 0, "rgb16": 1, "rgba8": 2, "rgb8": 3}


def _random_rgba16(h, w, seed=3):
    rng = np.random.default_rng(seed)
    im = rng.integers(0, 65536, size=(h, w, 4), dtype=np.uint16)
    # the corners of the 16 -> 8 bit rounding rule: every multiple of 257 +- 128/129
    edge = np.array([0, 127, 128, 129, 256, 257, 385, 386, 65278, 65406, 65407, 65408, 65534, 65535], dtype=np.uint16)
    flat = im.reshape(-1)
    k = min(edge.size, flat.size)
    flat[:k] = edge[:k]
    return im


def test_format_choice_follows_the_cli(sar):
    # src/bin/main.rs:52-57
    assert sar.image_format(True, False) == sar.SAR_FMT_RGBA16
    assert sar.image_format(False, False) == sar.SAR_FMT_RGB16
    assert sar.image_format(True, True) == sar.SAR_FMT_RGBA8
    assert sar.image_format(False, True) == sar.SAR_FMT_RGB8
    lib = sar.load_library()
    assert lib.sar_image_bytes(sar.SAR_FMT_RGB8, 2048, 2048) == 2048 * 2048 * 3
    assert lib.sar_image_bytes(sar.SAR_FMT_RGBA16, 3, 5) == 3 * 5 * 8
    assert lib.sar_image_bytes(9, 3, 5) == 0


def test_oracle_16_to_8_bit_rule_is_the_rounded_inverse_of_times_257(oracle):
    """image 0.25 `FromPrimitive<u16> for u8`: ((c + 128) / 257) — exhaustive: it inverts c8 * 257 exactly and is
    round-to-nearest of c * 255 / 65535."""
    c = np.arange(65536, dtype=np.uint16)
    im = np.zeros((256, 256, 4), dtype=np.uint16)
    im[..., 0] = c.reshape(256, 256)
    got = oracle.convert(FMT["rgba8"], im)[..., 0].reshape(-1).astype(np.int64)
    np.testing.assert_array_equal(got, np.floor(c.astype(np.float64) * 255.0 / 65535.0 + 0.5).astype(np.int64))
    c8 = np.arange(256, dtype=np.int64)
    np.testing.assert_array_equal(got[c8 * 257], c8)
    # alpha is dropped, not pre-multiplied; 16-bit channels pass through
    im = _random_rgba16(5, 7)
    np.testing.assert_array_equal(oracle.convert(FMT["rgb16"], im), im[..., :3])
    np.testing.assert_array_equal(oracle.convert(FMT["rgba16"], im), im)
    np.testing.assert_array_equal(oracle.convert(FMT["rgb8"], im), oracle.convert(FMT["rgba8"], im)[..., :3])


@pytest.mark.parametrize("size", [(1, 1), (3, 5), (53, 97), (64, 64)])
@pytest.mark.parametrize("fmt", ["rgba16", "rgb16", "rgba8", "rgb8"])
def test_png_decodes_to_the_same_samples(sar, oracle, tmp_path, fmt, size):
    h, w = size
    img = oracle.convert(FMT[fmt], _random_rgba16(h, w, seed=h * 131 + w))
    path = str(tmp_path / f"a_{fmt}.png")
    sar.write_image(img, path, "png")
    back = D.decode_png(path)
    assert back.dtype == img.dtype and back.shape == img.shape
    np.testing.assert_array_equal(back, img)
    if img.dtype == np.uint8:  # second opinion
        from PIL import Image
        np.testing.assert_array_equal(np.asarray(Image.open(path)), img)


def test_png_adaptive_filter_compresses_smooth_images(sar, tmp_path):
    """A horizontal + vertical gradient: the adaptive filter (main.rs:89) must beat 'no filter' by a wide margin."""
    y, x = np.mgrid[0:256, 0:256]
    img = np.stack([(x * 257) & 0xFFFF, (y * 257) & 0xFFFF, ((x + y) * 128) & 0xFFFF], axis=2).astype(np.uint16)
    path = str(tmp_path / "grad.png")
    sar.write_image(img, path, "png")
    np.testing.assert_array_equal(D.decode_png(path), img)
    import zlib
    unfiltered = len(zlib.compress(b"".join(b"\x00" + r.astype(">u2").tobytes() for r in img), 6))
    assert os.path.getsize(path) < unfiltered / 4


@pytest.mark.parametrize("size", [(1, 1), (3, 5), (53, 97)])
@pytest.mark.parametrize("fmt", ["rgba8", "rgb8"])
def test_bmp_and_pam_decode_to_the_same_samples(sar, oracle, tmp_path, fmt, size):
    h, w = size
    img = oracle.convert(FMT[fmt], _random_rgba16(h, w, seed=7))
    bmp, pam = str(tmp_path / "a.bmp"), str(tmp_path / "a.pam")
    sar.write_image(img, bmp, "bmp")
    sar.write_image(img, pam, "pam")
    np.testing.assert_array_equal(D.decode_bmp(bmp), img)
    np.testing.assert_array_equal(D.decode_pam(pam), img)
    from PIL import Image
    pil = np.asarray(Image.open(bmp).convert("RGBA" if fmt == "rgba8" else "RGB"))
    np.testing.assert_array_equal(pil, img)


def test_encoder_errors_are_status_codes(sar, oracle, tmp_path):
    img16 = oracle.convert(FMT["rgb16"], _random_rgba16(4, 4))
    with pytest.raises(sar.SarError):  # --bmp / --pam require --8bit (main.rs:256-258)
        sar.write_image(img16, str(tmp_path / "x.bmp"), "bmp")
    with pytest.raises(sar.SarError):
        sar.write_image(img16, str(tmp_path / "x.pam"), "pam")
    with pytest.raises(sar.SarError) as e:  # File::create(..).unwrap() panics in the reference (main.rs:103)
        sar.write_image(img16, str(tmp_path / "no_such_dir" / "x.png"), "png")
    assert "cannot write" in str(e.value)


# ---- device side ----------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 1), (3, 5), (97, 53), (512, 512)])
def test_device_conversion_matches_oracle(sar, oracle, gpu, size):
    """k_convert against the oracle for every format, including pixel counts that are not a multiple of four."""
    import torch
    w, h = size
    cfg = sar.Config.poisson_saturne(iterations=1000, width=w, height=h, jobs_total=1)
    rt = sar.Runtime(cfg)
    im = _random_rgba16(h, w, seed=w * 7 + h)
    src = torch.from_numpy(im.view(np.int16)).cuda()
    for name, fmt in FMT.items():
        ch, dt = {0: (4, torch.int16), 1: (3, torch.int16), 2: (4, torch.uint8), 3: (3, torch.uint8)}[fmt]
        out = torch.zeros((h, w, ch), dtype=dt, device="cuda")
        torch.cuda.synchronize()  # src / out were produced on torch's stream, the conversion runs on the runtime's
        sar.convert_device(rt, src.data_ptr(), fmt, out.data_ptr())
        rt.synchronize()
        got = out.cpu().numpy()
        got = got.view(np.uint16) if dt == torch.int16 else got
        np.testing.assert_array_equal(got, oracle.convert(fmt, im), err_msg=name)


@pytest.mark.gpu
def test_write_image_matches_end_to_end(sar, oracle, gpu, tmp_path):
    """render -> colorize -> convert on the device -> encode, for the four (transparent, 8bit) cases and the three
    encoders; every file decodes to the oracle's colorize + convert."""
    jobs, n, w, h = 512, 400, 160, 120
    for transparent in (False, True):
        cfg = sar.Config.poisson_saturne(iterations=jobs * n, width=w, height=h, jobs_total=jobs, transparent=int(transparent))
        st = sar.start_points(5, 0, jobs)
        rt, ort = sar.Runtime(cfg), oracle.Runtime(w, h)
        sar.render_jobs(cfg, rt, st)
        oracle.render_jobs(cfg.c, ort, st, n)
        ref16 = oracle.colorize(cfg.c, ort)
        for eight in (False, True):
            fmt = sar.image_format(transparent, eight)
            want = oracle.convert(fmt, ref16)
            np.testing.assert_array_equal(sar.colorize_format(cfg, rt, fmt), want)
            path = sar.write_image_matches(cfg, rt, str(tmp_path / f"img_{int(transparent)}{int(eight)}.xyz"), transparent, eight)
            assert path.endswith(".png")
            np.testing.assert_array_equal(D.decode_png(path), want)
            if eight:
                np.testing.assert_array_equal(D.decode_bmp(sar.write_image_matches(cfg, rt, path, transparent, True, bmp=True)), want)
                np.testing.assert_array_equal(D.decode_pam(sar.write_image_matches(cfg, rt, path, transparent, True, pam=True)), want)
            else:
                with pytest.raises(ValueError):
                    sar.write_image_matches(cfg, rt, path, transparent, False, bmp=True)


REF_PNG = "/root/reference/media/poisson-saturne.png"


@pytest.mark.skipif(not os.path.exists(REF_PNG), reason="the reference checkout is only present in the build container")
def test_reencoding_the_reference_png_gives_the_same_samples_and_file_size(sar, tmp_path):
    """The one PNG the reference ships, decoded and encoded again by sar_write_png: identical samples, and a file within
    0.5 % of the original's size — i.e. the same encoder settings (default compression, per-row adaptive filter,
    src/bin/main.rs:84-92); the bytes themselves differ because the crate deflates with miniz_oxide, not zlib."""
    im = D.decode_png(REF_PNG)
    assert im.shape == (1080, 1920, 3) and im.dtype == np.uint16
    out = str(tmp_path / "again.png")
    sar.write_image(im, out, "png")
    np.testing.assert_array_equal(D.decode_png(out), im)
    ref_size, size = os.path.getsize(REF_PNG), os.path.getsize(out)
    assert abs(size / ref_size - 1.0) < 0.005, (size, ref_size)


@pytest.mark.gpu
def test_conversion_and_read_back_as_two_steps(sar, oracle, gpu):
    """sar_colorize_format_async with no host image leaves the converted frame in device memory; sar_runtime_read_image_async
    fetches it (a sweep enqueues a batch's conversions at once and hands out host images as they come free), sar_runtime_image_done
    asks without waiting: the same bytes as the one-step call, for every format, also with the runtime rendered into again in between."""
    cfg = sar.Config.solar_sail(iterations=256 * 700, width=200, height=120, jobs_total=256, scale=1.0, transparent=1)
    st = sar.start_points(3, 0, 256)
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt, st)
    for fmt in (sar.SAR_FMT_RGBA16, sar.SAR_FMT_RGB16, sar.SAR_FMT_RGBA8, sar.SAR_FMT_RGB8):
        want = sar.colorize_format(cfg, rt, fmt)
        img = sar.HostImage(200, 120, fmt)
        sar.colorize_format_device(cfg, rt, fmt)
        ticket = sar.read_image_async(rt, img)
        sar.wait_image(rt, ticket)
        assert sar.image_done(rt, ticket)
        np.testing.assert_array_equal(img.array, want)
        img.close()
    # the converted image survives a reset + render of the runtime's other buffers until it is read
    want = sar.colorize_format(cfg, rt, sar.SAR_FMT_RGB16)
    sar.colorize_format_device(cfg, rt, sar.SAR_FMT_RGB16)
    rt.reset()
    sar.render_jobs(cfg, rt, sar.start_points(4, 0, 256))
    img = sar.HostImage(200, 120, sar.SAR_FMT_RGB16)
    ticket = sar.read_image_async(rt, img)
    sar.wait_image(rt, ticket)
    np.testing.assert_array_equal(img.array, want)
    img.close()
    with pytest.raises(sar.SarError):
        fresh = sar.Runtime(cfg)
        try:
            sar.read_image_async(fresh, sar.HostImage(200, 120, sar.SAR_FMT_RGB16))   # nothing colorized yet
        finally:
            fresh.close()
    rt.close()


@pytest.mark.gpu
def test_page_locked_images_from_announced_blocks(sar, oracle, gpu):
    """sar_host_reserve + sar_host_alloc: images that take a block the helper thread prepared, images beyond the announcement and
    images of another size are all ordinary page-locked images (read-backs land in them), in any order of alloc and free; what an
    announcement leaves is released by the next one."""
    cfg = sar.Config.solar_sail(iterations=256 * 700, width=1200, height=900, jobs_total=256, scale=1.0, transparent=0)
    rt = sar.Runtime(cfg)
    sar.render_jobs(cfg, rt, sar.start_points(3, 0, 256))
    want = sar.colorize_format(cfg, rt, sar.SAR_FMT_RGB16)
    nbytes = sar.image_bytes(sar.SAR_FMT_RGB16, 1200, 900)
    assert nbytes == 1200 * 900 * 6 and nbytes >= 4 << 20
    sar.host_reserve(nbytes, 3)
    imgs = [sar.HostImage(1200, 900, sar.SAR_FMT_RGB16) for _ in range(5)]       # three announced, two beyond
    other = sar.HostImage(1200, 900, sar.SAR_FMT_RGBA16)                          # another size: mapped on the spot
    sar.colorize_format_device(cfg, rt, sar.SAR_FMT_RGB16)
    for img in imgs:
        sar.wait_image(rt, sar.read_image_async(rt, img))
        np.testing.assert_array_equal(img.array, want)
    imgs[1].close()
    imgs[3].close()
    sar.host_reserve(nbytes, 2)
    again = sar.HostImage(1200, 900, sar.SAR_FMT_RGB16)                           # one of two taken, one left for the drop
    sar.wait_image(rt, sar.read_image_async(rt, again))
    np.testing.assert_array_equal(again.array, want)
    sar.host_reserve(0, 0)
    for img in (imgs[0], imgs[2], imgs[4], other, again):
        img.close()
    rt.close()
