"""The C ABI from a compiled C99 program (tests/c/sar_driver.c) — no Python, no ctypes in the process: the call
sequence of the Rust safe layer (bindings/rust-safe: GpuRuntime::new / render / colorize / reset, GpuRenderer::new /
new_multi / render_parallel / shutdown) exactly as a Rust, cgo or JNI host would issue it."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "strange_attractor_renderer_amd")


def _build(tmp_path):
    exe = str(tmp_path / "sar_driver")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "sar_driver.c"), "-o", exe, "-L", PKG, "-l:libsar_hip.so",
                    f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_c_driver_builds_and_reports_a_missing_device_as_a_status(sar, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, str(tmp_path), "64", "48", "8", "100", "5", "1"], capture_output=True, text=True)
    if sar.device_count() > 0:
        assert out.returncode == 0, out.stderr
    else:
        assert out.returncode == 3 and "no HIP device" in out.stderr, (out.returncode, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [1, 3])
def test_c_driver_results_equal_the_oracle(sar, oracle, gpu, tmp_path, shards):
    W, H, jobs, n, seed = 300, 222, 192, 2500, 31
    exe = _build(tmp_path)
    out = subprocess.run([exe, str(tmp_path), str(W), str(H), str(jobs), str(n), str(seed), str(shards)],
                         capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", (out.returncode, out.stdout, out.stderr)
    cfg = oracle.poisson_saturne()
    cfg.width, cfg.height, cfg.transparent = W, H, 0
    ort = oracle.Runtime(W, H)
    oracle.render_jobs(cfg, ort, oracle.start_points(seed, 0, jobs), n)
    rd = lambda name, dt: np.fromfile(str(tmp_path / name), dtype=dt)  # noqa: E731
    assert np.array_equal(rd("count.bin", np.uint32).reshape(H, W), ort.count)
    assert int(rd("max.bin", np.uint32)[0]) == ort.max
    assert np.array_equal(rd("zbuf.bin", np.uint32).reshape(H, W), ort.zbuf.view(np.uint32))
    assert np.array_equal(rd("steps.bin", np.uint64).reshape(H, W), ort.steps.view(np.uint64))
    want = oracle.colorize(cfg, ort)
    assert np.array_equal(rd("rgba.bin", np.uint16).reshape(H, W, 4), want)
    # render_parallel with units = jobs, 1 job per unit: the same jobs (the renderer's stream starts at the same seed),
    # on one device or sharded over three — contiguous shards folded in order == the sequential result
    assert np.array_equal(rd("count_parallel.bin", np.uint32).reshape(H, W), ort.count)
    assert np.array_equal(rd("rgba_parallel.bin", np.uint16).reshape(H, W, 4), want)
    # sar_render_jobs_batch from C: three frames (own runtime, angle 0.3 i, stream seeded seed + 1 + i) in one set of launches
    sums = rd("count_batch_fnv.bin", np.uint64)
    for i in range(3):
        c = oracle.copy_config(cfg)
        c.angle = 0.3 * i
        o = oracle.Runtime(W, H)
        oracle.render_jobs(c, o, oracle.start_points(seed + 1 + i, 0, jobs), n)
        assert np.array_equal(rd(f"count_batch_{i}.bin", np.uint32).reshape(H, W), o.count), f"batched frame {i}"
        assert np.array_equal(rd(f"rgba_batch_{i}.bin", np.uint16).reshape(H, W, 4), oracle.colorize(c, o)), f"batched frame {i}"
        assert int(sums[i]) == oracle.fnv1a64(o.count)                      # sar_checksum_fnv1a64
        # ... and read back in two steps (conversion in device memory, then sar_runtime_read_image_async) as RGB8
        assert np.array_equal(rd(f"rgb8_batch_{i}.bin", np.uint8).reshape(H, W, 3), oracle.convert(3, oracle.colorize(c, o))), f"RGB8 frame {i}"
    if shards > 1:   # the second frame of the renderer went through the dense exchange: the stream ran on, other jobs
        o2 = oracle.Runtime(W, H)
        oracle.render_jobs(cfg, o2, oracle.start_points(seed, jobs, jobs), n)
        assert np.array_equal(rd("rgba_parallel_dense.bin", np.uint16).reshape(H, W, 4), oracle.colorize(cfg, o2))
