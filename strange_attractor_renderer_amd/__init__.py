"""MI355X-native iterate/accumulate path of Icelk/strange-attractor-renderer.

`libsar_hip.so` (HIP, gfx950) is the product; this package is the thin host-side mirror of the
reference crate's Config / Runtime / render / colorize / ParallelRenderer / render_parallel surface
over the C ABI declared in include/sar.h. No CPU fallback exists by design.
"""
from .api import (Config, Exchange, ParallelRenderer, Runtime, SarError, Timing, attractor_extent, colorize,  # noqa: F401
                  colorize_device, colorize_device_batch, reset_batch, colorize_range_device, exchange_slice_pixels, bin_geometry,
                  colorize_format, colorize_format_async, colorize_format_device, read_image_async, image_done, wait_image, HostImage, host_reserve, image_bytes, convert_device, device_count, image_format, render, render_job_range, render_job_range_device, prefetch_device, render_jobs, render_jobs_batch, batch_frames, render_parallel_into,
                  render_parallel, start_points, write_image, write_image_matches)
from ._abi import (SAR_CT_ADJUSTED_VELOCITY, SAR_CT_POISSON_SATURNE, SAR_FMT_RGB8, SAR_FMT_RGB16,  # noqa: F401
                   SAR_FMT_RGBA8, SAR_FMT_RGBA16, SAR_RENDER_DEPTH, SAR_RENDER_GAS, load_library, use_hooks_build)

RenderKind = type("RenderKind", (), {"Gas": SAR_RENDER_GAS, "Depth": SAR_RENDER_DEPTH})
