"""Times the device side of the sliced multi-GPU exchange on ONE GPU (pack, merge of the owned slice, colorize of the
slice) for a given image size and GPU count — the inputs of the scaling prediction in DESIGN.md section 7 (the wire time
comes from the xGMI figures; 8-GPU runs are the driver's)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import strange_attractor_renderer_amd as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--worlds", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
cfg = S.Config.poisson_saturne(iterations=65536 * 200, width=a.size, height=a.size, jobs_total=65536, transparent=0, seed=1)
rt = S.Runtime(cfg)
S.render_jobs(cfg, rt, S.start_points(1, 0, 65536))
npix = a.size * a.size


def timed(fn):
    fn()
    rt.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        fn()
    rt.synchronize()
    return (time.perf_counter() - t0) / a.reps * 1e3


for world in a.worlds:
    sp = S.exchange_slice_pixels(npix, world)
    pack = torch.empty(world * sp * 16, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(world * sp * 16, dtype=torch.uint8, device="cuda")
    rgba = torch.empty(sp * 8, dtype=torch.uint8, device="cuda")
    sc = torch.empty(4, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    ex = S.Exchange(rt, world, 0)
    flags = torch.empty(ex.granules, dtype=torch.uint8, device="cuda")
    flags_all = torch.zeros(world * ex.granules, dtype=torch.uint8, device="cuda")
    ex.flags(flags.data_ptr())
    rt.synchronize()
    flags_all.view(world, ex.granules)[:] = flags        # every rank touched what this one did
    torch.cuda.synchronize()
    out = {"size": a.size, "world": world, "slice_pixels": sp,
           "pack_dense_ms": timed(lambda: ex.pack(None, 0.5, pack.data_ptr())),
           "merge_dense_ms": timed(lambda: ex.merge(recv.data_ptr(), sc.data_ptr())),
           "flags_ms": timed(lambda: ex.flags(flags.data_ptr())),
           "plan_and_pack_sparse_ms (one host wait)": timed(lambda: ex.pack(flags_all.data_ptr(), 2.0, pack.data_ptr())),
           "scalars_ms": timed(lambda: ex.finish(sc.data_ptr())),
           "colorize_slice_ms": timed(lambda: S.colorize_range_device(cfg, rt, 0, min(sp, npix), rgba.data_ptr())),
           "all_to_all_bytes_out_per_gpu": (world - 1) * sp * 16, "gather_bytes_to_root": (world - 1) * sp * 8}
    print(json.dumps(out), flush=True)
