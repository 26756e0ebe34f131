"""One GPU's eighth of BASELINE configs[3] (4096^2, 131 072 jobs x 9536 iterations = rank 5's slice of the job list) as
back-to-back frames, with and without the next frame announced: the per-GPU render time behind DESIGN.md section 7's
strong-scaling prediction. python tools/share_loop.py (needs a GPU)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import strange_attractor_renderer_amd as S
jobs, n, W = 131072, 9536, 4096
cfg = S.Config.poisson_saturne(iterations=jobs * n, width=W, height=W, jobs_total=jobs, transparent=0, seed=1)
st = torch.from_numpy(np.ascontiguousarray(S.start_points(3, 5 * jobs, jobs))).cuda()
rgba = torch.empty(W * W * 4, dtype=torch.int16, device="cuda")
for announce in (False, True, False, True):
    rt = S.Runtime(cfg)
    rt.enable_timing(True)
    def step(more):
        rt.reset()
        S.render_job_range_device(cfg, rt, jobs, n, st.data_ptr())
        if announce and more:
            S.prefetch_device(cfg, rt, jobs, n, st.data_ptr())
        S.colorize_device(cfg, rt, rgba.data_ptr())
    for k in range(3): step(k < 2)
    rt.synchronize(); rt.set_option("timing_accumulate", 1)
    t0 = time.perf_counter()
    K = 20
    for k in range(K): step(k + 1 < K)
    rt.synchronize()
    el = (time.perf_counter() - t0) / K * 1e3
    t = rt.last_timing()
    print("share announce=%s: %.3f ms/frame  warmup %.3f iterate %.3f fold %.3f  launch %s" % (announce, el, t.warmup_ms / K, t.iterate_ms / K, t.resolve_ms / K, rt.describe_last_launch()[:40]))
    rt.close()
