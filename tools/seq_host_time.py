import sys, time, collections
sys.path.insert(0, ".")
import strange_attractor_renderer_amd as S
from strange_attractor_renderer_amd import api, sequence
acc = collections.Counter()
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(mod, name, g)
for n in ("render_jobs", "colorize_format_async", "wait_image", "start_points"):
    wrap(api, n)
orig_reset = api.Runtime.reset
def reset(self):
    t = time.perf_counter(); orig_reset(self); acc["reset"] += time.perf_counter() - t
api.Runtime.reset = reset
orig_replace = api.Config.replace
def replace(self, **k):
    t = time.perf_counter(); r = orig_replace(self, **k); acc["replace"] += time.perf_counter() - t; return r
api.Config.replace = replace
scfg = S.Config.solar_sail(iterations=100_000_000, width=1800, height=2000, scale=1.0, transparent=0)
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for frames in (5, 90):
    acc.clear(); t0 = time.perf_counter()
    sequence.render_sequence(scfg, 0.0, float(frames), 1.0, units=16384, jobs_per_thread=4, seed=4, sink=lambda *a: None, image_format=S.SAR_FMT_RGB16, lanes=lanes)
    el = time.perf_counter() - t0
    print(frames, "frames", "%.3f ms/frame" % (el / frames * 1e3), {k: round(v / frames * 1e3, 3) for k, v in acc.items()})
