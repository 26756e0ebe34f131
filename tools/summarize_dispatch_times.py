"""Condenses tools/r06_questions.sh (i): per run, the iterate kernel's dispatch-by-dispatch durations by HIP events (the library's
spans) and — for the traced runs — by the tracer's own timestamps, side by side."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def stats(v):
    s = sorted(v)
    return f"min {s[0]:.3f} / median {s[len(s) // 2]:.3f} / max {s[-1]:.3f} / mean {sum(s) / len(s):.3f}"


def slow(v):
    med = sorted(v)[len(v) // 2]
    return [k for k, x in enumerate(v) if x > 1.08 * med]


print("# Dispatch-by-dispatch duration of `k_iterate_split<60,u32,PH=2>` over 24 announced frames of BASELINE configs[1]\n")
for name in sorted(glob.glob(os.path.join(root, "*events*.json"))):
    d = json.load(open(name))
    tag = os.path.basename(name)[:-5]
    it = d["iterate_ms"]
    print(f"## {tag} (prefetch {d['prefetch']})\n")
    print(f"* HIP events: {stats(it)}; dispatches > 1.08 x median: {slow(it)}")
    print(f"* frame (event to event on the launch stream): {stats(d['frame_ms'])}")
    print(f"* per dispatch: {' '.join(f'{x:.2f}' for x in it)}")
    tdir = os.path.join(root, tag.replace("traced_events_", "traced_"))
    tr = []
    warm = []
    if tag.startswith("traced_events_") and os.path.isdir(tdir):
        rows = []
        for f in glob.glob(os.path.join(tdir, "**", "*kernel_trace.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            if "k_iterate_split" in r["Kernel_Name"]:
                tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
            elif "k_warmup" in r["Kernel_Name"]:
                warm.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        tr = tr[-len(it):]
        dur = [(e - s) / 1e6 for s, e in tr]
        print(f"* tracer, the same dispatches: {stats(dur)}; > 1.08 x median: {slow(dur)}")
        print(f"* per dispatch (tracer): {' '.join(f'{x:.2f}' for x in dur)}")
        # how much of every iterate dispatch a k_warmup of the side stream ran under
        ov = []
        for s, e in tr:
            o = sum(max(0, min(e, we) - max(s, ws)) for ws, we in warm)
            ov.append(o / 1e6)
        print(f"* k_warmup overlapping each dispatch (ms): {' '.join(f'{x:.2f}' for x in ov)}")
        diff = [a - b for a, b in zip(it, dur)]
        print(f"* HIP events minus tracer: {stats(diff)}")
    print()
