"""Condenses a tools/profile_round.sh output directory into one markdown summary (kernel stats + per-kernel
PMC averages). The summary is what gets committed under profiles/."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print(f"# rocprofv3 summary — {os.path.basename(root)}\n")
print("Commands profiled (N=1, BASELINE configs[1]): `python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline --sustained-seconds 0` "
      "under `--kernel-trace --stats`; the same with `--steps 3 --warmup 1` under each `--pmc` pass.\n")
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("## Kernel time (`--kernel-trace --stats`, no counters)\n")
    print("| kernel | calls | total ms | avg ms | % |")
    print("|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {r['Percentage']} |")
    print()
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "k_iterate" in r["Kernel_Name"]]
    if d:
        timed = d[3:23]
        print("the iterate kernel per dispatch (ms): " + ", ".join(f"{x:.3f}" for x in d) +
              f" — the first three dispatches are bench.py's untimed warm-up steps (cold caches, first touch of the arena), the 24th "
              f"is the untimed frame whose checksums the line's `parity` reports; mean of the 20 timed ones "
              f"{sum(timed) / max(len(timed), 1):.3f} ms, which is what bench.py's HIP events average.\n")
        med = sorted(timed)[len(timed) // 2] if timed else 0.0
        slow = [k + 4 for k, x in enumerate(timed) if x > 1.08 * med]
        print(f"timed dispatches slower than 1.08 x their median ({med:.3f} ms): {slow or 'none'} (round 5's trace held six, in pairs eight "
              f"dispatches apart; profiles/r06_questions.md: the tracer and HIP events agree to 6 us on the same dispatches, the pairs did not "
              f"come back on three other boxes).\n")
try:
    line = [l for l in open(os.path.join(root, "trace_bench.json")) if l.startswith("{")][-1]
    b = json.loads(line)
    print(f"bench line under the tracer: value {b['value']:.4g} {b['unit']}, ms_per_step {b['ms_per_step']:.3f}, "
          f"kernel_ms_per_step {b.get('kernel_ms_per_step')}, roofline.kernel_ms {b['roofline'].get('kernel_ms')}\n")
except Exception as e:  # noqa
    print(f"(no bench line: {e})\n")

pmc_json = {}
print("## PMC passes (each counter group in its own run; averages per dispatch of each kernel)\n")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"### {os.path.basename(d)}: no counter file (see {os.path.basename(d)}.err)\n")
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name") or r.get("Kernel Name") or "?"
            agg[k][r.get("Counter_Name") or r.get("Counter Name")].append(float(r.get("Counter_Value") or r.get("Counter Value") or 0))
    print(f"### {os.path.basename(d)}\n")
    print("| kernel | counter | dispatches | mean per dispatch |")
    print("|---|---|---|---|")
    for k, cs in agg.items():
        if not any(t in k for t in ("k_iterate", "k_bin_accumulate", "k_fold", "k_colorize")):
            continue
        short = next(t for t in ("k_iterate_split", "k_iterate_lean", "k_iterate_binned", "k_iterate", "k_bin_accumulate", "k_fold_resolve", "k_colorize_gas", "k_colorize") if t in k)
        for c, v in cs.items():
            print(f"| `{k[:48]}` | {c} | {len(v)} | {sum(v)/len(v):.6g} |")
            pmc_json.setdefault(short, {})[c] = sum(v) / len(v)
    print()

if len(sys.argv) > 2:
    json.dump(pmc_json, open(sys.argv[2], "w"), indent=1, sort_keys=True)
