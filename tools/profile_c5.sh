#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4] (the batched `sequence` sweep): kernel + memory-copy statistics of both sweeps of
# `bench.py --config c5` and one PMC pass over the batched iterate kernel.   tools/profile_c5.sh <tag>  -> gpurun_out/profiles_c5_<tag>/
tag=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_c5_$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in readback hbm; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/trace_$mode -o k -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 720 --warmup 32 --c5-only $mode > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
done
pmc() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 96 --warmup 32 --c5-only hbm > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
}
pmc insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY
pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
print("# rocprofv3 summary — BASELINE configs[4], `bench.py --config c5` (batches of frames, sar_render_jobs_batch)\n")
for mode in ("readback", "hbm"):
    try:
        d = json.loads(open(os.path.join(root, f"bench_{mode}.json")).read().strip().splitlines()[-1])
        ms = d["ms_per_frame_per_gpu"] if mode == "readback" else d["rgba16_in_hbm"]["ms_per_frame_per_gpu"]
        print(f"## sweep `{mode}` under the tracer: {ms:.3f} ms per frame, frames per launch {d['config']['frames_per_launch'] or d['config']['frames_per_launch_in_hbm']}, {d['config']['launch']}\n")
    except Exception as e:
        print(f"## sweep `{mode}`: no bench line ({e})\n")
    for f in glob.glob(os.path.join(root, f"trace_{mode}", "**", "*kernel_stats.csv"), recursive=True):
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in list(csv.DictReader(open(f)))[:14]:
            print(f"| `{r['Name'][:64]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
        print()
    for f in glob.glob(os.path.join(root, f"trace_{mode}", "**", "*memory_copy_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            print(f"* {r['Name']}: {r['Calls']} copies, avg {float(r['AverageNs'])/1e3:.1f} us, total {float(r['TotalDurationNs'])/1e6:.1f} ms")
        print()
print("## PMC passes over the `hbm` sweep (each group in its own run; mean per dispatch)\n")
print("| kernel | counter | dispatches | mean per dispatch |\n|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r.get("Kernel_Name", "?")][r.get("Counter_Name")].append(float(r.get("Counter_Value") or 0))
    for k, cs in agg.items():
        if not any(t in k for t in ("k_iterate_split_batch", "k_warmup_batch", "k_bin_accumulate_batch", "k_fold_resolve_batch")):
            continue
        for c, v in cs.items():
            print(f"| `{k[:44]}` | {c} | {len(v)} | {sum(v)/len(v):.6g} |")
PY
