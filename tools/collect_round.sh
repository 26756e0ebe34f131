#!/bin/bash
# Copies what a closing visit (tools/gpu_round_final.sh <tag>) left under gpurun_out/ into profiles/<tag>_* (the committed evidence).
#   bash tools/collect_round.sh r06
tag=${1:-rXX}; o=gpurun_out/$tag
for f in bench_n1 bench_c4_n1 bench_c5_n1 bench_n2_gloo_one_gpu bench_n8_gloo_one_gpu bench_native_1dev_c2 bench_native_8shards_c2 bench_native_8shards_c4; do
  [ -s $o/$f.json ] && cp $o/$f.json profiles/${tag}_$f.json
done
cp $o/config_table.jsonl profiles/${tag}_config_table.jsonl
cp $o/pytest_gpu.log profiles/${tag}_pytest_gpu.log
for v in split1 split2 spare2; do echo "$v: $(tail -1 $o/pytest_$v.log)" ; done >> profiles/${tag}_pytest_gpu.log
cp gpurun_out/profiles_$tag/SUMMARY.md profiles/${tag}_rocprofv3_summary.md
cp gpurun_out/profiles_$tag/pmc.json profiles/${tag}_pmc.json
cp $(ls gpurun_out/profiles_$tag/trace/*/bench_kernel_stats.csv gpurun_out/profiles_$tag/trace/bench_kernel_stats.csv 2>/dev/null | head -1) profiles/${tag}_kernel_stats.csv
cp gpurun_out/pmc_c4_$tag/SUMMARY.md profiles/${tag}_pmc_c4_summary.md
cp gpurun_out/pmc_c4_$tag/pmc_c4.json profiles/${tag}_pmc_c4.json
sed -n '/^# rocprofv3 summary/,$p' $o/profile_c5.log > profiles/${tag}_c5_rocprofv3_summary.md
for m in readback hbm; do cp $(ls gpurun_out/profiles_c5_$tag/trace_$m/*/k_kernel_stats.csv gpurun_out/profiles_c5_$tag/trace_$m/k_kernel_stats.csv 2>/dev/null | head -1) profiles/${tag}_c5_kernel_stats_$m.csv; done
for f in 360 45; do for m in readback hbm; do python - $o/cold_${f}_$m.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "cold ms per repetition (the first is process-cold; 1 s pause before each of the others):", [round(r["cold_ms"], 1) for r in d["reps"]], "batches", d["reps"][-1]["frames_per_launch"][:4])
PY
done; done > profiles/${tag}_cold_sweep_tool.txt
cat profiles/${tag}_cold_sweep_tool.txt
