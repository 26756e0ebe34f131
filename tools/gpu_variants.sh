#!/bin/bash
# Times variant builds of the library (strange_attractor_renderer_amd/libsar_hip_x_*.so, see build.py --variant).
# Usage: tools/gpu_variants.sh <tag> "<perf_explore args per run, ';'-separated>" [variant names...]
set -u
TAG=${1:-x}; RUNS=${2:-"--jobs 131072 --records 28;--jobs 196608 --records 20"}; shift; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
VARS=${@:-$(ls strange_attractor_renderer_amd/libsar_hip_x_*.so | sed 's/.*libsar_hip_x_//; s/\.so//')}
for v in $VARS; do
  export SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_x_$v.so
  IFS=';' read -ra RR <<< "$RUNS"
  for r in "${RR[@]}"; do
    timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --tag $v $r > /dev/null 2>>$OUT/perf.err || echo "FAILED $v $r" >> $OUT/perf.err
  done
done
python - <<PY
import json
for l in open("$OUT/perf.jsonl"):
    d=json.loads(l)
    print("%-12s stager %s pipe %s jobs %6d R %2d  iter_ms %.3f fold_ms %.3f wall %.3f"%(d["tag"],d.get("stager"),d.get("depth_pipe"),d["jobs"],d["records"],d["iter_ms"],d["fold_ms"],d["wall_ms"]))
PY
tail -5 $OUT/perf.err
