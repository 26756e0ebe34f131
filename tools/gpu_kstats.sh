#!/bin/bash
# rocprofv3 kernel-trace statistics of one config_table run: tools/gpu_kstats.sh <tag> "<config_table args>"
tag=$1; args=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python $GRAFT_REPO_ROOT/tools/config_table.py $args --out $OUT/table.jsonl > $OUT/run.out 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/trace/*/k_kernel_stats.csv $OUT/trace/k_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %5s  avg_us %10.1f  total_ms %9.2f  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
