#!/bin/bash
set -u
OUT=gpurun_out/r2d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null > $GRAFT_REPO_ROOT/$OUT/counters_all.txt
cd $GRAFT_REPO_ROOT
grep -o -E "\b(TA_|TCP_|TD_|SQ_|TCC_|GRBM_|SPI_)[A-Za-z0-9_]+" $OUT/counters_all.txt | sort -u > $OUT/counter_names.txt
wc -l $OUT/counter_names.txt
PE="python tools/perf_explore.py --blocks 256 --out $OUT/perf.jsonl"
for st in 0 1; do
 for v in 0x3 0x13 0x23; do
  $PE --jobs 65536 131072 --records 28 --variants $v --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 65536 131072 196608 --records 20 --variants $v --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 65536 131072 196608 262144 --records 12 --variants $v --opt stager=$st > /dev/null 2>>$OUT/perf.err
 done
done
python - <<PY
import json
for l in open("$OUT/perf.jsonl"):
    d=json.loads(l)
    print("stager",d.get("stager"),"jobs",d["jobs"],"R",d["records"],d["variant"],"iter_ms %.3f fold_ms %.3f"%(d["iter_ms"],d["fold_ms"]))
PY
