#!/bin/bash
set -u
OUT=gpurun_out/r2h
mkdir -p $OUT
SAR_STAGER=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_stager2.log 2>&1; echo "rc=$?" >> $OUT/pytest_stager2.log; tail -3 $OUT/pytest_stager2.log
SAR_STAGER=2 SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_spare2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_spare2.log 2>&1; echo "rc=$?" >> $OUT/pytest_spare2.log; tail -3 $OUT/pytest_spare2.log
for st in 1 2; do for j in "131072 28 2048 1e9" "196608 20 2048 1e9" "131072 12 4096 1.25e9" "196608 12 4096 1.25e9" "131072 20 4096 1.25e9"; do set -- $j; timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs $1 --records $2 --size $3 --n $4 --opt stager=$st > /dev/null 2>>$OUT/perf.err; done; done
timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs 131072 196608 > /dev/null 2>>$OUT/perf.err
timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs 65536 131072 196608 --size 4096 --n 1.25e9 > /dev/null 2>>$OUT/perf.err
timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs 131072 196608 --preset solar_sail --size 2000 > /dev/null 2>>$OUT/perf.err
python - <<PY
import json
for l in open("$OUT/perf.jsonl"):
    d=json.loads(l)
    print("stager",d.get("stager"),d["preset"][:6],"jobs",d["jobs"],"size",d["size"],"R",d["records"],"iter_ms %.3f fold_ms %.3f wall %.3f"%(d["iter_ms"],d["fold_ms"],d["wall_ms"]))
PY
tail -3 $OUT/perf.err
