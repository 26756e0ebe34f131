#!/bin/bash
# A/B of the two stagers of the iterate kernel: parity of the pool stager over the whole parity suite (also with a ring of
# 2 spare buffers, which forces the many-fillers rounds), then kernel times. Usage: tools/gpu_stager_ab.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
SAR_STAGER=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_sequence.py -m gpu -x -q > $OUT/pytest_pool.log 2>&1; echo "rc=$?" >> $OUT/pytest_pool.log
tail -4 $OUT/pytest_pool.log
if [ -f strange_attractor_renderer_amd/libsar_hip_spare2.so ]; then
  SAR_STAGER=2 SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_spare2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_pool_spare2.log 2>&1; echo "rc=$?" >> $OUT/pytest_pool_spare2.log
  tail -3 $OUT/pytest_pool_spare2.log
fi
PE="python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl"
for st in 1 2; do
  $PE --jobs 131072 --records 28 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 196608 --records 20 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 131072 --records 20 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 131072 --records 28 --variants 0x13 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 131072 --size 4096 --n 1.25e9 --records 12 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 65536 --size 4096 --n 1.25e9 --records 28 --opt stager=$st > /dev/null 2>>$OUT/perf.err
  $PE --jobs 131072 --preset solar_sail --size 2000 --records 28 --opt stager=$st > /dev/null 2>>$OUT/perf.err
done
python - <<PY
import json
for l in open("$OUT/perf.jsonl"):
    d=json.loads(l)
    print("stager",d.get("stager"),"jobs",d["jobs"],"size",d["size"],"R",d["records"],d["variant"],d["preset"],"iter_ms %.3f fold_ms %.3f wall %.3f"%(d["iter_ms"],d["fold_ms"],d["wall_ms"]))
PY
