#!/bin/bash
set -u
OUT=gpurun_out/r2g
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
SAR_STAGER=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_stager1.log 2>&1; echo "rc=$?" >> $OUT/pytest_stager1.log; tail -3 $OUT/pytest_stager1.log
SAR_STAGER=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest_stager2.log 2>&1; echo "rc=$?" >> $OUT/pytest_stager2.log; tail -3 $OUT/pytest_stager2.log
SAR_STAGER=2 SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_spare2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_spare2.log 2>&1; echo "rc=$?" >> $OUT/pytest_spare2.log; tail -3 $OUT/pytest_spare2.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err; python -c "
import json;d=json.loads(open('$OUT/bench_n1.json').read().strip().splitlines()[-1]);print('bench',d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['kernel_ms_per_step'])"
timeout 300 python tools/exchange_kernels.py --size 4096 > $OUT/exchange_kernels.jsonl 2>$OUT/exchange_kernels.err; timeout 300 python tools/exchange_kernels.py --size 2048 >> $OUT/exchange_kernels.jsonl 2>>$OUT/exchange_kernels.err; cat $OUT/exchange_kernels.jsonl
for st in 1 2; do for j in "131072 28" "196608 20"; do set -- $j; timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs $1 --records $2 --opt stager=$st > /dev/null 2>>$OUT/perf.err; done; done
timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs 131072 196608 > /dev/null 2>>$OUT/perf.err
timeout 120 python tools/perf_explore.py --blocks 256 --variants 0x3 --out $OUT/perf.jsonl --jobs 65536 131072 --size 4096 --n 1.25e9 > /dev/null 2>>$OUT/perf.err
python - <<PY
import json
for l in open("$OUT/perf.jsonl"):
    d=json.loads(l)
    print("stager",d.get("stager"),"jobs",d["jobs"],"size",d["size"],"R",d["records"],"iter_ms %.3f fold_ms %.3f wall %.3f"%(d["iter_ms"],d["fold_ms"],d["wall_ms"]))
PY
