#!/bin/bash
# A/B of two builds of the same ABI on one GPU-box visit: parity of the candidate (both stagers), then interleaved timing.
#   tools/gpu_ab_lib.sh <tag> <baseline .so> [candidate .so]
tag=${1:-ab}; base=$2; cand=${3:-}
out=gpurun_out/$tag; mkdir -p $out
export SAR_STAGER=2
[ -n "$cand" ] && export SAR_LIBRARY=$cand
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
unset SAR_STAGER SAR_LIBRARY
for rep in 1 2 3; do
  for which in base cand; do
    if [ $which = base ]; then lib=$base; else lib=$cand; fi
    SAR_LIBRARY=$lib timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-pipeline > $out/bench_${which}_$rep.json 2> $out/bench_${which}_$rep.err
    SAR_LIBRARY=$lib timeout 120 python tools/config_table.py --only C2 C3 > $out/table_${which}_$rep.txt 2>&1
  done
done
tail -3 $out/pytest.log
for f in $out/bench_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['achieved'])"; done
