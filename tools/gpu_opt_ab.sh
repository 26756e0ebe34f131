#!/bin/bash
# Parity suite, then A/B of one runtime option on the config table. Usage: tools/gpu_opt_ab.sh <tag> "<opt=a>" "<opt=b>" [configs...]
tag=$1; A=$2; B=$3; shift 3; cfgs=${@:-C2 C3}
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
for rep in 1 2 3; do for o in "$A" "$B"; do
  echo "== $o"; timeout 200 python tools/config_table.py --only $cfgs --reps 5 --option $o --out $out/t.jsonl 2>&1 | grep -o '"config": "[^"]*".*"fold_ms": [0-9.]*' | sed 's/"jobs.*wall_ms/ wall_ms/'
done; done
