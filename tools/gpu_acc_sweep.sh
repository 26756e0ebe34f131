#!/bin/bash
# Sweep of the accumulate grid: workgroups per bin (splits) x lists walked at once per lane group (acc_lists) x block size.
tag=${1:-acc}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log; tail -2 $out/pytest.log
for combo in "8 1 1024" "4 1 1024" "4 2 1024" "2 4 1024" "2 2 1024" "16 1 1024" "8 2 512" "8 4 256" "4 4 512" "16 1 512" "8 1 1024"; do
  set -- $combo
  echo "== splits=$1 lists=$2 threads=$3"
  timeout 120 python tools/config_table.py --only C2 C3 --reps 6 --option splits=$1 acc_lists=$2 acc_threads=$3 --out $out/t_$1_$2_$3.jsonl 2>&1 | grep -o '"config": "C[23]".*"fold_ms": [0-9.]*'
done
