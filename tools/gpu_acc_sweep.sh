#!/bin/bash
# Sweep of the accumulate grid: workgroups per bin (splits) x block size, on the config table.
tag=${1:-acc}; out=gpurun_out/$tag; mkdir -p $out
for combo in "8 1024" "4 1024" "16 1024" "8 512" "16 512" "32 512" "16 256" "8 1024"; do
  set -- $combo
  echo "== splits=$1 threads=$2"
  timeout 120 python tools/config_table.py --only C2 C3 --reps 6 --option splits=$1 acc_threads=$2 --out $out/t_$1_$2.jsonl 2>&1 | grep -o '"config": "C[23]".*"fold_ms": [0-9.]*' | sed 's/"jobs.*wall_ms/ wall_ms/'
done
