#!/bin/bash
# Round 4: narrow depth hints in 8 x 8 tiles (power-of-two widths) against row-major — parity, then A/B on one box.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4_tile_$1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/parity.log 2>&1; tail -2 $OUT/parity.log
T="python tools/config_table.py --reps 5 --out $OUT/table.jsonl"
for rep in 1 2 3; do
$T --only C4/8 --option hint_tile=1
$T --only C4/8
done 2>&1 | grep -v "amdgpu.ids"
$T --only X8192 2>&1 | grep -v "amdgpu.ids"; $T --only X8192 --option hint_tile=1 2>&1 | grep -v "amdgpu.ids"
$T --only XC4 --jobs 1048576 --reps 2 2>&1 | grep -v "amdgpu.ids"; $T --only XC4 --jobs 1048576 --reps 2 --option hint_tile=1 2>&1 | grep -v "amdgpu.ids"
