#!/bin/bash
# parity suite of one variant build, then the timing table of several: tools/gpu_r3_ab.sh <tag> <parity variant|none> <variants...>
tag=$1; pv=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
P=$PWD/strange_attractor_renderer_amd
if [ $pv != none ]; then
  lib=$P/libsar_hip_$pv.so; [ $pv = base ] && lib=$P/libsar_hip.so
  SAR_LIBRARY=$lib timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
fi
bash tools/gpu_libs_table.sh $tag "--only C2 C3 C4/8 --reps 3" "$@"
