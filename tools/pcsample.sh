#!/bin/bash
# PC sampling of the C2 iterate kernel (stochastic = hardware sampling with stall reasons on gfx950).
# usage (on the GPU box): tools/pcsample.sh <tag> [perf_explore args...]
tag=${1:-pcs}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for method in stochastic host_trap; do
  if [ $method = stochastic ]; then unit=cycles; iv=65536; else unit=time; iv=100; fi
  timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit \
     --pc-sampling-interval $iv --kernel-trace --output-format csv json -d $out/$method -- \
     python $GRAFT_REPO_ROOT/tools/perf_explore.py --jobs 131072 --blocks 256 --out $out/$method.jsonl "$@" > $out/$method.log 2>&1
  echo "$method rc=$?"; tail -3 $out/$method.log
  find $out/$method -type f | head; du -sh $out/$method
done
