#!/bin/bash
# builds x option sets on ONE box: tools/gpu_libs_opts.sh <tag> "<config_table args>" "<opts a>|<opts b>|..." <lib suffix>...
tag=$1; args=$2; IFS='|' read -ra OPTS <<< "$3"; shift; shift; shift
out=gpurun_out/$tag; mkdir -p $out
P=$PWD/strange_attractor_renderer_amd
for v in "$@"; do
  lib=$P/libsar_hip_$v.so; [ $v = base ] && lib=$P/libsar_hip.so
  for o in "${OPTS[@]}"; do
    echo "== $v [$o]"
    SAR_LIBRARY=$lib timeout 300 python tools/config_table.py $args ${o:+--option $o} --out $out/table_$v.jsonl 2>> $out/$v.err | grep -o '"config.*"fold_ms[^,]*\|"depth_atomics": [0-9]*' | paste - -
  done
done
