#!/bin/bash
# Times several builds of the library on ONE box: tools/gpu_libs_table.sh <tag> "<config_table args>" <lib suffix>...
# ("base" = the product build; other names = build/variants/libsar_hip_<name>.so, see build.py --variant)
tag=$1; args=$2; shift; shift
out=gpurun_out/$tag; mkdir -p $out
P=$PWD/strange_attractor_renderer_amd
for v in "$@"; do
  lib=$PWD/build/variants/libsar_hip_$v.so; [ $v = base ] && lib=$P/libsar_hip.so
  echo "== $v"
  SAR_LIBRARY=$lib timeout 300 python tools/config_table.py $args --out $out/table_$v.jsonl 2> $out/$v.err | grep -o '"config.*"colorize' 
  grep prof $out/$v.err | sort | uniq -c | sort -rn | head -4
done
