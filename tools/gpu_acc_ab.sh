#!/bin/bash
# A/B of k_bin_accumulate variants on ONE box: tools/gpu_acc_ab.sh <tag> "<configs>" <variant>...   ("base" = the product's hooks build)
# three interleaved rounds of the config table (accumulate + fold = fold_ms), then (SAR_AB_PARITY=1) the parity suite on every variant
tag=$1; cfgs=$2; shift 2
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$GRAFT_REPO_ROOT/build/variants/libsar_hip_${v}_hooks.so; [ $v = base ] && lib=$GRAFT_REPO_ROOT/tests/hooks/libsar_hip_hooks.so
    SAR_LIBRARY=$lib timeout 300 python tools/config_table.py --only $cfgs --reps 6 --out $out/table_$v.jsonl 2>$out/$v.err | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('%-8s %-9s wall %7.3f iterate %7.3f acc+fold %6.3f' % ('$v', d['config'], d['wall_ms'], d['iterate_ms'], d['fold_ms']))"
  done
done
[ -n "$SAR_AB_PARITY" ] || exit 0
for v in "$@"; do
  [ $v = base ] && continue
  SAR_LIBRARY=$GRAFT_REPO_ROOT/build/variants/libsar_hip_${v}_hooks.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest_$v.log 2>&1; echo "$v: $(tail -1 $out/pytest_$v.log)"
done
