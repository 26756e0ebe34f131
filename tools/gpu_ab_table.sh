#!/bin/bash
# A/B of two builds on ONE box over the config table: parity of the candidate first, then three interleaved timing rounds.
#   tools/gpu_ab_table.sh <tag> <candidate .so> "<configs>" [config_table options...]
tag=$1; cand=$GRAFT_REPO_ROOT/$2; cfgs=$3; shift 3
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
SAR_LIBRARY=$cand timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $out/pytest.log 2>&1; tail -1 $out/pytest.log
for rep in 1 2 3; do
  for which in base cand; do
    lib=$GRAFT_REPO_ROOT/strange_attractor_renderer_amd/libsar_hip.so; [ $which = cand ] && lib=$cand
    SAR_LIBRARY=$lib python tools/config_table.py --only $cfgs --reps 5 --out $out/table_$which.jsonl "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$which', d['config'], 'wall', d['wall_ms'], 'iterate', d['iterate_ms'], 'fold', d['fold_ms'], 'cand', d.get('depth_candidates'), 'atomics', d['depth_atomics'])"
  done
done
