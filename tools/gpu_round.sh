#!/bin/bash
# One closing-style GPU-box visit: the whole -m gpu suite, the parity suite with the whole / the split iterate kernel forced
# and through a build with a ring of two spare buffers, smoke(), the default bench line, the config table.
#   tools/gpu_round.sh <tag> [fast]      (fast: skip the variant suites)
tag=${1:-r}; fast=${2:-}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "rc=$?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
if [ -z "$fast" ]; then
  for v in 1 2; do SAR_SPLIT=$v timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $out/pytest_split$v.log 2>&1; echo "SAR_SPLIT=$v: $(tail -1 $out/pytest_split$v.log)"; done
  [ -f build/variants/libsar_hip_spare2.so ] || SAR_EXTRA_FLAGS=-DSAR_POOL_SPARE=2u python -m strange_attractor_renderer_amd.build --variant spare2 > $out/build_spare2.log 2>&1
  SAR_LIBRARY=$GRAFT_REPO_ROOT/build/variants/libsar_hip_spare2_hooks.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q > $out/pytest_spare2.log 2>&1; echo "spare2: $(tail -1 $out/pytest_spare2.log)"
fi
python __graft_entry__.py --smoke 2>&1 | tail -1
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; python - <<PY
import json
d=json.loads(open("$out/bench_n1.json").read().strip().splitlines()[-1])
print("bench: value %.4g  ms/step %.3f  kernel_ms %.3f  frac %.3f  sustained median %.3f  cpu %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("sustained",{}).get("ms_per_frame",{}).get("median",0), d.get("cpu_baseline",{}).get("value")))
PY
python tools/config_table.py --reps 4 --out $out/config_table.jsonl 2>/dev/null | cut -c1-330
