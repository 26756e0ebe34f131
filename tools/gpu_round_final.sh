#!/bin/bash
# Final visit of a round: whole -m gpu suite, the pool stager with a ring of 2 through the parity suite, smoke, the default
# bench line (with the CPU baseline) and the rocprofv3 evidence. Usage: tools/gpu_round_final.sh <tag>
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -4 $OUT/pytest_gpu.log
SAR_STAGER=2 SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_spare2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_spare2.log 2>&1; echo "rc=$?" >> $OUT/pytest_spare2.log; tail -3 $OUT/pytest_spare2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 1500 $OUT/bench_n1.json
timeout 300 python bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; tail -c 600 $OUT/bench_c4_n1.json
timeout 2400 tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
