#!/bin/bash
# One GPU-box visit of a round: the -m gpu suite, the default bench line, the multi-rank paths on one GPU (gloo), the C4
# frame. Usage (through gpurun, from the repo root): tools/gpu_round.sh <tag>
set -u
TAG=${1:-x}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocminfo | grep -E "Marketing|Compute Unit" | head -4 > $OUT/box.txt 2>&1
nproc >> $OUT/box.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 600 $OUT/bench_n1.json
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --check > $OUT/bench_n2_gloo.json 2> $OUT/bench_n2_gloo.err; tail -c 900 $OUT/bench_n2_gloo.json; tail -3 $OUT/bench_n2_gloo.err
timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --exchange rooted --check > $OUT/bench_n2_gloo_rooted.json 2> $OUT/bench_n2_gloo_rooted.err; tail -3 $OUT/bench_n2_gloo_rooted.err
timeout 600 python bench.py --config c4 --steps 3 --warmup 1 > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; tail -c 900 $OUT/bench_c4_n1.json
timeout 600 python bench.py --config c4 --gpus 2 --steps 2 --warmup 1 --check > $OUT/bench_c4_n2_gloo.json 2> $OUT/bench_c4_n2_gloo.err; tail -c 600 $OUT/bench_c4_n2_gloo.json; tail -3 $OUT/bench_c4_n2_gloo.err
