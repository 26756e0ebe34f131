#!/bin/bash
# Closing visit of round 3: whole -m gpu suite, smoke(), the bench lines (N=1 default, c4, c5, the one-command SCALE form with
# 2 ranks on the one GPU over gloo, the native renderer with 2 and 8 shards), kernel statistics of the c2 and c4 benches.
set -u
TAG=${1:-r03_final}
OUT=gpurun_out/$TAG; mkdir -p $OUT
rocminfo | grep -E "Marketing|Compute Unit" | head -2 > $OUT/box.txt 2>&1; nproc >> $OUT/box.txt
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
SAR_STAGER=2 SAR_LIBRARY=$PWD/strange_attractor_renderer_amd/libsar_hip_spare2.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_spare2.log 2>&1; echo "rc=$?" >> $OUT/pytest_spare2.log; tail -2 $OUT/pytest_spare2.log
SAR_STAGER=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_stager1.log 2>&1; echo "rc=$?" >> $OUT/pytest_stager1.log; tail -2 $OUT/pytest_stager1.log
SAR_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_whole_kernel.log 2>&1; echo "rc=$?" >> $OUT/pytest_whole_kernel.log; tail -2 $OUT/pytest_whole_kernel.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err; tail -c 300 $OUT/bench_n1.json; echo
timeout 900 python bench.py --config c4 --steps 5 --warmup 1 > $OUT/bench_c4_n1.json 2> $OUT/bench_c4_n1.err; tail -c 300 $OUT/bench_c4_n1.json; echo
timeout 900 python bench.py --config c4 --jobs 524288 --steps 5 --warmup 1 > $OUT/bench_c4_n1_524288_jobs.json 2> $OUT/bench_c4_n1_524288.err
timeout 900 python bench.py --config c5 --steps 360 --warmup 4 > $OUT/bench_c5_n1.json 2> $OUT/bench_c5_n1.err; tail -c 300 $OUT/bench_c5_n1.json; echo
timeout 1200 python bench.py --gpus 2 --steps 5 --warmup 2 --check > $OUT/bench_n2_gloo_one_gpu.json 2> $OUT/bench_n2_gloo.err; tail -c 400 $OUT/bench_n2_gloo_one_gpu.json; echo
timeout 900 python bench.py --native --gpus 2 --config c2 --steps 5 --warmup 1 > $OUT/bench_native_2shards_c2.json 2> $OUT/native.err
timeout 900 python bench.py --native --gpus 8 --config c2 --steps 3 --warmup 1 > $OUT/bench_native_8shards_c2.json 2>> $OUT/native.err
timeout 900 python bench.py --native --gpus 8 --config c4 --steps 3 --warmup 1 > $OUT/bench_native_8shards_c4.json 2>> $OUT/native.err; tail -c 500 $OUT/bench_native_8shards_c4.json; echo
bash tools/gpu_kstats.sh $TAG/ks_c4 "--only XC4 --jobs 1048576 --reps 2" > $OUT/c4_kernel_stats.txt 2>&1
bash tools/gpu_kstats.sh $TAG/ks_c2 "--only C2 C3 C4/8 --reps 3" > $OUT/c2_c3_c4share_kernel_stats.txt 2>&1
timeout 600 python tools/config_table.py --reps 4 --out $OUT/config_table.jsonl > $OUT/config_table.out 2>&1
timeout 600 python tools/config_table.py --reps 3 --only XHD X2560 X3072 X4K X8192 --out $OUT/config_table.jsonl >> $OUT/config_table.out 2>&1
