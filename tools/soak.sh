#!/bin/bash
# Soak: the seeded random GPU tests over OTHER seeds (SAR_FUZZ_BASE = FIRST..LAST), until the first failure.
#   bash tools/soak.sh FIRST LAST [outdir]      (on the GPU box; ~25 s per base)
first=${1:-1}; last=${2:-10}; out=${3:-gpurun_out/soak}; mkdir -p $out
sel="seeded_random or random_configurations or custom_attractors"
for k in $(seq $first $last); do
    SAR_FUZZ_BASE=$k timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_dist.py -m gpu -x -q -k "$sel" \
        -p no:cacheprovider > $out/base_$k.log 2>&1
    rc=$?
    echo "base $k: rc=$rc $(tail -1 $out/base_$k.log)" | tee -a $out/summary.log
    if [ $rc -ne 0 ]; then tail -40 $out/base_$k.log; exit 1; fi
done
