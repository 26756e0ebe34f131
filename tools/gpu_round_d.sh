#!/bin/bash
# Round-2 closing measurements: parity of the final build, config table (every BASELINE shape), C4 frame with both job
# splits, the > 32 Mpx fallback, native multi-device renderer timing on one device listed twice.
set -u
OUT=gpurun_out/${1:-r2l}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 300 python tools/config_table.py --jobs 131072 196608 --out $OUT/config_table.jsonl 2>$OUT/config_table.err | tee $OUT/config_table.txt
timeout 300 python tools/config_table.py --jobs 131072 --only X2560 X3072 XHD X4K --out $OUT/config_table.jsonl 2>>$OUT/config_table.err | tee -a $OUT/config_table.txt
timeout 300 python bench.py --config c4 --steps 3 --warmup 1 > $OUT/bench_c4_n1_1m.json 2> $OUT/bench_c4_n1_1m.err; tail -c 400 $OUT/bench_c4_n1_1m.json
timeout 300 python bench.py --config c4 --jobs 524288 --steps 3 --warmup 1 > $OUT/bench_c4_n1_512k.json 2> $OUT/bench_c4_n1_512k.err; tail -c 400 $OUT/bench_c4_n1_512k.json
timeout 200 python tools/perf_explore.py --blocks 256 --variants 0x0 --jobs 131072 --size 8192 --n 1e9 --out $OUT/perf_8192.jsonl 2>$OUT/perf_8192.err | tail -2
timeout 300 python - > $OUT/native_multi.txt 2>&1 <<'PY'
import time, numpy as np, strange_attractor_renderer_amd as S
for size, iters in ((2048, 1_000_000_000), (4096, 2_500_000_000)):
    cfg = S.Config.poisson_saturne(iterations=iters, width=size, height=size, transparent=0)
    for devs in ([0], [0, 0]):
        r = S.ParallelRenderer(devices=devs, units=16384 * len(devs), seed=1) if len(devs) > 1 else S.ParallelRenderer(units=16384, seed=1)
        S.render_parallel(r, cfg, 8)
        t0 = time.perf_counter(); img = S.render_parallel(r, cfg, 8); dt = (time.perf_counter() - t0) * 1e3
        print(size, "devices", devs, "render_parallel ms %.2f" % dt, r.last_timing() if len(devs) > 1 else "")
        r.shutdown()
PY
cat $OUT/native_multi.txt
