#!/bin/bash
# Closing visit of a round on ONE box: tools/gpu_round.sh (whole -m gpu suite, variant suites, smoke, default bench line with
# the CPU baseline, config table), the other bench records (configs[3] and configs[4] at N=1, the one-command SCALE form with
# two ranks sharing the GPU over gloo, the C-ABI renderer over 1 device and 8 shards), then the rocprofv3 evidence
# (tools/profile_round.sh: kernel statistics + PMC passes of the headline workload; tools/pmc_c4.sh: the 4096^2 kernels).
#   tools/gpu_round_final.sh <tag>      -> gpurun_out/<tag>/..., gpurun_out/profiles_<tag>/..., gpurun_out/pmc_c4_<tag>/...
tag=${1:-rXX}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
bash tools/gpu_round.sh $tag
B="--no-cpu-baseline --no-pipeline --sustained-seconds 0 --no-traffic"
python bench.py --config c4 --steps 6 --warmup 2 $B > $out/bench_c4_n1.json 2> $out/bench_c4_n1.err
python bench.py --config c5 --steps 1440 --warmup 720 > $out/bench_c5_n1.json 2> $out/bench_c5_n1.err
timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --extras-seconds 400 > $out/bench_n2_gloo_one_gpu.json 2> $out/bench_n2_gloo_one_gpu.err
timeout 900 python bench.py --gpus 8 --steps 4 --warmup 1 --extras-seconds 600 > $out/bench_n8_gloo_one_gpu.json 2> $out/bench_n8_gloo_one_gpu.err
# the cold sweeps by the stand-alone tool as well (a fresh process each: its first repetition is a PROCESS-cold sweep)
for f in 360 45; do for m in readback hbm; do python tools/cold_sweep.py --frames $f --mode $m --reps 3 --pause 1.0 --json $out/cold_${f}_$m.json > /dev/null 2>> $out/cold.err; done; done
for c in c2 c4; do python bench.py --native --gpus 8 --config $c --steps 4 --warmup 2 > $out/bench_native_8shards_$c.json 2> /dev/null; done
python bench.py --native --gpus 1 --steps 20 --warmup 3 > $out/bench_native_1dev_c2.json 2> /dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], {k: d[k] for k in ("host_ms_per_step",) if k in d},
              {k: (d[k].get("ms_per_step"), d[k].get("value")) for k in ("strong_c4",) if k in d and isinstance(d[k], dict)})
    except Exception as e:
        print(f, "unreadable:", e)
PY
bash tools/profile_round.sh $tag > $out/profile_round.log 2>&1; tail -5 $out/profile_round.log
bash tools/pmc_c4.sh $tag both > $out/pmc_c4.log 2>&1; grep -c derived $out/pmc_c4.log
bash tools/profile_c5.sh $tag > $out/profile_c5.log 2>&1; tail -30 $out/profile_c5.log
