#!/bin/bash
# round 3, visit 1: s_memtime attribution of k_iterate_split (prof variant) next to the product build on the same box
out=gpurun_out/r03_prof; mkdir -p $out
P=$PWD/strange_attractor_renderer_amd
timeout 300 python tools/config_table.py --only C2 C3 C4/8 --reps 3 --out $out/table_base.jsonl > $out/base.out 2> $out/base.err
SAR_LIBRARY=$P/libsar_hip_prof.so timeout 300 python tools/config_table.py --only C2 C3 C4/8 --reps 3 --out $out/table_prof.jsonl > $out/prof.out 2> $out/prof.err
SAR_SPLIT=1 SAR_LIBRARY=$P/libsar_hip_prof.so timeout 300 python tools/config_table.py --only C2 --reps 3 --out $out/table_prof_whole.jsonl > $out/prof_whole.out 2> $out/prof_whole.err
cat $out/base.out; echo; cat $out/prof.out; grep prof $out/prof.err | sort | uniq -c | head -20; grep prof $out/prof_whole.err | sort | uniq -c | head
