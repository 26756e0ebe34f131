#!/bin/bash
# Round 6: the three questions round 5's profiles left open (VERDICT "Next round" #5). Each bounded to a few GPU-minutes.
#   (i)   the +0.75 ms dispatch pairs of profiles/r05_rocprofv3_summary.md: the same 24 dispatches timed by HIP events alone and
#         with `rocprofv3 --kernel-trace` attached (both clocks on the same dispatches)
#   (ii)  where the 36.75 GB of the configs[3] share's iterate kernel go: HBM or Infinity Cache (the DRAM-side counters this
#         rocprofv3 offers), one pass per counter group
#   (iii) read-only: the compute partition mode of the leased box
# Usage (on the GPU box, from the repo root): tools/r06_questions.sh [i|ii|iii|all]
set -u
WHAT=${1:-all}
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_questions
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT

if [ "$WHAT" = all ] || [ "$WHAT" = iii ]; then
  { rocm-smi --showcomputepartition; rocm-smi --showmemorypartition; rocm-smi --showtopo 2>/dev/null | head -30; } > $OUT/partition.txt 2>&1
  grep -i -E "partition" $OUT/partition.txt | head -6
fi

if [ "$WHAT" = all ] || [ "$WHAT" = i ]; then
  for rep in 1 2; do
    timeout 300 python $R/tools/dispatch_times.py --steps 24 --json $OUT/events_only_$rep.json > /dev/null 2> $OUT/events_only_$rep.err
  done
  timeout 300 python $R/tools/dispatch_times.py --steps 24 --no-prefetch --json $OUT/events_only_noprefetch.json > /dev/null 2> $OUT/events_only_noprefetch.err
  for rep in 1 2; do
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/traced_$rep -o t -- python $R/tools/dispatch_times.py --steps 24 --json $OUT/traced_events_$rep.json > /dev/null 2> $OUT/traced_$rep.err
  done
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/traced_noprefetch -o t -- python $R/tools/dispatch_times.py --steps 24 --no-prefetch --json $OUT/traced_events_noprefetch.json > /dev/null 2> $OUT/traced_noprefetch.err
  python $R/tools/summarize_dispatch_times.py $OUT > $OUT/dispatch_summary.md 2>&1
  cat $OUT/dispatch_summary.md
fi

if [ "$WHAT" = all ] || [ "$WHAT" = ii ]; then
  rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCC|TCP|MALL|HBM|UMC|DF)[A-Z0-9_]*(DRAM|MALL|HBM|IO|GMI|EA0_RD|EA0_WR|UNCACHED|32B|64B|128B|PROBE)[A-Za-z0-9_]*" | sort -u > $OUT/counter_names_memside.txt
  wc -l $OUT/counter_names_memside.txt
  SHARE="python $R/tools/config_table.py --only C4/8 --reps 3 --out /tmp/r06_share.jsonl"
  run_pmc() { # name counters...
    local name=$1; shift
    timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/share_$name -o pmc -- $SHARE > $OUT/share_$name.out 2> $OUT/share_$name.err
    echo "pass $name: rc=$?"
  }
  run_pmc ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run_pmc dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum
  run_pmc rddram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum
  run_pmc wrdram TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum
  run_pmc io TCC_EA0_RDREQ_IO_sum TCC_EA0_RDREQ_GMI_sum TCC_EA0_WRREQ_IO_sum TCC_EA0_WRREQ_GMI_sum
  run_pmc fetch FETCH_SIZE
  run_pmc write WRITE_SIZE
  run_pmc l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  # the same DRAM-side pass for the headline kernel (2048^2): what its 6.95 GB are
  BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --sustained-seconds 0 --no-traffic --no-parity"
  timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/c2_rddram -o pmc -- $BENCH > $OUT/c2_rddram.out 2> $OUT/c2_rddram.err
  timeout 400 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/c2_wrdram -o pmc -- $BENCH > $OUT/c2_wrdram.out 2> $OUT/c2_wrdram.err
  python $R/tools/summarize_memside.py $OUT > $OUT/memside_summary.md 2>&1
  cat $OUT/memside_summary.md
fi
