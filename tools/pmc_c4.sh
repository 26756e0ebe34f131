#!/bin/bash
# PMC passes for the kernels of BASELINE configs[3] (4096^2), which profiles/r03_pmc.json did not cover:
#   share  one GPU's eighth of the frame (1.25e9 iterations, 131 072 jobs): k_iterate_split<28,u16,PH=1> in one round
#   full   the whole 1e10-iteration frame on one GPU: k_iterate_lean<28,u16,pool> + k_bin_accumulate<28,4,packed>
# Each counter group in its own run, --kernel-trace only (never with other trace domains).
# Usage (on the GPU box, from the repo root): tools/pmc_c4.sh <tag> [share|full|both] [extra config_table options]
set -u
TAG=${1:-r04}
WHAT=${2:-both}
shift 2 2>/dev/null
OPTS="$*"
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_c4_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHARE="python $GRAFT_REPO_ROOT/tools/config_table.py --only C4/8 --reps 3 --out /tmp/pmc_c4_share.jsonl ${OPTS:+--option $OPTS}"
FULL="python $GRAFT_REPO_ROOT/bench.py --config c4 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline --sustained-seconds 0 --no-traffic"
run_pmc() { # case name counters...
  local cmd=$1 name=$2; shift 2
  local c="$SHARE"; [ "$cmd" = full ] && c="$FULL"
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${cmd}_$name -o pmc -- $c > $OUT/${cmd}_$name.out 2> $OUT/${cmd}_$name.err
}
for cmd in share full; do
  [ "$WHAT" = both ] || [ "$WHAT" = $cmd ] || continue
  c="$SHARE"; [ "$cmd" = full ] && c="$FULL"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${cmd}_trace -o t -- $c > $OUT/${cmd}_trace.out 2> $OUT/${cmd}_trace.err
  run_pmc $cmd fetch FETCH_SIZE
  run_pmc $cmd write WRITE_SIZE
  run_pmc $cmd l2 TCC_HIT_sum TCC_MISS_sum
  run_pmc $cmd l2req TCC_REQ_sum TCC_ATOMIC_sum
  run_pmc $cmd ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
  run_pmc $cmd rdsize TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  run_pmc $cmd wrsize TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
  run_pmc $cmd insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run_pmc $cmd act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY
  run_pmc $cmd lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
  run_pmc $cmd grbm GRBM_GUI_ACTIVE GRBM_COUNT
done
cd $GRAFT_REPO_ROOT
python tools/summarize_pmc_c4.py $OUT $OUT/pmc_c4.json > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
