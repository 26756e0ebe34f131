#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box:
#   1. kernel trace + stats of the default bench command (timing; no counters)
#   2. PMC passes, each in its own run (gpurun refuses --pmc combined with sys/hip/hsa traces)
# Usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipeline --sustained-seconds 0 --no-traffic"
cd /tmp && export TMPDIR=/tmp
# the timing pass runs the bench's default step count, so that the tracer's average per kernel and the bench line's HIP-event
# average (roofline.kernel_ms) are averages over the same dispatches; the counter passes below need only a few
BENCH20="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline --sustained-seconds 0 --no-traffic"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH20 > $OUT/trace_bench.json 2> $OUT/trace.err
run_pmc() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc l2 TCC_HIT_sum TCC_MISS_sum
run_pmc l2req TCC_REQ_sum TCC_ATOMIC_sum
run_pmc ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run_pmc eaatom TCC_EA0_ATOMIC_sum
# what a fabric request weighs: reads by size (FETCH_SIZE tallies 64 B per request whatever its size), writes in 32-byte units
run_pmc rdsize TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run_pmc wrsize TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
run_pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
run_pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pmc insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run_pmc act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|TCC_.*ATOMIC|TCC_EA0" | head -60 > $OUT/counter_names.txt
cd $GRAFT_REPO_ROOT
python tools/summarize_profiles.py $OUT $OUT/pmc.json > $OUT/SUMMARY.md 2>&1
cat $OUT/SUMMARY.md
