#!/bin/bash
# PMC passes focused on the iterate kernel (count-only and full), to find what its waves wait for.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_iterate_${1:-x}
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/tools/perf_explore.py --variants 0x13 0x3 --jobs 131072 --blocks 256 --out /tmp/pmc_x.jsonl"
cd /tmp && export TMPDIR=/tmp
p() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- $CMD > /dev/null 2> $OUT/$name.err; }
p act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_BUSY_CYCLES
p wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY
p fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
p misc SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$OUT/*/")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_iterate" in k: agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in agg.items():
        print(d.split("/")[-2], k, {c: round(sum(v)/len(v)/1e6,1) for c,v in cs.items()})
PY
