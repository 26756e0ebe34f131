#!/bin/bash
# PMC passes over four shapes of the iterate kernel (full / count-only at 2, 3, 4 waves per SIMD): what saturates?
# Usage: tools/pmc_study.sh <tag> [stager]
set -u
TAG=${1:-x}
ST=${2:-0}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_study_$TAG
mkdir -p $OUT
cat > /tmp/pmc_cmd.sh <<EOS
#!/bin/bash
cd $GRAFT_REPO_ROOT
PE="python tools/perf_explore.py --blocks 256 --out /tmp/pmc_x.jsonl --opt stager=$ST"
\$PE --jobs 131072 --records 28 --variants 0x3 0x13
\$PE --jobs 196608 --records 20 --variants 0x3
\$PE --jobs 262144 --records 12 --variants 0x13
EOS
chmod +x /tmp/pmc_cmd.sh
cd /tmp && export TMPDIR=/tmp
p() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o pmc -- /tmp/pmc_cmd.sh > /dev/null 2> $OUT/$name.err; }
p insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES
p act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU
p ta TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
p lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_LDS_ATOMIC_RETURN GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
python - > $OUT/summary.txt <<PY
import csv,glob,collections,re
for d in sorted(glob.glob("$OUT/*/")):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_iterate_lean" in k:
                m=re.search(r"k_iterate_lean<(.*?)>",k)
                agg[m.group(1) if m else k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,cs in sorted(agg.items()):
        print(d.split("/")[-2], k, {c: round(sum(v)/len(v)/1e6,2) for c,v in sorted(cs.items())})
PY
cat $OUT/summary.txt
