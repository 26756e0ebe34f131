"""Condenses tools/r06_questions.sh (ii): the memory-side request counters of the iterate kernels, mean per dispatch —
how many of the L2's fabric requests (TCC_EA0_*) went on to DRAM (…_DRAM) and how many the Infinity Cache served."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
out = {}
print("# Memory-side counters (mean per dispatch)\n")
for case, kern in (("share", "k_iterate_split"), ("c2", "k_iterate_split")):
    agg = defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, case + "_*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in (r.get("Kernel_Name") or ""):
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        continue
    c = {n: sum(v) / len(v) for n, v in agg.items()}
    out[case] = c
    print(f"## {case}: `{kern}`\n")
    print("| counter | mean per dispatch |")
    print("|---|---|")
    for n in sorted(c):
        print(f"| {n} | {c[n]:.6g} |")
    rd, rdd = c.get("TCC_EA0_RDREQ_sum"), c.get("TCC_EA0_RDREQ_DRAM_sum")
    wr, wrd = c.get("TCC_EA0_WRREQ_sum"), c.get("TCC_EA0_WRREQ_DRAM_sum")
    if rd and rdd is not None:
        print(f"\n* read requests that went to DRAM: {rdd / rd:.3f} of {rd:.4g}")
    if wr and wrd is not None:
        print(f"* write requests that went to DRAM: {wrd / wr:.3f} of {wr:.4g}")
    print()
json.dump(out, open(os.path.join(root, "memside.json"), "w"), indent=1)
