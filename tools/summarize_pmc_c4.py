"""Condenses a tools/pmc_c4.sh output directory: per case (share / full) and kernel, the mean per dispatch of every counter
and the derived figures the round's attribution uses (L2 hit rate, L2 requests and VALU instructions per visit, LDS bank
conflict share, clock, HBM traffic). Writes the JSON committed as profiles/rNN_pmc_c4.json."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
KERNELS = ("k_iterate_split", "k_iterate_lean", "k_bin_accumulate", "k_fold_resolve", "k_warmup", "k_colorize_gas")
VISITS = {"share": 131072 * (1250000000 // 131072)}  # counted iterations per launch of the share's single launch


def short(name):
    for k in KERNELS:
        if k in name:
            m = re.search(k + r"<([^>]*)>", name)
            return k + ("<" + m.group(1) + ">" if m else "")
    return None


out = {}
print(f"# PMC passes over the 4096^2 kernels — {os.path.basename(root)}\n")
for case in ("share", "full"):
    agg = defaultdict(lambda: defaultdict(list))
    for d in sorted(glob.glob(os.path.join(root, case + "_*"))):
        if not os.path.isdir(d) or d.endswith("_trace"):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name") or "")
                if k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, case + "_trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if not agg and not dur:
        continue
    print(f"## {case}\n")
    out[case] = {}
    for k in sorted(set(agg) | set(dur)):
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        if dur[k]:
            d = sorted(dur[k])
            c["duration_ms_mean"] = sum(d) / len(d)
            c["duration_ms_min"] = d[0]
            c["dispatches_traced"] = len(d)
        out[case][k] = c
        print(f"### `{k}`\n")
        print("| counter | mean per dispatch |")
        print("|---|---|")
        for n in sorted(c):
            print(f"| {n} | {c[n]:.6g} |")
        der = []
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            der.append(f"L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            der.append(f"LDS bank-conflict share {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.2f}")
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            der.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
        if "GRBM_GUI_ACTIVE" in c and c.get("duration_ms_mean"):
            der.append(f"clock ~ {c['GRBM_GUI_ACTIVE'] / 8 / c['duration_ms_mean'] / 1e6:.2f} GHz (GRBM_GUI_ACTIVE, summed over the 8 XCDs, / 8 / traced duration)")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            der.append(f"HBM traffic (2*FETCH_SIZE + WRITE_SIZE) * 1024 = {(2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per dispatch "
                       f"(FETCH_SIZE uncorrected: {c['FETCH_SIZE'] * 1024 / 1e9:.2f} GB read)")
        if case in VISITS and k.startswith("k_iterate"):
            v = VISITS[case]
            if "SQ_INSTS_VALU" in c:
                der.append(f"VALU per trajectory iteration {c['SQ_INSTS_VALU'] * 64 / v:.1f} (SQ_INSTS_VALU counts wave instructions)")
            if "TCC_REQ_sum" in c:
                der.append(f"L2 requests per visit {c['TCC_REQ_sum'] / v:.2f}")
        if der:
            print("\nderived: " + "; ".join(der))
        print()
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
