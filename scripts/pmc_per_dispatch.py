#!/usr/bin/env python3
"""Per-dispatch table of the counters collected by scripts/final_measure.sh for one kernel (one row per launch, in launch order): duration, L1 / L2 hit rates,
fabric bytes and rate — shows how unlike the 41 launches of a PatchMatch level are (init / propagation / propagation + random search).
usage: pmc_per_dispatch.py <dir with p*/…_counter_collection.csv> <kernel name prefix[|prefix...]>"""
import csv, glob, os, sys, collections
root, prefix = sys.argv[1], sys.argv[2]
tabs = []
for p in sorted(glob.glob(os.path.join(root, "p*"))):
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(tuple(prefix.split("|"))):
                d = rows.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
                d[r["Counter_Name"]] = float(r["Counter_Value"])
        tabs.append(list(rows.values()))
n = min(len(t) for t in tabs) if tabs else 0
print("| launch | us | L1 accesses M | L1 hit % | L2 requests M | L2 hit % | fabric MB (FETCH_SIZE x 2) | fabric TB/s | VMEM rd M | VALU M |")
print("|---|---|---|---|---|---|---|---|---|---|")
for i in range(n):
    m = {}
    for t in tabs:
        for k, v in t[i].items():
            if k != "us": m[k] = v
            elif "FETCH_SIZE" in t[i]: m["us_fetch"] = v
            else: m.setdefault("us", v)
    g = lambda k: m.get(k, float("nan"))
    fb = g("FETCH_SIZE") * 2048.0
    print(f"| {i} | {g('us'):.1f} | {g('TCP_TOTAL_CACHE_ACCESSES_sum') / 1e6:.1f} | {100 * (1 - g('TCP_TCC_READ_REQ_sum') / g('TCP_TOTAL_CACHE_ACCESSES_sum')):.1f} | {g('TCC_REQ_sum') / 1e6:.1f} | "
          f"{100 * g('TCC_HIT_sum') / g('TCC_REQ_sum'):.1f} | {fb / 1e6:.0f} | {fb / 1e12 / (g('us_fetch') * 1e-6):.2f} | {g('SQ_INSTS_VMEM_RD') / 1e6:.2f} | {g('SQ_INSTS_VALU') / 1e6:.1f} |")
