#!/usr/bin/env python3
"""Per-launch means of every counter collected by scripts/pmc_patchmatch.sh for the kernels whose name starts with the given prefix.
usage: pmc_summary.py <dir with p*/…_counter_collection.csv> <kernel name prefix[|prefix...]>"""
import csv, glob, os, sys
from collections import defaultdict
root, prefix = sys.argv[1], sys.argv[2]
for p in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(p):
        continue
    f = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print(p, "no counter file"); continue
    tot, disp = defaultdict(float), defaultdict(set)
    name = None
    for r in csv.DictReader(open(f[0])):
        if r["Kernel_Name"].startswith(tuple(prefix.split("|"))):
            name = r["Kernel_Name"].split("(")[0]
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
    # keep the level's launches only: the largest grid (the finest level) is selected through the dispatch count of the pair: all launches of the prefix
    for c in sorted(tot):
        n = len(disp[c])
        print(f"{os.path.basename(p)} {name} {c}: {tot[c] / n:,.0f} per launch over {n} launches")
