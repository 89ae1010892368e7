#!/usr/bin/env python3
"""Per-launch counter means grouped by (kernel name prefix, grid size): the conv kernel runs every VGG layer, the grid tells the layers apart.
usage: pmc_by_grid.py <dir with p*/…_counter_collection.csv> <kernel name prefix[|prefix...]>"""
import csv, glob, os, sys
from collections import defaultdict
root, prefix = sys.argv[1], sys.argv[2]
tab = defaultdict(lambda: defaultdict(lambda: [0.0, set()]))
for p in sorted(glob.glob(os.path.join(root, "p*"))):
    for f in glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(tuple(prefix.split("|"))):
                key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
                e = tab[key][r["Counter_Name"]]; e[0] += float(r["Counter_Value"]); e[1].add(r["Dispatch_Id"])
for key in sorted(tab):
    row = {c: v[0] / len(v[1]) for c, v in tab[key].items()}
    n = max(len(v[1]) for v in tab[key].values())
    line = f"{key[0]} grid={key[1]} launches={n}"
    if "GRBM_GUI_ACTIVE" in row and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
        cyc = row["GRBM_GUI_ACTIVE"] / 8.0
        line += f" cycles={cyc:,.0f} mfma_util={row['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}"
    if "SQ_WAIT_INST_ANY" in row and "SQ_WAVE_CYCLES" in row:
        line += f" waitcnt_share={row['SQ_WAIT_INST_ANY'] / row['SQ_WAVE_CYCLES']:.3f}"
    print(line)
    for c in sorted(row):
        print(f"    {c}: {row[c]:,.0f}")
