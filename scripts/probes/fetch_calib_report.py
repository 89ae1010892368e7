"""Turns the counter CSVs of the fetch_calib passes into the calibration table (profiles/round4_fetch_calibration.md).
usage: python scripts/probes/fetch_calib_report.py <dir with pass_*/ subdirs> """
import csv, glob, json, os, sys
d = sys.argv[1]
known = {"k_stream": (1 << 30), "k_pm_rows": (1 << 20) * 768}
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in known:
            rows.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
print("| kernel | known bytes per dispatch | counter | raw value per dispatch (all dispatches) | bytes implied | implied / known |")
print("|---|---|---|---|---|---|")
for (k, c), v in sorted(rows.items()):
    for x in v:
        unit = 1024.0 if c in ("FETCH_SIZE", "WRITE_SIZE") else (64.0 if "RDREQ" in c and "32B" not in c else 32.0)
        print(f"| `{k}` | {known[k]} | {c} | {x:.1f} | {x * unit:.0f} ({'KB' if unit == 1024 else str(int(unit)) + ' B per request'}) | {x * unit / known[k]:.4f} |")
