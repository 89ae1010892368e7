"""Per WLS solve of a rocprofv3 --kernel-trace of scripts/wls_levels.py: span, busy time, idle, mean duration of the finest-level kernels. usage: python scripts/wls_solve_report.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
starts = [i for i, e in enumerate(ev) if "k_pcg_start<" in e[2]]
ends = [i for i, e in enumerate(ev) if "k_pcg_finish" in e[2]]
print("solves:", len(starts))
for n, (a, b) in enumerate(list(zip(starts, ends))[-10:]):
    seg = ev[a:b + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = 0; cur = t0; gaps = []
    for s, e, k in seg:
        if s > cur: gaps.append((s - cur) / 1e3)
        if e > cur: busy += e - max(s, cur); cur = e
    d = collections.defaultdict(list)
    for s, e, k in seg:
        nm = k.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        d[nm].append((e - s) / 1e3)
    big = sum(1 for g in gaps if g > 4)
    print("solve %d: %d kernels span %.2f ms busy %.2f idle %.2f (gaps > 4 us: %d, sum %.2f ms)" % (n, len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, big, sum(g for g in gaps if g > 4) / 1e3))
    print("   " + "  ".join("%s %.1f" % (k[:28], sum(v) / len(v)) for k, v in sorted(d.items()) if len(v) >= 10))
# ---- what runs between k_wls_system (end of the colour stage) and k_pcg_start of each solve
ws = [i for i, e in enumerate(ev) if e[2].startswith("k_wls_system")]
for n, (a, b) in enumerate(list(zip(ws, starts))[-5:]):
    seg = ev[a:b + 1]
    print("setup %d: %d kernels, span %.3f ms" % (n, len(seg), (seg[-1][0] - seg[0][0]) / 1e6))
    cur = seg[0][1]
    for s, e, k in seg[1:]:
        nm = k.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:36]
        if s - cur > 20000 or e - s > 50000: print("    gap %.1f us before %s (%.1f us)" % ((s - cur) / 1e3, nm, (e - s) / 1e3))
        cur = max(cur, e)
