"""Where a single pair in flight leaves the GPU idle: gaps between consecutive kernels of a rocprofv3 --kernel-trace of scripts/s1_levels.py (one pair at a time).
usage: python scripts/gap_report.py <kernel_trace.csv> [pairs_in_trace=5]   -> totals for the LAST pair: busy (union of kernel intervals), idle, idle by following kernel"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]) for r in rows))
# pairs are separated by the k_bgr2lab ... first kernel of a pair: find the starts of "k_prep"/first conv of each pair via the biggest gaps
gaps = sorted(((ev[i + 1][0] - max(e[1] for e in ev[max(0, i - 8):i + 1]), i) for i in range(len(ev) - 1)), reverse=True)
npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cuts = sorted(i for _, i in gaps[:npairs - 1])
last = ev[cuts[-1] + 1:]
t0, t1 = last[0][0], max(e[1] for e in last)
busy = 0; cur_end = t0; idle_by = collections.Counter(); idle_n = collections.Counter()
for s, e, n in last:
    if s > cur_end:
        idle_by[n] += s - cur_end; idle_n[n] += 1
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print("last pair: %d kernels, span %.2f ms, busy %.2f ms, idle %.2f ms" % (len(last), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6))
small = sum(v for k, v in idle_by.items())
hist = collections.Counter()
cur_end = t0
for s, e, n in last:
    if s > cur_end:
        g = (s - cur_end) / 1e3
        hist["<2us" if g < 2 else "2-5us" if g < 5 else "5-20us" if g < 20 else "20-100us" if g < 100 else ">100us"] += s - cur_end
    cur_end = max(cur_end, e)
print("idle by gap size (ms):", {k: round(v / 1e6, 2) for k, v in hist.items()})
print("idle before kernel (top 25):")
for n, v in idle_by.most_common(25):
    print("  %-62s %7.3f ms in %4d gaps (%.1f us each)" % (n, v / 1e6, idle_n[n], v / 1e3 / idle_n[n]))
