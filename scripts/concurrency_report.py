"""How the GPU is shared when several pairs are in flight: from a rocprofv3 --kernel-trace of `bench.py --inflight 4` (kernel_trace.csv), over the window of the timed steps:
union busy time, average number of kernels running, and per kernel class the time it runs ALONE against the time it overlaps with kernels of other queues (and with which).
usage: python scripts/concurrency_report.py <kernel_trace.csv> [skip_fraction=0.5]   (the first skip_fraction of the trace — warm-ups — is dropped)"""
import csv, sys, collections, heapq

def cls(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    for key, c in (("k_conv3x3", "conv"), ("k_pm_step<1", "pm_fine_rs"), ("k_pm_prop<1", "pm_fine_prop"), ("k_pm_", "pm_coarse"), ("k_s1_apply<true>", "s1_apply_big"), ("k_s1_", "s1_other"),
                   ("k_cg_", "wls_fine"), ("k_mg_block", "wls_fine"), ("k_mg_down<6, 32, 14", "wls_fine"), ("k_mg_up<6, 32, 16, float, false", "wls_fine"), ("k_mg_", "wls_coarse"), ("k_pcg", "wls_coarse"),
                   ("k_knn", "knn"), ("k_vote", "votes"), ("k_km_", "kmeans"), ("rocprim", "rocprim")):
        if key in n: return c
    return "other"

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), cls(r["Kernel_Name"])) for r in rows)
T0, T1 = ev[0][0], max(e[1] for e in ev)
w0 = T0 + int((T1 - T0) * skip)
pts = []
for s, e, c in ev:
    if e <= w0: continue
    s = max(s, w0)
    pts.append((s, 1, c)); pts.append((e, -1, c))
pts.sort()
active = collections.Counter(); n_active = 0; last = w0
busy = 0; weighted = 0
alone = collections.Counter(); shared = collections.Counter(); total = collections.Counter(); with_ = collections.defaultdict(collections.Counter)
hist = collections.Counter()
for t, d, c in pts:
    dt = t - last
    if dt > 0 and n_active > 0:
        busy += dt; weighted += dt * n_active; hist[min(n_active, 6)] += dt
        for k, v in active.items():
            if v <= 0: continue
            total[k] += dt
            if n_active == v: alone[k] += dt
            else:
                shared[k] += dt
                for k2, v2 in active.items():
                    if v2 > 0 and k2 != k: with_[k][k2] += dt
    last = t
    active[c] += d; n_active += d
span = last - w0
print(f"window {span / 1e6:.1f} ms: busy (union) {busy / 1e6:.1f} ms = {100 * busy / span:.1f} %, idle {100 - 100 * busy / span:.1f} %, mean kernels running while busy {weighted / busy:.2f}")
print("time with n kernels running:", {k: f"{100 * v / span:.1f} %" for k, v in sorted(hist.items())})
print(f"{'class':14s} {'present ms':>10s} {'alone %':>8s}  most often together with")
for k, v in sorted(total.items(), key=lambda kv: -kv[1]):
    top = ", ".join(f"{k2} {100 * t / v:.0f} %" for k2, t in with_[k].most_common(4))
    print(f"{k:14s} {v / 1e6:10.1f} {100 * alone[k] / v:8.1f}  {top}")
