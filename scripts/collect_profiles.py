#!/usr/bin/env python3
"""Turns the output of scripts/final_measure.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/:
round2_bench.json, round2_pmc_patchmatch.json, round2_e2e_kernels.md, round2_workloads.md, round2_pytest_gpu.txt.
usage: python scripts/collect_profiles.py <tag>"""
import csv, json, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")


def jline(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("{"):
            return json.loads(line)
    return None


bench = jline("bench.json")
json.dump(bench, open(os.path.join(dst, "round2_bench.json"), "w"), indent=1)
pmc = bench["roofline"].get("pmc")
if pmc and bench["roofline"].get("traffic_source") == "live":
    json.dump(pmc, open(os.path.join(dst, "round2_pmc_patchmatch.json"), "w"), indent=1)
open(os.path.join(dst, "round2_pytest_gpu.txt"), "w").write("".join(open(os.path.join(src, "pytest_gpu.txt")).readlines()[-6:]))

# ---- per-kernel table of the one-pair-in-flight profiled run
prof = jline("prof_bench.json")
rows = list(csv.DictReader(open(os.path.join(src, "prof", "b_kernel_stats.csv"))))
pairs = 3 * 4          # context warm-up + (1 warm-up + 2 timed) steps + host-to-host + latency / stage / roofline pairs, one in flight: see bench.py
calls = sum(int(r["Calls"]) for r in rows)
total = sum(int(r["TotalDurationNs"]) for r in rows)
pm = [r for r in rows if r["Name"].startswith("void k_pm_step<1, 1,")][0]
L = [f"# round 2 — `rocprofv3 --kernel-trace --stats` of `python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency-flag` (one pair in flight), MI355X", "",
     f"Build id {prof['build_id']} (the build of profiles/round2_bench.json: {bench['build_id']}). The run processes {pairs} pairs (context warm-up, 1 warm-up + 2 timed steps, the host-to-host",
     f"region, the latency / stage / roofline pairs): divide calls and totals by {pairs} for one 700x700 pair. Bench line of this profiled run: {prof['value']:.2f} pairs/s, single pair {prof['single_pair_ms']:.1f} ms (tracing on);",
     f"its event-timed average launch of `{prof['roofline']['kernel'].split(' (')[0]}` is {prof['roofline']['avg_launch_ms'] * 1e3:.1f} us, the trace's own average below {float(pm['AverageNs']) / 1e3:.1f} us (the un-traced bench: {bench['roofline']['avg_launch_ms'] * 1e3:.1f} us).",
     f"Total: {calls} kernel launches = {calls // pairs} per pair, {total / 1e6:.1f} ms of kernel time = {total / 1e6 / pairs:.1f} ms per pair.", "",
     "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for r in rows:
    L.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | {r['Percentage']} |")
open(os.path.join(dst, "round2_e2e_kernels.md"), "w").write("\n".join(L) + "\n")

# ---- other workloads: refresh the numbers of the table rows in place (the prose around them is edited by hand)
wl = {"pair700": bench, "pair1000": jline("bench_1000.json"), "pair256l5": jline("bench_256l5.json"), "batch64": jline("bench_batch.json"), "mixed256": jline("bench_mixed.json")}
p = os.path.join(dst, "round2_workloads.md")
txt = open(p).read()
for name, b in wl.items():
    if not b:
        continue
    def sub(m, b=b):
        cells = m.group(0).split("|")
        cells[4] = f" {b['value']:.2f} "; cells[5] = f" {b['ms_per_step']:.1f} "; cells[6] = f" {b['single_pair_ms']:.1f} ms "
        return "|".join(cells)
    txt = re.sub(r"^\| %s[^\n]*$" % re.escape(name), sub, txt, count=1, flags=re.M)
txt = re.sub(r"over the same 4-in-flight batches: [0-9.]+ pairs/s vs [0-9.]+ resident", f"over the same 4-in-flight batches: {bench['host_to_host_pairs_per_s']:.2f} pairs/s vs {bench['value']:.2f} resident", txt)
two = jline("bench_2rank_gloo.json")
if two:
    txt = re.sub(r"n_gpus 2, [0-9.]+ pairs/s in total", f"n_gpus 2, {two['value']:.2f} pairs/s in total", txt)
open(p, "w").write(txt)
print("bench:", bench["value"], "pairs/s, single pair", bench["single_pair_ms"], "ms, roofline frac", bench["roofline"]["frac"], "launches/pair", calls // pairs, "build", bench["build_id"])
for name, b in wl.items():
    if b: print(name, round(b["value"], 2), "pairs/s", round(b["single_pair_ms"], 1), "ms")
print(open(os.path.join(src, "pm_modes.log")).read())
