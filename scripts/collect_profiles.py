#!/usr/bin/env python3
"""Turns the output of scripts/final_measure.sh (gpurun_out/<tag>/) into the tracked summaries under profiles/ (round 3 names):
round3_bench.json, round3_pmc_patchmatch.json, round3_e2e_kernels.md, round3_workloads.md, round3_pmc_patchmatch_levels.md, round3_pmc_vgg_mfma.md,
round3_wls_rtol_sweep.json, round3_pytest_gpu.txt.
usage: python scripts/collect_profiles.py <tag>"""
import csv, json, os, re, sys, collections

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")
R = "round3"


def jline(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    for line in open(p):
        if line.startswith("{"):
            return json.loads(line)
    return None


bench = jline("bench.json")
json.dump(bench, open(os.path.join(dst, f"{R}_bench.json"), "w"), indent=1)
pmc = bench["roofline"].get("pmc")
if pmc and bench["roofline"].get("traffic_source") == "live":
    json.dump(pmc, open(os.path.join(dst, f"{R}_pmc_patchmatch.json"), "w"), indent=1)
open(os.path.join(dst, f"{R}_pytest_gpu.txt"), "w").write("".join(open(os.path.join(src, "pytest_gpu.txt")).readlines()[-6:]))
sweep = os.path.join(src, "wls_rtol_sweep.json")
if os.path.exists(sweep) and os.path.getsize(sweep) > 10:
    json.dump(json.load(open(sweep)), open(os.path.join(dst, f"{R}_wls_rtol_sweep.json"), "w"), indent=1)

# ---- per-kernel table of the one-pair-in-flight profiled run
prof = jline("prof_bench.json")
rows = list(csv.DictReader(open(os.path.join(src, "prof", "b_kernel_stats.csv"))))
pairs = 3 * 4 + 2      # context warm-up + (1 warm-up + 2 timed) steps + host-to-host + the single-pair latency (warm run + best of three) / latency-flag / stage / roofline pairs,
                       # one in flight: see bench.py
calls = sum(int(r["Calls"]) for r in rows)
total = sum(int(r["TotalDurationNs"]) for r in rows)
pm = [r for r in rows if r["Name"].startswith("void k_pm_step<1, 1,")][0]
L = [f"# round 3 — `rocprofv3 --kernel-trace --stats` of `python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-latency-flag` (one pair in flight), MI355X", "",
     f"Build id {prof['build_id']} (the build of profiles/{R}_bench.json: {bench['build_id']}). The run processes {pairs} pairs (context warm-up, 1 warm-up + 2 timed steps, the host-to-host",
     f"region, the latency / stage / roofline pairs): divide calls and totals by {pairs} for one 700x700 pair. Bench line of this profiled run: {prof['value']:.2f} pairs/s, single pair {prof['single_pair_ms']:.1f} ms (tracing on);",
     f"its event-timed average launch of `{prof['roofline']['kernel'].split(' (')[0]}` is {prof['roofline']['avg_launch_ms'] * 1e3:.1f} us, the trace's own average below {float(pm['AverageNs']) / 1e3:.1f} us (the un-traced bench: {bench['roofline']['avg_launch_ms'] * 1e3:.1f} us).",
     f"Total: {calls} kernel launches = {calls // pairs} per pair, {total / 1e6:.1f} ms of kernel time = {total / 1e6 / pairs:.1f} ms per pair.", "",
     "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for r in rows:
    L.append(f"| `{r['Name'][:120]}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {int(r['MinNs']) / 1e3:.1f} | {int(r['MaxNs']) / 1e3:.1f} | {r['Percentage']} |")
open(os.path.join(dst, f"{R}_e2e_kernels.md"), "w").write("\n".join(L) + "\n")

# ---- other workloads
wl = {"pair700": bench, "pair1000": jline("bench_1000.json"), "pair256l5": jline("bench_256l5.json"), "batch64": jline("bench_batch.json"), "mixed256": jline("bench_mixed.json")}
two = jline("bench_2rank_gloo.json")
cli = None
p = os.path.join(src, "cli_mixed.txt")
if os.path.exists(p):
    for line in open(p):
        if line.startswith("{"):
            cli = json.loads(line)
W = [f"# round 3 — the other bench workloads (BASELINE configs 1, 3, 4, 5) on one MI355X, build {bench['build_id']}", "",
     "`python bench.py --workload <name>` (4 pairs in flight per GPU unless noted; `value` = pairs/s; batch workloads are host buffer in -> host buffer out):", "",
     "| workload | BASELINE config | what a step is | pairs/s | ms per step | single pair |", "|---|---|---|---|---|---|"]
desc = {"pair700": ("2", "4 resident 700x700 pairs"), "pair1000": ("4", "4 resident 1000x1000 pairs"), "pair256l5": ("1", "4 resident 256x256 pairs, L=5 only"),
        "batch64": ("3 (one GPU's share: 16 of 64)", "16 pairs of 700x700 through nct_process_pair"), "mixed256": ("5 (one GPU's share: 32 of 256)", "32 pairs with sides 256..1000, dynamic tickets")}
for name, b in wl.items():
    if b:
        W.append(f"| {name} | {desc[name][0]} | {desc[name][1]} | {b['value']:.2f} | {b['ms_per_step']:.1f} | {b['single_pair_ms']:.1f} ms |")
W += ["", f"Host buffer in -> host buffer out over the same 4-in-flight 700x700 batches (`host_to_host_pairs_per_s`): {bench['host_to_host_pairs_per_s']:.2f} pairs/s vs {bench['value']:.2f} resident.",
      f"Single 700x700 pair: {bench['single_pair_ms']:.1f} ms; with `NCT_FLAG_LATENCY` (the CLI's `-inflight 1`): {bench['single_pair_latency_flag_ms']:.1f} ms, output identical: {bench['latency_flag_output_identical']}."]
if two:
    W.append(f"Two ranks (gloo) sharing GPU 0 through `bench.py --gpus 2 --dist-backend gloo --device-override 0 --inflight 2`: n_gpus 2, {two['value']:.2f} pairs/s in total.")
if cli and wl["mixed256"]:
    W += ["", "## The CLI against the library on the same mixed batch (VERDICT r2 #7: PNG work on the GPU workers' critical path)",
          f"`scripts/mixed_batch_cli.py 32 4`: the first 32 pairs of the mixed256 workload as PNG files through `neural_color_transfer -inflight 4` (decode + shrink + encode included):",
          f"**{cli['cli_pairs_per_s_io_pool']:.2f} pairs/s with the I/O pool** (`-io 2`, the default for one GPU) vs {cli['cli_pairs_per_s_io0']:.2f} with `-io 0` (every worker does its own zlib, the round-2 behaviour)",
          f"vs {wl['mixed256']['value']:.2f} pairs/s for `bench.py --workload mixed256 --batch 32` (host buffers, no files): the pool runs at {100 * cli['cli_pairs_per_s_io_pool'] / wl['mixed256']['value']:.0f} % of the library rate."]
cb = bench.get("cpu_baseline")
if cb:
    W += ["", "## CPU baseline", f"{cb['sample']}: **{cb['value']:.5f} pairs/s on {cb['cores']} threads** (1 thread, scaled from a 64x64 pair: {cb['value_1thread']:.5f})."]
open(os.path.join(dst, f"{R}_workloads.md"), "w").write("\n".join(W) + "\n")

# ---- PatchMatch counters per instantiation + per dispatch of the finest level
pm_all = os.path.join(src, "pmc_pm_all.txt")
if os.path.exists(pm_all):
    d = collections.defaultdict(dict)
    for l in open(pm_all):
        m = re.match(r"(\S+) void (k_pm_step<[^>]+>) (\S+): ([\d,]+) per launch over (\d+)", l)
        if m:
            d[m.group(2)][m.group(3)] = float(m.group(4).replace(",", "")); d[m.group(2)]["n"] = int(m.group(5))
    names = sorted(d)
    T = [f"# round 3 — counters of every PatchMatch instantiation the pipeline launches (one real 700x700 pair, `scripts/final_measure.sh`: six counter-only passes), build {bench['build_id']}", "",
         "Per launch (mean over the level's 41 launches; C = 512 runs two levels = 82). cycles = GRBM_GUI_ACTIVE / 8 XCDs. FETCH_SIZE x 2 (gfx950 tallies 128-B requests as 64 B) counts",
         "L2 misses INCLUDING Infinity-Cache hits: the feature maps of every level fit the 256 MiB MALL, so the HBM share of `fabric GB/s` is unknown and <= it.", "",
         "| | " + " | ".join(f"`{n}`" for n in names) + " |", "|---|" + "---|" * len(names)]
    lvl = {"k_pm_step<1, 1, 2, 2, 8>": "C = 64, 700^2 (finest), ROWREJECT, 8 lanes/query", "k_pm_step<2, 1, 2, 1, 16>": "C = 128, 350^2, ROWREJECT", "k_pm_step<4, 0, 1, 1, 16>": "C = 256, 175^2, PLAIN", "k_pm_step<8, 0, 1, 1, 16>": "C = 512, 88^2 + 44^2, PLAIN"}
    T.append("| level | " + " | ".join(lvl.get(n, "") for n in names) + " |")
    def row(label, f):
        T.append(f"| {label} | " + " | ".join(f(d[n]) for n in names) + " |")
    cyc = lambda v: v["GRBM_GUI_ACTIVE"] / 8.0
    row("launch, cycles / us at 2.4 GHz", lambda v: f"{cyc(v):,.0f} / {cyc(v) / 2400:.0f}")
    row("waves", lambda v: f"{v['SQ_WAVES']:,.0f}")
    row("VALU instructions (issue share of the launch)", lambda v: f"{v['SQ_INSTS_VALU'] / 1e6:.1f} M ({100 * v['SQ_INSTS_VALU'] * 4 / 1024 / cyc(v):.0f} %)")
    row("vector-memory reads = TA wavefronts (TA busy)", lambda v: f"{v['SQ_INSTS_VMEM_RD'] / 1e6:.2f} M ({100 * v['TA_BUSY_avr'] / cyc(v):.0f} %)")
    row("L1 data return busy (TD)", lambda v: f"{100 * v['TD_TD_BUSY_sum'] / 256 / cyc(v):.0f} %")
    row("wave cycles in s_waitcnt / issuing", lambda v: f"{100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.0f} % / {100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.0f} %")
    row("L1 hit rate", lambda v: f"{100 * (1 - v['TCP_TCC_READ_REQ_sum'] / v['TCP_TOTAL_CACHE_ACCESSES_sum']):.0f} %")
    row("L2 hit rate", lambda v: f"{100 * v['TCC_HIT_sum'] / v['TCC_REQ_sum']:.0f} %")
    row("fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE)", lambda v: f"{(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024) / 1e9:.3f} GB")
    row("fabric GB/s (over the launch's cycles at 2.4 GHz) = share of 8 TB/s", lambda v: f"{(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024) / (cyc(v) / 2.4e9) / 1e9:,.0f} = {(v['FETCH_SIZE'] * 2048 + v['WRITE_SIZE'] * 1024) / (cyc(v) / 2.4e9) / 8e12:.2f}")
    row("L1 return bandwidth used (VMEM reads x 1 KB / (cycles x 256 CUs x 64 B))", lambda v: f"{v['SQ_INSTS_VMEM_RD'] * 1024 / (cyc(v) * 256 * 64):.2f}")
    per = os.path.join(src, "pmc_pm_finest_per_dispatch.txt")
    if os.path.exists(per):
        T += ["", "## The 41 launches of the finest level one by one (`scripts/pmc_per_dispatch.py`)", "",
              "launch 0 = initial distances; 1-40 = 10 iterations x jumps 8, 4, 2, 1; every fourth (4, 8, ..., 40) adds the random search (6 candidates per query, up to +-32 px from the match):",
              "those ten launches move most of the level's fabric bytes, at the rate the fabric sustains; the propagation-only launches mostly hit L1/L2 and — with the candidates that cannot win",
              "skipped — are short.", ""] + [l.rstrip("\n") for l in open(per)]
    open(os.path.join(dst, f"{R}_pmc_patchmatch_levels.md"), "w").write("\n".join(T) + "\n")

vg = os.path.join(src, "vgg_mfma_by_grid.txt")
if os.path.exists(vg):
    V = [f"# round 3 — MFMA counters of the shipped conv kernel `k_conv3x3_mfma2b<WCO, CT, POOL>` (VGG19 forward 700x700 -> conv5_1, three forwards; two counter-only passes), build {bench['build_id']}", "",
         "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (launch cycles x 1024 SIMDs), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs (`scripts/pmc_by_grid.py`; the grid size tells the layers apart:",
         "491520 = conv1_1 at 700^2, 493568 = conv1_2 + pool, 245760 = conv2_1, 247808 = conv2_2 + pool, 122880 = conv3_x, 65536 = conv4_x, 32768 = conv5_1). The cycle base is the GRBM counter's, so the",
         "shares compare layers and builds, not an absolute peak fraction (conv5_1's 124 workgroups could not exceed 0.48). bench.py repeats the two passes live (`vgg_mfma.mfma_util`:",
         f"{bench['vgg_mfma'].get('mfma_util')} over all conv launches).", "", "```"] + [l.rstrip("\n") for l in open(vg)] + ["```"]
    open(os.path.join(dst, f"{R}_pmc_vgg_mfma.md"), "w").write("\n".join(V) + "\n")

print("bench:", bench["value"], "pairs/s, single pair", bench["single_pair_ms"], "ms, roofline frac", bench["roofline"]["frac"], "launches/pair", calls // pairs, "build", bench["build_id"])
for name, b in wl.items():
    if b: print(name, round(b["value"], 2), "pairs/s", round(b["single_pair_ms"], 1), "ms")
print(cli)
print(open(os.path.join(src, "pm_modes.log")).read())
