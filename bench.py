#!/usr/bin/env python3
"""bench.py — throughput of the hot path on N MI355X GPUs (one process per GPU, pairs sharded, no data collective).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" = one pass of the hot path over one synthetic 700x700 source/reference pair per GPU (BASELINE config 2).
`value` = pairs/s summed over all ranks, inputs resident in HBM when the timed region starts.

Extra objects on the JSON line:
  roofline     — dominant kernel (PatchMatch Jacobi step at the finest level, k_pm_step<1>): algorithmic GB/s from the
                 device eval counter x SURVEY §8d bytes-per-eval, over the kernel time measured with HIP events on the
                 library's own stream (nct_pm_bench_run), vs the 8 TB/s HBM peak.
  cpu_baseline — the CPU oracle (oracle/liboracle.so, "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def level_geometry(h, w):
    """Feature pyramid of a HxW image, coarse -> fine (Caffe ceil-mode pooling) + rs_max (main.cu:77-83)."""
    dims, hh, ww = [], h, w
    for c in (64, 128, 256, 512, 512):
        dims.append((c, hh, ww))
        hh, ww = (hh + 1) // 2, (ww + 1) // 2
    dims = dims[::-1]
    max_len = max(h, w)
    rs = [max_len // 16, max_len // 32, max_len // 64, 32, 32]
    return [(c, hh, ww, r) for (c, hh, ww), r in zip(dims, rs)]


def pm_bytes(evals, n_queries, n_launches, C):
    """candidate tiles + query tile once per launch + NNF/dist read-modify-write."""
    return evals * 9 * C * 4 + n_launches * n_queries * 9 * C * 4 + n_launches * n_queries * 5 * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=700, help="image side (BASELINE config 2 = 700)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import nct
    import synth
    ctx = nct.Context(local_rank)
    levels = level_geometry(args.size, args.size)

    # ---- synthetic, device-resident inputs (one pair per rank; seeds follow SURVEY §8d: 1000+2i / 1001+2i)
    ctxs = []
    for li, (C, h, w, rs) in enumerate(levels):
        c = ctx if li == 0 else nct.Context(local_rank)
        fa = synth.features(1000 + 2 * rank + 10 * li, C, h, w)
        fb = synth.features(1001 + 2 * rank + 10 * li, C, h, w)
        c.pm_bench_setup(fa, fb)
        ctxs.append(c)

    def one_pair(count=False):
        """PatchMatch L=5->1, both directions (the second direction reuses the same feature pair, roles swapped in
        cost terms only: same geometry => same work)."""
        tot_ms, per_level = 0.0, []
        for (C, h, w, rs), c in zip(levels, ctxs):
            ms_l, ev_l = 0.0, 0
            for d in range(2):
                ms, ev, _, _ = c.pm_bench_run(iters=10, rs_max=rs, seed=17 + d, count_evals=count)
                ms_l += ms
                ev_l += ev or 0
            per_level.append((ms_l, ev_l))
            tot_ms += ms_l
        return tot_ms, per_level

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pair()
    barrier()
    t0 = time.perf_counter()
    kern_levels = None
    for _ in range(args.steps):
        _, kern_levels = one_pair()
    for c in ctxs:
        c.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel (finest level, C=64): un-timed counting pass + the timed pass' event time
    _, counted = one_pair(count=True)
    C, h, w, rs = levels[-1]
    n_launch = 2 * 41
    evals = counted[-1][1]
    fin_ms = kern_levels[-1][0]
    alg_bytes = pm_bytes(evals, h * w, n_launch, C)
    achieved = alg_bytes / (fin_ms * 1e-3) / 1e9
    tot_evals = sum(e for _, e in counted)
    tot_bytes = sum(pm_bytes(e, hh * ww, n_launch, cc) for (_, e), (cc, hh, ww, _) in zip(counted, levels))
    tot_ms = sum(m for m, _ in kern_levels)

    out = {
        "metric": "700x700 pairs/sec end-to-end L=5->1; PatchMatch HBM GB/s vs peak",
        "value": world * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PARTIAL: PatchMatch stage only (L=5->1, both directions, iters=10) of one {args.size}x{args.size} pair; "
                               "VGG19 + colour stage not yet in the timed region",
                   "pairs_per_gpu_per_step": 1, "parallelism": f"pairs sharded over {world} GPU(s), no data collective"},
        "roofline": {"bound": "hbm", "kernel": "k_pm_step<1> (C=64, %dx%d)" % (h, w), "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "launches": n_launch, "avg_launch_ms": fin_ms / n_launch, "algorithmic_bytes_per_launch": alg_bytes / n_launch,
                     "evals": evals},
        "patchmatch_all_levels": {"evals": tot_evals, "algorithmic_GB": tot_bytes / 1e9, "kernel_ms": tot_ms,
                                  "algorithmic_GBps": tot_bytes / (tot_ms * 1e-3) / 1e9,
                                  "per_level_ms": [m for m, _ in kern_levels]},
    }

    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(levels, tot_evals)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(levels, evals_per_pair):
    """Oracle ('port') PatchMatch on a bounded sample: one direction of level 4 (88x88x512) with few iterations,
    all host cores (OpenMP). Reported as pairs/s by scaling with distance evaluations weighted by 9*C MACs."""
    import oracle_bind
    import synth
    orc = oracle_bind.load()
    C, h, w, rs = levels[1]
    a = orc.feat_normalize(synth.features(1, C, h, w))
    b = orc.feat_normalize(synth.features(2, C, h, w))
    nnf0 = orc.nnf_init(h, w, h, w)
    iters = 2
    t0 = time.perf_counter()
    orc.patchmatch(a, b, nnf0, iters=iters, rs_max=rs, seed=1)
    dt = time.perf_counter() - t0
    ev = orc.last_evals()
    mac_rate = ev * 9 * C / dt                      # MAC/s of the oracle on this host
    # MACs per pair: evals per level * 9*C (both directions) — use the model E*n per level
    macs_pair = 0
    for (c, hh, ww, r) in levels:
        R = 0
        m = min(r, max(hh, ww))
        while m >= 1:
            R += 1
            m //= 2
        macs_pair += 2 * hh * ww * (1 + 10 * (16 + R)) * 9 * c
    cores = os.cpu_count() or 1
    return {"value": mac_rate / macs_pair, "unit": "pairs/s (PatchMatch stage only)", "cores": cores, "kind": "port",
            "sample": f"oracle orc_patchmatch, one direction, level 4 ({h}x{w}x{C}), {iters} iters, {ev} evals in {dt:.2f} s, "
                      f"OpenMP on {cores} threads; scaled to a pair by 9*C MACs per eval over both directions of 5 levels"}


if __name__ == "__main__":
    main()
