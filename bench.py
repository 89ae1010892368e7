#!/usr/bin/env python3
"""bench.py — end-to-end throughput of the hot path on N MI355X GPUs (one process per GPU, pairs sharded, no data collective).

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0. With N > 1 and no
torch.distributed environment the script re-launches itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per
GPU over RCCL); launched by torchrun directly it just checks WORLD_SIZE == N.

A "step" = one full pass of the hot path (VGG19 features, k-means, L=5->1 PatchMatch both ways, BDS votes, kNN graphs, nonlocal +
WLS colour solves, re-predicts) over one batch of `--inflight` (default 4) synthetic 700x700 source/reference pairs per GPU —
BASELINE config 2 (`--workload pair700`). The pairs of a batch are independent jobs (own context, streams, arena, host thread)
that run concurrently on the GPU (`--inflight 1` gives the single-pair latency, reported as `single_pair_ms` either way).
`value` = pairs/s summed over ranks with the inputs resident in HBM when the timed region starts (nct_pair_run); the host-buffer
in -> host-buffer out rate over the same batch shape (nct_process_pair: + 2.9 MB of PCIe per pair) is `host_to_host_pairs_per_s`.
Other workloads (parity-test configurations of BASELINE.json, selectable for scaling runs): pair1000 (config 4), pair256l5 (config 1:
256x256, L=5 only), batch64 (config 3: 64 pairs of 700x700 per step, static i mod N, strong scaling), mixed256 (config 5: 256 pairs
with sides 256..1000 per step, dynamic tickets from the rendezvous store = work stealing across ranks without a collective).

Extra objects:
  roofline     — the dominant kernel AS THE PIPELINE RUNS IT: the finest PatchMatch level's 41 launches, k_pm_step<1, 1, 2, 2, 8> / k_pm_prop<1, 1, 2, 2, 8> (C = 64, both directions per
                 launch, unit-norm features with the exact row rejection). avg launch time = HIP events recorded on the library's stream around the
                 level's 41 launches of a real pair; `traffic` = fabric-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE
                 x2 gfx950 correction + WRITE_SIZE, separate passes) of THIS build — collected live when rocprofv3 is on PATH, else
                 read from profiles/ only if the recorded build id equals this libnct.so's; `achieved` = traffic / launch time,
                 `frac` = achieved / 8 TB/s. The SURVEY §8d no-reuse byte model is reported next to it (`algorithmic_GBs`).
  roofline_color — the colour solvers' full-resolution kernels (57 % of a pair in round 3): S1 operator / direction / update and the WLS PCG's finest V-cycle
                 legs, operator and vector update, each with its compulsory bytes per launch (stated per pixel), the event-timed average of single launches
                 (NCT_FLAG_TIME_KERNELS: HIP events on the pair's stream around the launch) and the fraction of the 8 TB/s peak; plus SURVEY §8(d)'s per-iteration
                 byte models over the iteration times.
  roofline_1000 — the same PatchMatch roofline on BASELINE config 4 (1000x1000: the finest level's 512 MB footprint exceeds the 256 MiB Infinity Cache, so its
                 fabric-side bytes are HBM bytes), from two live counter passes.
  cpu_baseline — the CPU oracle ("port") end-to-end on a bounded sample of the same workload, on this box's host cores.
  stages_ms    — per-stage device time of one extra pair (events on the stream; not part of the timed region).
  natural      — the reference's own demo photographs (the five image pairs of demo/example/pairs.txt, sizes as shipped, synthetic weights), one pair in flight, next to
                 synthetic pairs of the same sizes: ms per pair, pairs/s, the natural/synthetic ratio, WLS iterations (outside the timed region; `available: false` where the
                 photographs are not staged — they are never committed). `--workload natural` makes them the timed workload.
"""
import argparse
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time
import zlib

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))
sys.path.insert(0, os.path.join(REPO, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
MALL_BYTES = 256.0 * 1024 * 1024     # Infinity Cache
WORKLOADS = ("pair700", "pair1000", "pair256l5", "batch64", "mixed256", "natural")
NATURAL = (("in0", "tar0"), ("in1", "tar1"), ("in2", "tar2"), ("in3", "tar3"), ("in4", "tar4"))      # demo/example/pairs.txt lines 1-4 and 7: the five image pairs of the demo at bds 2.0 (tests/natural_inputs.py stages the photographs)


def load_natural():
    """the five demo pairs as BGR arrays, or None where the photographs are not staged (they are never committed: tests/natural_inputs.py)"""
    try:
        import natural_inputs
        from PIL import Image
        if not natural_inputs.stage():
            return None
        ld = lambda n: np.ascontiguousarray(np.asarray(Image.open(os.path.join(natural_inputs.DIR, n + ".png")).convert("RGB"))[..., ::-1])
        return [(ld(a), ld(b)) for a, b in NATURAL]
    except Exception:       # noqa: BLE001
        return None


def natural_extra(ctx, prm, synth):
    """One pair in flight, outside the timed region: the reference's own demo photographs (synthetic weights) next to tests/synth.py pairs of the SAME sizes — the
    driver-visible natural/synthetic ratio. median of 3 runs each."""
    imgs = load_natural()
    if imgs is None:
        return {"available": False, "why": "demo photographs not staged on this box (tests/natural_inputs.py; never committed)"}

    def med(src, ref):
        ctx.pair_upload(src, ref)
        ctx.pair_run(prm)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); ctx.pair_run(prm); ts.append(time.perf_counter() - t)
        tm = ctx.pair_run(prm, want_timing=True)
        return 1e3 * sorted(ts)[1], tm
    out = {"available": True, "pairs": {}, "basis": "one pair in flight, inputs resident, median of 3; synthetic = tests/synth.py images of the same sizes"}
    nat, syn = [], []
    for (a, b), (src, ref) in zip(NATURAL, imgs):
        n_ms, n_tm = med(src, ref)
        s_ms, s_tm = med(synth.image(1000, *src.shape[:2]), synth.image(1001, *ref.shape[:2]))
        out["pairs"][f"{a}_{b}"] = {"size": list(src.shape[:2]) + list(ref.shape[:2]), "ms": round(n_ms, 2), "synthetic_ms": round(s_ms, 2),
                                    "wls_iters": n_tm["wls_iters"], "synthetic_wls_iters": s_tm["wls_iters"],
                                    "stages_ms": {k: round(n_tm[k], 2) for k in ("vgg_ms", "patchmatch_ms", "vote_ms", "knn_ms", "nonlocal_ms", "wls_ms")}}
        nat.append(n_ms); syn.append(s_ms)
    out["mean_ms"] = round(sum(nat) / len(nat), 2); out["synthetic_mean_ms"] = round(sum(syn) / len(syn), 2)
    out["pairs_per_s"] = round(1e3 / out["mean_ms"], 3); out["natural_over_synthetic"] = round(out["mean_ms"] / out["synthetic_mean_ms"], 3)
    return out


def pm_bytes(evals, n_queries, n_launches, C):
    """candidate tiles + query tile once per launch + NNF/dist read-modify-write."""
    return evals * 9 * C * 4 + n_launches * n_queries * 9 * C * 4 + n_launches * n_queries * 5 * 8


def lib_build_id():
    """Identity of the kernel build: sha256 over the library's sources (csrc/*, include/nct.h) — reproducible across rebuilds of the same tree, unlike the .so bytes."""
    import glob
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "neural-color-transfer_amd", "csrc", "*")) + [os.path.join(REPO, "include", "nct.h"), os.path.join(REPO, "neural-color-transfer_amd", "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def lib_so_id(nct):
    """sha256[:16] of the shared object as loaded by this process (ctypes handle's path)."""
    try:
        return hashlib.sha256(open(nct.lib()._name, "rb").read()).hexdigest()[:16]
    except Exception as e:      # noqa: BLE001
        return f"unavailable: {type(e).__name__}"


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pair700", choices=WORKLOADS)
    ap.add_argument("--size", type=int, default=0, help="[test hook] image side for the pair* workloads (0 = the workload's own)")
    ap.add_argument("--inflight", type=int, default=4, help="independent pairs in flight per GPU (contexts + host threads)")
    ap.add_argument("--batch", type=int, default=0, help="[test hook] pairs per step for batch64 / mixed256 (0 = 64 / 256)")
    ap.add_argument("--dist-backend", default="nccl", help="[test hook] torch.distributed backend (gloo exercises the N>1 path on a 1-GPU box)")
    ap.add_argument("--device-override", type=int, default=-1, help="[test hook] run every rank on this device instead of LOCAL_RANK")
    ap.add_argument("--dist-single", default="auto", choices=("auto", "on", "off"),
                    help="N = 1: run the barrier / MAX-reduce through a ONE-rank RCCL process group as well, so that the code the 8-GPU run depends on executes in every "
                         "1-GPU run (auto: try, fall back to no group and say why in rccl_error; on: fail if RCCL cannot start; off: no group, rounds 1-4)")
    ap.add_argument("--force-dist", action="store_true", help="alias of --dist-single on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time the oracle on the full SxS pair even on a small host (the default from 32 host threads on: ~2 min)")
    ap.add_argument("--cpu-baseline-sample", action="store_true", help="time the oracle on the bounded 350x350 sample only (seconds) and scale by pixel count")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-natural", action="store_true", help="skip the extra single-pair runs on the reference's demo photographs (key `natural`)")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 PMC passes (traffic falls back to profiles/)")
    ap.add_argument("--no-pmc-1000", action="store_true", help="skip the two counter passes on the 1000x1000 pair (roofline_1000)")
    ap.add_argument("--no-vary", action="store_true", help="every step re-runs the same resident pairs (rounds 1-3); default: two resident sets of pairs per GPU, alternating step by step")
    ap.add_argument("--no-latency-flag", action="store_true", help="skip the two extra single-pair runs with NCT_FLAG_LATENCY (kernel-trace runs: keeps the launch count per pair comparable)")
    ap.add_argument("--print-launch", action="store_true", help="[test hook] print the torchrun command --gpus N would re-launch with, and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: re-launch under torchrun (the driver may also do this itself)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--print-launch"]
        if args.print_launch:
            print(json.dumps(cmd)); sys.exit(0)
        sys.exit(subprocess.call(cmd))

    # stdout carries ONE line, the JSON record: everything else any library writes to file descriptor 1 (RCCL prints a version banner through C stdio at exit, the
    # ROCm runtime may print warnings) is sent to stderr for the whole run; the record goes to the saved descriptor at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if args.device_override >= 0:
        local_rank = args.device_override
    import torch
    dist = None
    rccl_ranks = None
    rccl_error = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm
            rccl_ranks = dist.get_world_size()
        else:
            dist.init_process_group(args.dist_backend)
    elif (args.force_dist or args.dist_single != "off") and args.dist_backend == "nccl":
        # one rank: the SAME branch the N > 1 run takes (communicator on the device, device-side barrier, MAX all-reduce of a cuda tensor in timed_region) — VERDICT r4 item 6
        import datetime
        import torch.distributed as dist
        try:
            torch.cuda.set_device(local_rank)
            if "MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ:
                os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(free_port())
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
            dist.barrier()
            rccl_ranks = dist.get_world_size()
        except Exception as e:      # noqa: BLE001
            if args.force_dist or args.dist_single == "on":
                raise
            rccl_error = f"{type(e).__name__}: {str(e)[:200]}"
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:       # noqa: BLE001
                pass
            dist = None

    import nct
    import synth
    from caffemodel_io import synthetic_vgg19, write_caffemodel
    from nct.shard import shard_pairs, timed_region

    K = max(1, args.inflight)
    NSETS = 1 if (args.no_vary or not args.workload.startswith("pair")) else 2       # resident input sets per GPU (pair workloads): step i runs set i mod NSETS
    ctxs = [nct.Context(local_rank) for _ in range(K * NSETS)]
    ctx = ctxs[0]
    # synthetic VGG19 (He-normal, seed 19) serialised as a V1-format caffemodel and loaded through the ingest path
    ws, bs = synthetic_vgg19(19)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "VGG_ILSVRC_19_layers.caffemodel")
        write_caffemodel(path, ws, bs, fmt="v1")
        # parsed once per process, uploaded once per GPU; the other contexts in flight on this GPU share the read-only device copy (as the CLI does)
        model = nct.Model(path)
        ctxs[0].vgg19_load_model(model)
        model.close()
        for c in ctxs[1:]:
            c.vgg19_share_weights(ctxs[0])

    prm = nct.Params.default()
    wl = args.workload
    S = args.size or {"pair700": 700, "pair1000": 1000, "pair256l5": 256, "batch64": 700, "mixed256": 0, "natural": 0}[wl]
    if wl == "pair256l5":
        prm.levels = 1

    def sync():
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()

    def run_workers(fn):
        """fn(k) on K host threads (ctypes releases the GIL inside the library calls)."""
        if K == 1:
            fn(0)
            return
        ths = [threading.Thread(target=fn, args=(k,)) for k in range(1, K)]
        for t in ths:
            t.start()
        fn(0)
        for t in ths:
            t.join()

    host_to_host = None
    if wl.startswith("pair"):
        # weak scaling: K pairs per GPU per step; pair i of the job: seeds 1000+2i / 1001+2i, pair i -> rank i mod N;
        # slot k of this rank holds its k-th pair, resident on the device before the timed region starts
        def pair(i):
            return synth.image(1000 + 2 * i, S, S), synth.image(1001 + 2 * i, S, S)
        # NSETS x K contexts per GPU, each with its own resident pair (distinct images): step i runs the K pairs of set i mod NSETS, so consecutive steps do not
        # re-read the same images / features out of a warm Infinity Cache. Pair index of (set, rank, slot) = set * world * K + the rank's share of world * K.
        my_pairs = [sidx * world * K + pi for sidx in range(NSETS) for pi in shard_pairs(world * K, rank, world)]
        host = [pair(pi) for pi in my_pairs]
        for c, (src, ref) in zip(ctxs, host):
            c.pair_upload(src, ref)
        if NSETS > 1 and args.warmup < NSETS:
            for sidx in range(NSETS):                  # every resident set runs once before the clock (arena blocks, code objects), whatever --warmup says
                run_workers(lambda k: ctxs[sidx * K + k].pair_run(prm))
        elapsed = timed_region(lambda i: run_workers(lambda k: ctxs[(i % NSETS) * K + k].pair_run(prm)), args.steps, args.warmup, dist=dist, sync=sync,
                               device=torch.device("cuda", local_rank) if rccl_ranks else None)
        pairs_per_step = world * K
        scaling = "weak"
        e2 = timed_region(lambda i: run_workers(lambda k: ctxs[(i % NSETS) * K + k].process_pair(host[(i % NSETS) * K + k][0], host[(i % NSETS) * K + k][1], prm)), args.steps, 0,
                          dist=dist, sync=sync, device=torch.device("cuda", local_rank) if rccl_ranks else None)
        host_to_host = pairs_per_step * args.steps / e2
        src, ref = host[0]
        desc = (f"{K} independent {S}x{S} source/reference pair(s) in flight per GPU per step ({NSETS} resident set(s) of distinct pairs, alternating), " +
                ("L=5 only (BASELINE config 1)" if wl == "pair256l5" else "full L=5->1 pyramid") + ", bds=2.0, Config.h defaults"
                + {"pair700": " (BASELINE config 2)", "pair1000": " (BASELINE config 4)"}.get(wl, ""))
    else:
        # a step = one whole batch through nct_process_pair (host in -> host out), K workers per GPU
        nb = args.batch or {"batch64": 64, "natural": 15}.get(wl, 256)
        nat_imgs = None
        if wl == "natural":
            # the reference's own demo photographs (sizes as shipped, synthetic weights): each of the five pairs nb / 5 times per step, static i mod N
            nat_imgs = load_natural()
            if nat_imgs is None:
                raise SystemExit("bench.py --workload natural: the demo photographs are not staged (python tests/natural_inputs.py where the reference is mounted, or NCT_DEMO_DIR)")
            sizes = [nat_imgs[i % len(NATURAL)][0].shape[:2] + nat_imgs[i % len(NATURAL)][1].shape[:2] for i in range(nb)]
            mine = shard_pairs(nb, rank, world)
        elif wl == "batch64":
            sizes = [(S, S, S, S)] * nb
            mine = shard_pairs(nb, rank, world)                     # static i mod N (config 3)
        else:
            sizes = []
            for i in range(nb):                                     # sides ~U{256..1000}, independently for S and R, seed 5000+i
                r = np.random.default_rng(5000 + i).integers(256, 1001, 4)
                sizes.append(tuple(int(v) for v in r))
            mine = None                                             # dynamic tickets (config 5)
        cache = {}

        def images(i):
            if nat_imgs is not None:
                return nat_imgs[i % len(NATURAL)]
            if i not in cache:
                sh, sw, rh, rw = sizes[i]
                cache[i] = (synth.image(1000 + 2 * i, sh, sw), synth.image(1001 + 2 * i, rh, rw))
            return cache[i]
        if mine is not None:
            for i in mine:
                images(i)
        else:
            for i in range(nb):
                images(i)                                           # every rank may draw any ticket
        store = dist.distributed_c10d._get_default_store() if dist is not None else None
        lock = threading.Lock()
        local = {"next": 0, "step": -1, "done": 0}

        def ticket(step_id):
            """next unprocessed pair of this step: a shared counter in the rendezvous store (work stealing across ranks, no collective)"""
            if store is not None and mine is None:
                return store.add(f"ticket{step_id}", 1) - 1
            with lock:
                t = local["next"]; local["next"] += 1
                return t

        def batch_step(i):
            with lock:
                local["next"] = 0
            todo = mine if mine is not None else None

            def worker(k):
                while True:
                    t = ticket(i)
                    if todo is not None:
                        if t >= len(todo):
                            return
                        idx = todo[t]
                    else:
                        if t >= nb:
                            return
                        idx = t
                    s_, r_ = images(idx)
                    ctxs[k].process_pair(s_, r_, prm)
                    with lock:
                        local["done"] += 1
            run_workers(worker)
        elapsed = timed_region(batch_step, args.steps, args.warmup, dist=dist, sync=sync, device=torch.device("cuda", local_rank) if rccl_ranks else None)
        pairs_per_step = nb
        scaling = "strong"
        host_to_host = pairs_per_step * args.steps / elapsed
        src, ref = images(mine[0] if mine else 0)
        ctx.pair_upload(src, ref)
        desc = (f"{nb} pairs per step through nct_process_pair (host in -> host out), {K} workers per GPU; " +
                {"batch64": "700x700, static i mod N (BASELINE config 3)", "natural": "the reference's demo photographs in0/tar0 ... in4/tar4 (520x600 .. 700x528 sources), static i mod N"}.get(wl, "sides 256..1000, dynamic tickets (BASELINE config 5)"))

    # single-pair latency (nothing else on the GPU) and per-stage device times of one more pair
    ctx.pair_run(prm)
    lat = []
    for _ in range(5):                       # five samples: a single one carries the box's hiccups (105.8 vs 97-98 ms seen once); median = the headline, min beside it
        t1 = time.perf_counter()
        ctx.pair_run(prm)
        lat.append(time.perf_counter() - t1)
    single_pair_s = sorted(lat)[len(lat) // 2]
    # the same pair with NCT_FLAG_LATENCY (a-/b-halves of the WLS solves on two streams: more launches, same bytes, same result)
    plat = nct.Params.default()
    for k, _ in nct.Params._fields_:
        setattr(plat, k, getattr(prm, k))
    plat.flags |= nct.FLAG_LATENCY
    single_pair_latency_s, latency_checksum = None, None
    if not args.no_latency_flag:
        ctx.pair_run(plat)
        t1 = time.perf_counter()
        ctx.pair_run(plat)
        single_pair_latency_s = time.perf_counter() - t1
        latency_checksum = int(np.asarray(ctx.pair_download(), dtype=np.uint64).sum())
    stages = ctx.pair_run(prm, want_timing=True)
    out = ctx.pair_download()

    res = {
        "metric": "700x700 pairs/sec end-to-end L=5->1; PatchMatch HBM GB/s vs peak",
        "value": pairs_per_step * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": desc + "; synthetic He-init VGG19 loaded from a V1 caffemodel", "name": wl,
                   "pairs_per_gpu_per_step": K if scaling == "weak" else pairs_per_step / world,
                   "parallelism": f"pairs sharded over {world} GPU(s), no data collective"},
        "rccl_ranks": rccl_ranks, "rccl_error": rccl_error,
        "host_to_host_pairs_per_s": host_to_host,
        "value_basis": "inputs resident in HBM when the timed region starts (nct_pair_run; the task contract's definition of `value`). SURVEY 8(d)'s host-in -> host-out "
                       "wall over the same batches (nct_process_pair: + 2.9 MB of PCIe per pair) = host_to_host_pairs_per_s",
        "single_pair_ms": 1e3 * single_pair_s, "single_pair_ms_min": 1e3 * min(lat), "single_pair_ms_samples": len(lat),
        "single_pair_latency_flag_ms": None if single_pair_latency_s is None else 1e3 * single_pair_latency_s,
        "stages_ms": stages,
        "output_checksum": int(out.astype(np.uint64).sum()),
        "latency_flag_output_identical": None if latency_checksum is None else latency_checksum == int(out.astype(np.uint64).sum()),
        "build_id": lib_build_id(), "build_id_basis": "sha256[:16] over csrc/*, include/nct.h, Makefile (the sources)",
        "build_id_so": lib_so_id(nct), "build_id_so_basis": "sha256[:16] of the libnct.so file this process loaded",
    }
    if wl == "pair700" and S == 700 and rank == 0:
        # the resident pair of slot 0 is the bench pair of tests/golden/pair700_oracle.json (seeds 1000 / 1001): its byte sum pins the timed code path to the oracle's result
        try:
            gold = json.load(open(os.path.join(REPO, "tests", "golden", "pair700_oracle.json")))["700"]
            res["parity_check"] = {"golden": "tests/golden/pair700_oracle.json[700] (CPU oracle, canonical order)", "golden_sum": gold["sum"], "golden_crc32": gold["crc32"],
                                   "output_crc32": zlib.crc32(out.tobytes()), "match": bool(gold["sum"] == res["output_checksum"] and gold["crc32"] == zlib.crc32(out.tobytes()))}
        except Exception as e:      # noqa: BLE001
            res["parity_check"] = {"match": None, "why": f"{type(e).__name__}: {e}"}
    res["vgg_mfma"] = vgg_mfma(src.shape[0], src.shape[1], ref.shape[0], ref.shape[1], prm.levels, stages["vgg_ms"])
    if rank == 0 and not args.no_roofline and not args.no_pmc and world == 1 and wl == "pair700" and S == 700:
        res["vgg_mfma"].update(vgg_mfma_util(local_rank))
    if rank == 0 and not args.no_roofline and prm.levels == 5:
        res["roofline"] = patchmatch_roofline(nct, ctx, prm, src.shape, ref.shape, local_rank, live_pmc=not args.no_pmc and wl == "pair700" and S == 700 and world == 1)
    if rank == 0 and not args.no_roofline and prm.levels == 5:
        res["roofline_color"] = color_roofline(nct, ctx, prm, src.shape)
    if rank == 0 and not args.no_roofline and not args.no_pmc and not args.no_pmc_1000 and world == 1 and wl == "pair700" and S == 700:
        res["roofline_1000"] = patchmatch_roofline_1000(local_rank)
    if rank == 0 and world == 1 and wl == "pair700" and not args.no_natural:
        res["natural"] = natural_extra(ctx, prm, synth)
    if rank == 0 and not args.no_cpu_baseline and world == 1:               # the CPU port is timed on rank 0 of the 1-GPU run only
        res["cpu_baseline"] = cpu_baseline(synth, ws, bs, src.shape[0], full=(args.cpu_baseline_full or (os.cpu_count() or 1) >= 32) and not args.cpu_baseline_sample)
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    os.close(real_stdout)


def pmc_traffic(device, live):
    """FETCH_SIZE / WRITE_SIZE of the finest-level PatchMatch launches of a real 700x700 pair, this build, two separate counter-only
    passes (MI355X_MICROARCH.md: FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2; FETCH_SIZE reports half the bytes of wide loads on
    gfx950 -> x2; the k_normalize dispatch of the same pass, a pure 125.44 MB streaming read, is kept as the calibration check).
    Returns (dict, how) or (None, why)."""
    exe = shutil.which("rocprofv3")
    rec = os.path.join(REPO, "profiles", "round5_pmc_patchmatch.json")
    bid = lib_build_id()
    why = "live PMC disabled"
    if live and exe:
        try:
            vals = {}
            for ctrs in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE")):
                with tempfile.TemporaryDirectory(dir="/tmp") as td:
                    env = dict(os.environ, TMPDIR="/tmp", HIP_VISIBLE_DEVICES=str(device))
                    subprocess.run([exe, "--pmc", *ctrs, "--kernel-trace", "-d", td, "-o", "c", "--output-format", "csv", "--",
                                    sys.executable, os.path.join(REPO, "scripts", "pair_only.py"), "700", "2"],
                                   cwd=REPO, env=env, check=True, capture_output=True, timeout=300)
                    for ctr in ctrs:
                        vals[ctr] = read_pmc_csv(td, ctr)
                        if ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                            COLOR_PMC[ctr] = read_color_pmc(td, ctr)
            out = finish_pmc(vals, bid, "live rocprofv3 passes inside bench.py")
            return out, "live"
        except Exception as e:      # noqa: BLE001 — fall back to the recorded passes
            why = f"live PMC failed: {type(e).__name__}: {str(e)[:120]}"
    elif live:
        why = "rocprofv3 not on PATH"
    if os.path.exists(rec):
        d = json.load(open(rec))
        if d.get("build_id") == bid:
            return d, "recorded (profiles/round5_pmc_patchmatch.json, same build id)"
        why += "; the recorded PMC passes belong to another build"
    return None, why


# the colour-solver kernels of roofline_color in the same counter passes (the pair of scripts/pair_only.py runs them all): name prefix in the trace per roofline_color key
COLOR_KERNELS = {"s1_apply": "void k_s1_apply<true>", "s1_update": "void k_s1_update<false>", "wls_down": "void (anonymous namespace)::k_mg_down<6, 32, 14, float, false, true>",
                 "wls_up": "void (anonymous namespace)::k_mg_up<6, 32, 16, float, false>", "wls_block_pre": "void (anonymous namespace)::k_mg_block<6, false>",
                 "wls_block_post": "void (anonymous namespace)::k_mg_block<6, true>", "wls_apply": "void (anonymous namespace)::k_cg_apply<6>", "wls_update": "void (anonymous namespace)::k_cg_update<6>"}
COLOR_PMC = {}


def read_color_pmc(td, ctr):
    """mean counter value per FULL-RESOLUTION, WORKING launch of each colour kernel: the largest grid of the name (k_s1_apply<true> also runs at 350x350) and, for the WLS kernels,
    without the launches enqueued past convergence (they exit on a flag and count next to nothing: below a quarter of the median)"""
    import csv, glob
    f = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == ctr]
    out = {}
    for key, pre in COLOR_KERNELS.items():
        rr = [r for r in rows if r["Kernel_Name"].startswith(pre)]
        if not rr:
            continue
        gmax = max(int(r["Grid_Size"]) for r in rr)
        v = sorted(float(r["Counter_Value"]) for r in rr if int(r["Grid_Size"]) == gmax)
        med = v[len(v) // 2]
        w = [x for x in v if x >= 0.25 * med] if med > 0 else v
        out[key] = {"launches": len(w), "mean": sum(w) / len(w)}
    return out


PM_FINEST = ("void k_pm_step<1, 1,", "void k_pm_prop<1, 1,")      # the finest level's 41 launches: init + 10 x (3 packed propagation launches + 1 propagation/random-search launch)


def read_pmc_csv(td, ctr, prefix=PM_FINEST):
    import csv, glob
    f = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == ctr]
    pm = sorted((r for r in rows if r["Kernel_Name"].startswith(prefix)), key=lambda r: int(r["Dispatch_Id"]))
    v = [float(r["Counter_Value"]) for r in pm]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in pm]          # us, this pass's own kernel trace
    norm = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"].startswith("k_normalize(")]
    return {"dispatches": len(v), "mean": sum(v) / len(v), "normalize_max": max(norm) if norm else None, "per_dispatch": v, "per_dispatch_us": dur}


def finish_pmc(vals, bid, how):
    fetch = vals["FETCH_SIZE"]["mean"] * 1024 * 2          # KB -> B, x2: gfx950 counts 128-B requests as 64 B (guide, HBM section)
    write = vals["WRITE_SIZE"]["mean"] * 1024
    cal = {}
    if vals["FETCH_SIZE"]["normalize_max"]:
        cal = {"k_normalize_700x700x64_bytes": 125440000, "FETCH_SIZE_KB_raw": vals["FETCH_SIZE"]["normalize_max"], "WRITE_SIZE_KB_raw": vals["WRITE_SIZE"]["normalize_max"],
               "fetch_raw_over_actual": vals["FETCH_SIZE"]["normalize_max"] * 1024 / 125440000.0,
               "write_raw_over_actual": (vals["WRITE_SIZE"]["normalize_max"] or 0) * 1024 / 125440000.0}
    # L1 side: vector-memory read instructions x 64 lanes x 16 bytes (the candidate-tile loads; the ~6 % of 4-byte NNF loads make this an upper
    # bound) against what 256 L1s return in the launch's own cycles (GRBM_GUI_ACTIVE counts per XCD: / 8) at 64 bytes per clock and CU
    l1 = None
    if "SQ_INSTS_VMEM_RD" in vals:
        cyc = vals["GRBM_GUI_ACTIVE"]["mean"] / 8.0
        l1 = {"vmem_read_instructions_per_launch": vals["SQ_INSTS_VMEM_RD"]["mean"], "bytes_per_launch_upper_bound": vals["SQ_INSTS_VMEM_RD"]["mean"] * 1024.0,
              "launch_cycles": cyc, "frac_of_l1_return_bandwidth": vals["SQ_INSTS_VMEM_RD"]["mean"] * 1024.0 / (cyc * 256 * 64)}
    # The 41 launches of a level are not alike: step 0 evaluates the initial field, steps with jump 8 / 4 / 2 only propagate (mostly L1/L2 hits: neighbouring
    # queries propose overlapping tiles), every fourth step (jump 1) adds the random search, whose candidates lie up to +-32 px away and miss L2 — those launches
    # are the bandwidth-bound ones. Per class: fabric-side bytes (FETCH_SIZE x 2; writes are < 1 %) over the launch durations of the same counter pass.
    by_step = None
    fs = vals["FETCH_SIZE"]
    if fs.get("per_dispatch") and len(fs["per_dispatch"]) % 41 == 0:
        cls = {"init": [], "propagation": [], "random_search": []}
        for i, (kb, us) in enumerate(zip(fs["per_dispatch"], fs["per_dispatch_us"])):
            k = i % 41
            cls["init" if k == 0 else ("random_search" if k % 4 == 0 else "propagation")].append((kb * 2048.0, us))
        by_step = {name: {"launches_per_level": len(v) // (len(fs["per_dispatch"]) // 41), "avg_launch_us": sum(u for _, u in v) / len(v), "fabric_bytes_per_launch": sum(b for b, _ in v) / len(v),
                          "fabric_GBs": sum(b for b, _ in v) / sum(u for _, u in v) / 1e3, "frac_of_hbm_peak": sum(b for b, _ in v) / sum(u for _, u in v) / 1e3 / HBM_PEAK_GBS}
                   for name, v in cls.items() if v}
    return {"build_id": bid, "how": how, "kernel": "k_pm_step<1, 1, 2, 2, 8> + k_pm_prop<1, 1, 2, 2, 8> (the finest level's 41 launches)", "dispatches": vals["FETCH_SIZE"]["dispatches"], "by_step": by_step,
            "FETCH_SIZE_KB_per_dispatch_raw": vals["FETCH_SIZE"]["mean"], "WRITE_SIZE_KB_per_dispatch_raw": vals["WRITE_SIZE"]["mean"],
            "calibration": cal, "corrected_bytes_per_launch": {"fetch": fetch, "write": write, "total": fetch + write}, "l1": l1}


def patchmatch_roofline(nct, ctx, prm, sshape, rshape, device, live_pmc):
    """Finest pyramid level of the resident pair: 41 launches of k_pm_step<1, 1, 2, 2, 8> (init + 10 iterations x 4 jumps, S->R and R->S fields in
    the same launch). Time: HIP events on the library's stream around exactly those launches (nct_pair_timing.pm_level_ms), best of 3
    pairs; evaluations: device counter in a separate pair (the counting atomics would perturb the timing)."""
    tms = [ctx.pair_run(prm, want_timing=True) for _ in range(3)]
    ms = min(t["pm_level_ms"][4] for t in tms)
    n_launch = tms[0]["pm_level_launches"][4]
    p2 = nct.Params.default()
    for k, _ in nct.Params._fields_:
        setattr(p2, k, getattr(prm, k))
    p2.flags |= nct.FLAG_COUNT_EVALS
    evals = ctx.pair_run(p2, want_timing=True)["pm_level_evals"][4]
    C = 64
    nq = sshape[0] * sshape[1] + rshape[0] * rshape[1]
    alg = pm_bytes(evals, nq, n_launch, C)
    launch_s = ms * 1e-3 / n_launch
    alg_gbs = alg / n_launch / launch_s / 1e9
    traffic, how, pmc = None, "PMC passes exist for the 700x700 pair only", None
    if tuple(sshape[:2]) == (700, 700) and tuple(rshape[:2]) == (700, 700):
        pmc, how = pmc_traffic(device, live_pmc)
        if pmc is not None:
            traffic = pmc["corrected_bytes_per_launch"]["total"]
    footprint = nq * C * 4 + nq * 16
    achieved = (traffic if traffic is not None else alg / n_launch) / launch_s / 1e9
    return {"bound": "hbm", "kernel": f"k_pm_step<1, 1, 2, 2, 8> (init + the 10 propagation/random-search launches) and k_pm_prop<1, 1, 2, 2, 8> (the 30 packed propagation launches) — C=64, 8x8 queries per workgroup, "
                                      f"8 lanes per query, {sshape[1]}x{sshape[0]} <-> {rshape[1]}x{rshape[0]}, both directions per launch, pipeline features",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "basis": "fabric incl. Infinity Cache: PMC fabric-side bytes (FETCH_SIZE x2 + WRITE_SIZE, L2 misses; MALL hits are counted) per launch / event-timed launch" if traffic is not None else
                     "ALGORITHMIC bytes (no PMC pass of this build available) — not an HBM fraction, can exceed 1",
            # HBM share of frac: a launch sweeps its whole footprint, so at most MALL / footprint of the fabric-side bytes can be Infinity-Cache hits
            "hbm_frac_bounds": None if traffic is None else [achieved / HBM_PEAK_GBS * max(0.0, 1.0 - MALL_BYTES / footprint), achieved / HBM_PEAK_GBS],
            "traffic_source": how, "pmc": pmc, "launches": n_launch, "avg_launch_ms": 1e3 * launch_s, "evals": evals,
            "algorithmic_bytes_per_launch": alg / n_launch, "algorithmic_GBs": alg_gbs,
            "traffic_over_algorithmic": None if traffic is None else traffic / (alg / n_launch),
            # what the launch must touch at least once: both feature maps (read as query regions and as candidate tiles) + both NNF / distance fields in and out
            "footprint_bytes": footprint, "restream_factor": None if traffic is None else traffic / footprint,
            "l1_frac": None if not (pmc and pmc.get("l1")) else pmc["l1"]["frac_of_l1_return_bandwidth"],
            # frac averages 41 unlike launches; the random-search launches (every fourth: jump 1) are the bandwidth-bound ones
            "frac_random_search_launches": None if not (pmc and pmc.get("by_step")) else pmc["by_step"]["random_search"]["frac_of_hbm_peak"],
            "note": "traffic = L2-miss (fabric-side) bytes: FETCH_SIZE counts Infinity-Cache (MALL) hits as well, and the level's 251 MB footprint fits the 256 MiB MALL, so the "
                    "true HBM demand is <= frac; restream_factor = traffic / footprint. Overlapping candidate tiles are served by L1/L2, so the no-reuse byte model "
                    "(algorithmic_GBs) exceeds the HBM peak; DESIGN.md 3.2"}


def patchmatch_roofline_1000(device):
    """BASELINE config 4 (one 1000x1000 pair): the finest PatchMatch level's footprint (two 256 MB feature maps + fields = 528 MB) exceeds the 256 MiB Infinity
    Cache, so the fabric-side bytes of these launches ARE (at least half) HBM bytes — the 700x700 level's 267 MB can be MALL hits. Two counter-only passes
    (FETCH_SIZE, WRITE_SIZE) over scripts/pair_only.py 1000 2; launch durations from the FETCH_SIZE pass's own kernel trace."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {"frac": None, "why": "rocprofv3 not on PATH"}
    try:
        vals = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                env = dict(os.environ, TMPDIR="/tmp", HIP_VISIBLE_DEVICES=str(device))
                subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "-d", td, "-o", "c", "--output-format", "csv", "--",
                                sys.executable, os.path.join(REPO, "scripts", "pair_only.py"), "1000", "2"], cwd=REPO, env=env, check=True, capture_output=True, timeout=400)
                vals[ctr] = read_pmc_csv(td, ctr)
        fs = vals["FETCH_SIZE"]
        fetch, write = fs["mean"] * 2048.0, vals["WRITE_SIZE"]["mean"] * 1024.0
        us = sum(fs["per_dispatch_us"]) / len(fs["per_dispatch_us"])
        cls = {"init": [], "propagation": [], "random_search": []}
        if len(fs["per_dispatch"]) % 41 == 0:
            for i, (kb, u) in enumerate(zip(fs["per_dispatch"], fs["per_dispatch_us"])):
                k = i % 41
                cls["init" if k == 0 else ("random_search" if k % 4 == 0 else "propagation")].append((kb * 2048.0, u))
        by = {n: {"avg_launch_us": sum(u for _, u in v) / len(v), "fabric_bytes_per_launch": sum(b for b, _ in v) / len(v), "frac_of_hbm_peak": sum(b for b, _ in v) / sum(u for _, u in v) / 1e3 / HBM_PEAK_GBS}
              for n, v in cls.items() if v}
        nq = 2 * 1000 * 1000
        foot = nq * 64 * 4 + nq * 16
        gbs = (fetch + write) / us / 1e3
        return {"bound": "hbm", "kernel": "k_pm_step<1, 1, 2, 2, 8> + k_pm_prop<1, 1, 2, 2, 8> at 1000x1000 <-> 1000x1000 (BASELINE config 4), counter-pass launch durations", "dispatches": fs["dispatches"],
                "traffic": fetch + write, "avg_launch_us": us, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "by_step": by,
                "footprint_bytes": foot, "footprint_over_mall": foot / (256.0 * 1024 * 1024), "restream_factor": (fetch + write) / foot,
                "basis": "fabric incl. Infinity Cache (footprint 2x the MALL: at least half of it is HBM)", "hbm_frac_bounds": [gbs / HBM_PEAK_GBS * max(0.0, 1.0 - MALL_BYTES / foot), gbs / HBM_PEAK_GBS],
                "fetch_size_factor": 2.0, "fetch_size_factor_basis": "profiles/round4_fetch_calibration.md: FETCH_SIZE x 1024 / known bytes = 0.50 on a streaming read AND on 8-lane x 16-B tile rows at random pixels",
                "note": "footprint 2.0x the Infinity Cache: at most half of these bytes can be MALL hits, so HBM demand >= frac / 2 and <= frac; the 700x700 figure (roofline.frac) is fabric-side incl. MALL"}
    except Exception as e:      # noqa: BLE001
        return {"frac": None, "why": f"PMC pass failed: {type(e).__name__}: {str(e)[:160]}"}


def color_roofline(nct, ctx, prm, sshape):
    """Event-timed single launches of the full-resolution colour-solver kernels (NCT_FLAG_TIME_KERNELS: 8 iterations of the finest S1 level, 4 iterations of each
    of the five WLS solves) against their COMPULSORY bytes per launch: what the kernel must read and write once, per pixel of the N = H*W grid, stated here."""
    p2 = nct.Params.default()
    for k, _ in nct.Params._fields_:
        setattr(p2, k, getattr(prm, k))
    p2.flags |= nct.FLAG_TIME_KERNELS
    ctx.pair_run(p2, want_timing=True)
    tm = ctx.pair_run(p2, want_timing=True)
    us = dict(zip(nct.KT_NAMES, tm["kernel_us"])); ns = dict(zip(nct.KT_NAMES, tm["kernel_samples"]))
    N = sshape[0] * sshape[1]
    # bytes per pixel, reads + writes. S1 (3 channels x (a, b) = 6 fp64 unknowns per pixel, kNN k = 8): operator = p of the pixel (48) + 8 out-neighbour records (8 x 48) + in-edge
    # sources/weights/records (~8 x (4 + 8 + 48)) + 4 raster neighbours (L2 hits, not counted) + coefficients daa/dab/dbb (72) + gx, gy (16) + ids (32) + iw2 (64) + Ap out (48);
    # WLS (6 right-hand sides): see DESIGN.md 3.4
    model = {
        "s1_apply": (48 + 8 * 48 + 8 * 60 + 72 + 16 + 32 + 64 + 48, "r + 8 out-neighbour records + 8 in-edge (src, w, record) + coefficients + w = Op(r)"),
        "s1_update": (48 * 5 + 48 * 4, "r, w, p, s, x in; p, s, x, r out (fp64 x 6 each: the single-reduction recurrence's one vector pass)"),
        "wls_block_pre": (24 + 16 + 8 + 24, "r rounded to fp32 (x 6), Thomas factors of the x / y lines (4 floats), couplings in; the block step's iterate (fp32 x 6) out"),
        "wls_down": (24 + 24 + 16 + 24 + 6 + 9, "r (fp32 x 6), the block step's iterate + fp32 coefficients in; x (fp32 x 6) + coarse rhs + P columns"),
        "wls_up": (24 + 24 + 16 + 16 + 6 + 24, "r (fp32), x, coefficients, P weights, coarse correction in; the leg's iterate out"),
        "wls_block_post": (24 + 24 + 16 + 12 + 24, "r (fp32), the up leg's iterate, Thomas factors, diagonal + couplings in; z out"),
        "wls_apply": (24 + 48 + 24 + 48, "z (fp32 x 6), r, fp64 coefficients in; w out"),
        "wls_update": (48 * 4 + 24 + 48 + 48 * 4 + 24, "p, s, x, r, z, w in; p, s, x, r and the fp32 copy of r out"),
    }
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "pixels": N, "kernels": {},
           "traffic_note": "traffic = fabric-side bytes per full-resolution launch from the live FETCH_SIZE (x2, profiles/round4_fetch_calibration.md) and WRITE_SIZE passes of the roofline object "
                           "(same pair, scripts/pair_only.py), Infinity-Cache hits included; null without those passes; traffic_frac = traffic / event-timed launch / 8 TB/s"}
    for k, (bpp, what) in model.items():
        if ns.get(k):
            gbs = bpp * N / us[k] / 1e3
            e = {"avg_launch_us": us[k], "samples": ns[k], "bytes_per_pixel": bpp, "bytes_per_launch": bpp * N, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "bytes": what, "traffic": None}
            fs, wsz = COLOR_PMC.get("FETCH_SIZE", {}).get(k), COLOR_PMC.get("WRITE_SIZE", {}).get(k)
            if fs and wsz and N == 700 * 700:
                e["traffic"] = fs["mean"] * 2048.0 + wsz["mean"] * 1024.0
                e["traffic_over_compulsory"] = e["traffic"] / (bpp * N)
                e["traffic_frac"] = e["traffic"] / us[k] / 1e3 / HBM_PEAK_GBS
                e["traffic_launches"] = fs["launches"]
            out["kernels"][k] = e
    if ns.get("s1_scalars"):
        out["kernels"]["s1_scalars"] = {"avg_us": us["s1_scalars"], "samples": ns["s1_scalars"], "what": "one-workgroup reduction of the operator pass's 1915 x 6 block partials + CG scalars (latency bound)"}
    if ns.get("wls_coarse"):
        out["kernels"]["wls_coarse"] = {"avg_us": us["wls_coarse"], "samples": ns["wls_coarse"], "what": "everything below the finest level of one V-cycle (latency bound: 7 launches)"}
    it_s1 = sum(us[k] for k in ("s1_apply", "s1_scalars", "s1_update") if ns.get(k))
    it_wls = sum(us[k] for k in ("wls_block_pre", "wls_down", "wls_up", "wls_block_post", "wls_apply", "wls_update", "wls_coarse") if ns.get(k))
    # WLS per PCG iteration ~ (5 nnz + 6 vectors) x 8 B x n per right-hand side; nonlocal CG per iteration 2 passes over ~25 n rows x (2 idx + 2 val) + 4 vectors of 2 n fp64, per channel
    if it_wls:
        b = (5 + 6) * 8 * N * 6
        out["wls_iteration"] = {"us": it_wls, "survey_8d_bytes": b, "survey_8d_GBs": b / it_wls / 1e3, "survey_8d_frac": b / it_wls / 1e3 / HBM_PEAK_GBS,
                                "compulsory_bytes": sum(model[k][0] for k in ("wls_block_pre", "wls_down", "wls_up", "wls_block_post", "wls_apply", "wls_update")) * N}
    if it_s1:
        b = 3 * (2 * 25 * N * (2 * 4 + 2 * 8) + 4 * 2 * N * 8)
        out["s1_iteration"] = {"us": it_s1, "survey_8d_bytes": b, "survey_8d_GBs": b / it_s1 / 1e3, "survey_8d_frac": b / it_s1 / 1e3 / HBM_PEAK_GBS,
                               "compulsory_bytes": sum(model[k][0] for k in ("s1_apply", "s1_update")) * N,
                               "note": "the survey's model is the reference's explicit CSR A^T A product; the matrix-free operator moves ~3x fewer bytes, so its survey fraction can exceed 1"}
    out["dominant"] = max(out["kernels"].items(), key=lambda kv: kv[1].get("avg_launch_us", 0))[0] if out["kernels"] else None
    return out


def vgg_mfma(sh, sw, rh, rw, levels, vgg_ms):
    """VGG19 conv work of one pair (S and R forwards to conv5_1 + one re-predict per further level to conv4_1, 3_1, 2_1, 1_1: SURVEY 8a V2)
    over the instrumented pair's VGG stage time (which also contains preprocess, pools and layout transposes) vs the f32-MFMA peak."""
    cin = [3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512]
    cout = [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512]
    pool_after = {1, 3, 7, 11}
    tap_conv = [0, 2, 4, 8, 12]

    def cum_flops(h, w):
        cum, flops = [], 0
        for i in range(13):
            flops += 2 * 9 * cin[i] * cout[i] * h * w
            cum.append(flops)
            if i in pool_after:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return cum
    cs, cr = cum_flops(sh, sw), cum_flops(rh, rw)
    per_pair = cs[tap_conv[4]] + cr[tap_conv[4]] + sum(cs[tap_conv[t]] for t in range(4 - (levels - 1), 4))
    tf = per_pair / (vgg_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "k_conv3x3_mfma2b (v_mfma_f32_32x32x1_2b_f32), all forwards of a pair", "flops_per_pair": per_pair,
            "stage_ms": vgg_ms, "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3}


def vgg_mfma_util(device):
    """SQ_VALU_MFMA_BUSY_CYCLES of the conv kernel as shipped: two counter-only rocprofv3 passes over scripts/vgg_only.py (three 700x700 forwards to conv5_1);
    utilisation = MFMA-busy cycles / (launch cycles x 1024 SIMDs), launch cycles = GRBM_GUI_ACTIVE / 8 XCDs, summed over all conv launches (FLOP-weighted by
    construction) and for the conv4_x launches alone (the 244-workgroup layers, one round on 256 CUs)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return {"mfma_util": None, "mfma_util_source": "rocprofv3 not on PATH"}
    try:
        import csv, glob
        tot = {}
        for ctrs in (("SQ_VALU_MFMA_BUSY_CYCLES",), ("GRBM_GUI_ACTIVE",)):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                env = dict(os.environ, TMPDIR="/tmp", HIP_VISIBLE_DEVICES=str(device))
                subprocess.run([exe, "--pmc", *ctrs, "--kernel-trace", "-d", td, "-o", "c", "--output-format", "csv", "--", sys.executable, os.path.join(REPO, "scripts", "vgg_only.py")],
                               cwd=REPO, env=env, check=True, capture_output=True, timeout=300)
                f = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
                for r in csv.DictReader(open(f[0])):
                    if r["Kernel_Name"].startswith("void k_conv3x3_mfma") and r["Counter_Name"] == ctrs[0]:
                        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
                        tot.setdefault(key, {}).setdefault(ctrs[0], []).append(float(r["Counter_Value"]))
        busy = sum(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) for v in tot.values())
        cyc = sum(sum(v["GRBM_GUI_ACTIVE"]) for v in tot.values()) / 8.0
        per = {f"{k[0].replace('void ', '')} grid {k[1]}": round(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(v["GRBM_GUI_ACTIVE"]) / 8.0 * 1024), 4) for k, v in sorted(tot.items())}
        return {"mfma_util": busy / (cyc * 1024), "mfma_util_per_layer_group": per, "mfma_util_source": "live rocprofv3 passes inside bench.py (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs))"}
    except Exception as e:      # noqa: BLE001
        return {"mfma_util": None, "mfma_util_source": f"PMC pass failed: {type(e).__name__}: {str(e)[:120]}"}


def cpu_baseline(synth, ws, bs, S, full=False):
    """Oracle ('port') end-to-end, measured on this box's host cores on a bounded sample of the same workload: one 350x350 pair (full
    L=5->1 loop, same synthetic VGG19; a quarter of the 700x700 pair's pixels) on all cores and one 64x64 pair on a single thread,
    scaled to the SxS pair by pixel count. The law was checked against full-size runs (profiles/README.md): 670-830 s per megapixel
    from 176^2 to 700^2 on 8 cores; on the 64-thread GPU box the 700x700 pair itself takes 138 s (0.0072 pairs/s) — small samples
    parallelise worse there, so the scaled figure UNDERSTATES the CPU a little. `--cpu-baseline-full` times the real SxS pair (minutes)."""
    import oracle_bind
    orc = oracle_bind.load()
    threads = min(64, os.cpu_count() or 1)

    def run(n, th):
        orc.l.orc_set_threads(th)
        src, ref = synth.image(1000, n, n), synth.image(1001, n, n)
        t0 = time.perf_counter()
        orc.process_pair(src, ref, ws, bs)
        return time.perf_counter() - t0
    n = min(350, S)
    dt = run(n, threads)
    n1 = min(64, S)
    dt1 = run(n1, 1)
    scale = (n * n) / float(S * S)
    scale1 = (n1 * n1) / float(S * S)
    res = {"value": scale / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
           "sample": f"oracle orc_process_pair on one {n}x{n} pair (full L=5->1 loop, same synthetic VGG19): {dt:.2f} s on {threads} OpenMP threads, scaled to "
                     f"{S}x{S} by pixel count ({1 / scale:.2f}x; full-size check: profiles/README.md); one {n1}x{n1} pair on 1 thread: {dt1:.2f} s",
           "value_1thread": scale1 / dt1, "sample_seconds": dt, "sample_seconds_1thread": dt1}
    if full:
        # the real SxS pair, once: the reported value (the scaled sample stays beside it as a cross-check of the scaling law)
        dtf = run(S, threads)
        res["value_scaled_sample"] = res["value"]
        res["value"] = 1.0 / dtf
        res["full_pair_seconds"] = dtf
        res["sample"] = (f"oracle orc_process_pair on the full {S}x{S} bench pair (L=5->1, same synthetic VGG19): {dtf:.1f} s on {threads} OpenMP threads; cross-checks: one {n}x{n} pair "
                         f"{dt:.2f} s (scaled by pixel count: {1 / res['value_scaled_sample']:.1f} s), one {n1}x{n1} pair on 1 thread {dt1:.2f} s")
    return res


if __name__ == "__main__":
    main()
