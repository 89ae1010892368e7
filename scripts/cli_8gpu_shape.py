"""The CLI's 8-GPU host shape under load, proven on ONE GPU: `-gpus 8 -inflight 4` with NCT_DEVICE_OVERRIDE=0 = one process, 32 worker contexts (each with its
arena, streams and host thread) + the I/O pool, all on device 0 — the HIP runtime sees the launch rate and the thread count of a full node. If the runtime's locks or the WLS poll threads
collapsed under that, total pairs/s would fall below the saturated 1-GPU rate of `-gpus 1 -inflight 4`.
Writes N 700x700 pairs as PNGs, runs both shapes, reports pairs/s (PNG decode + encode included), host CPU seconds per wall second and kernel launches per second
(4.2 k launches per pair, profiles/round4_e2e_kernels.md).      usage: python scripts/cli_8gpu_shape.py [npairs=64] [size=700]"""
import json, os, re, resource, subprocess, sys, tempfile, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import synth
from PIL import Image
from caffemodel_io import synthetic_vgg19, write_caffemodel, write_deploy_prototxt
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 700
LAUNCHES_PER_PAIR = 4200
td = tempfile.mkdtemp()
os.makedirs(os.path.join(td, "model", "vgg19")); os.makedirs(os.path.join(td, "in"))
ws, bs = synthetic_vgg19(19)
write_caffemodel(os.path.join(td, "model", "vgg19", "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
write_deploy_prototxt(os.path.join(td, "model", "vgg19", "VGG_ILSVRC_19_layers_deploy.prototxt"))
lines = []
for i in range(npairs):
    Image.fromarray(synth.image(1000 + 2 * i, S, S)[..., ::-1].copy()).save(os.path.join(td, "in", f"s{i}.png"))
    Image.fromarray(synth.image(1001 + 2 * i, S, S)[..., ::-1].copy()).save(os.path.join(td, "in", f"r{i}.png"))
    lines.append(f"s{i}.png r{i}.png 2.0\n")
open(os.path.join(td, "in", "pairs.txt"), "w").writelines(lines)
exe = os.path.join("neural-color-transfer_amd", "bin", "neural_color_transfer")
res = {"npairs": npairs, "size": S, "host_threads": os.cpu_count()}
ref = None
# round 6: the process-per-GPU shape (-procs 8: eight processes, each its own HIP runtime, contexts, I/O pool and status.<r>.jsonl; here all on device 0) next to the one-process shape.
# Its rate is pairs / the parent's "All 8 process(es) finished in X sec" (process start, HIP initialisation and the model parse of every child included — the one-process
# shapes' own "pairs/sec" line starts the clock after the model is loaded, so `wall_pairs_per_s` = pairs / process wall is printed for every shape as the comparable figure).
for tag, gpus, K, procs in (("gpus1_inflight4", 1, 4, 0), ("gpus8_inflight4_on_one_device", 8, 4, 0), ("gpus8_inflight1_on_one_device", 8, 1, 0),
                            ("procs8_inflight4_on_one_device", 8, 4, 8), ("procs8_inflight1_on_one_device", 8, 1, 8)):
    out = os.path.join(td, "out_" + tag)
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN); t0 = time.time()
    shape = ["-procs", str(procs), "-steal", "1"] if procs else ["-gpus", str(gpus)]
    r = subprocess.run([exe, "-m", os.path.join(td, "model"), "-i", os.path.join(td, "in"), "-o", out] + shape + ["-inflight", str(K)],
                       capture_output=True, text=True, env=dict(os.environ, NCT_DEVICE_OVERRIDE="0"))
    wall = time.time() - t0; r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    last = r.stdout.strip().splitlines()[-1]
    if procs:
        pps = npairs / float(re.search(r"finished in ([0-9.]+) sec", last).group(1))
    else:
        pps = float(re.search(r"\(([0-9.]+) pairs/sec\)", last).group(1))
    files = sorted(n for n in os.listdir(out) if n.endswith(".png"))
    assert len(files) == npairs, (len(files), npairs)
    blobs = {n: open(os.path.join(out, n), "rb").read() for n in files}
    if ref is None:
        ref = blobs
    same = all(blobs[n] == ref[n] for n in files)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    res[tag] = {"contexts": gpus * K, "processes": procs or 1, "cli_pairs_per_s": pps, "wall_pairs_per_s": round(npairs / wall, 2), "process_wall_s": round(wall, 2), "host_cpu_seconds": round(cpu, 1), "host_cpus_busy": round(cpu / wall, 2),
                "kernel_launches_per_s": round(pps * LAUNCHES_PER_PAIR), "outputs_identical_to_gpus1": same, "last_line": last}
    print(tag, json.dumps(res[tag]), flush=True)
res["ratio_8x4_over_1x4"] = res["gpus8_inflight4_on_one_device"]["cli_pairs_per_s"] / res["gpus1_inflight4"]["cli_pairs_per_s"]
res["ratio_procs8x4_over_1x4_wall"] = res["procs8_inflight4_on_one_device"]["wall_pairs_per_s"] / res["gpus1_inflight4"]["wall_pairs_per_s"]
print(json.dumps(res))
