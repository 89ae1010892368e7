"""BASELINE config 5 on one GPU through the C++ CLI: the first N pairs of bench.py's `mixed256` workload (sides ~U{256..1000} from seed 5000+i, images from seeds
1000+2i / 1001+2i) written as PNGs, run with `-inflight K -io T`, every output checked for existence; prints the CLI's own pairs/s line (PNG decode + encode
included) so it can be set against `bench.py --workload mixed256 --batch N --inflight K` (host buffer in -> host buffer out, no files).
usage: mixed_batch_cli.py [npairs] [inflight] [io threads]"""
import os, subprocess, sys, tempfile, json, re
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, synth
from PIL import Image
from caffemodel_io import synthetic_vgg19, write_caffemodel, write_deploy_prototxt
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
io = sys.argv[3] if len(sys.argv) > 3 else "-1"
td = tempfile.mkdtemp()
os.makedirs(os.path.join(td, "model", "vgg19")); os.makedirs(os.path.join(td, "in"))
ws, bs = synthetic_vgg19(19)
write_caffemodel(os.path.join(td, "model", "vgg19", "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
write_deploy_prototxt(os.path.join(td, "model", "vgg19", "VGG_ILSVRC_19_layers_deploy.prototxt"))
lines = []
for i in range(npairs):
    sh, sw, rh, rw = (int(v) for v in np.random.default_rng(5000 + i).integers(256, 1001, 4))
    Image.fromarray(synth.image(1000 + 2 * i, sh, sw)[..., ::-1].copy()).save(os.path.join(td, "in", f"s{i}.png"))
    Image.fromarray(synth.image(1001 + 2 * i, rh, rw)[..., ::-1].copy()).save(os.path.join(td, "in", f"r{i}.png"))
    lines.append(f"s{i}.png r{i}.png 2.0\n")
open(os.path.join(td, "in", "pairs.txt"), "w").writelines(lines)
exe = os.path.join("neural-color-transfer_amd", "bin", "neural_color_transfer")
res = {}
for tag, ioarg in (("pool", io), ("io0", "0")):
    out = os.path.join(td, "out_" + tag)
    r = subprocess.run([exe, "-m", os.path.join(td, "model"), "-i", os.path.join(td, "in"), "-o", out, "-g", "0", "-inflight", str(K), "-io", ioarg],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    last = r.stdout.strip().splitlines()[-1]
    print(tag, last)
    res[tag] = float(re.search(r"\(([0-9.]+) pairs/sec\)", last).group(1))
    outs = [n for n in os.listdir(out) if n.endswith(".png")]          # + status.jsonl
    assert len(outs) == npairs, (len(outs), npairs)
print(json.dumps({"npairs": npairs, "inflight": K, "cli_pairs_per_s_io_pool": res["pool"], "cli_pairs_per_s_io0": res["io0"]}))
