"""BASELINE config 5 flavour on one GPU: a batch of mixed-size (256-1000 px) pairs through the C++ CLI with -inflight K.
Checks that every output exists and reports the CLI's own pairs/s line. usage: mixed_batch_cli.py [npairs] [inflight]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, synth
from PIL import Image
from caffemodel_io import synthetic_vgg19, write_caffemodel
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
td = tempfile.mkdtemp()
os.makedirs(os.path.join(td, "model", "vgg19")); os.makedirs(os.path.join(td, "in"))
ws, bs = synthetic_vgg19(19)
write_caffemodel(os.path.join(td, "model", "vgg19", "VGG_ILSVRC_19_layers.caffemodel"), ws, bs)
rng = np.random.default_rng(7)
lines = []
for i in range(npairs):
    sh, sw, rh, rw = (int(v) for v in rng.integers(256, 1001, 4))
    Image.fromarray(synth.image(100 + i, sh, sw)[..., ::-1].copy()).save(os.path.join(td, "in", f"s{i}.png"))
    Image.fromarray(synth.image(200 + i, rh, rw)[..., ::-1].copy()).save(os.path.join(td, "in", f"r{i}.png"))
    lines.append(f"s{i}.png r{i}.png 2.0\n")
open(os.path.join(td, "in", "pairs.txt"), "w").writelines(lines)
exe = os.path.join("neural-color-transfer_amd", "bin", "neural_color_transfer")
r = subprocess.run([exe, "-m", os.path.join(td, "model"), "-i", os.path.join(td, "in"), "-o", os.path.join(td, "out"), "-g", "0", "-inflight", str(K)],
                   capture_output=True, text=True)
print(r.stdout.strip().splitlines()[-1])
assert r.returncode == 0, r.stderr
outs = [n for n in os.listdir(os.path.join(td, "out")) if n.endswith(".png")]          # + status.jsonl
assert len(outs) == npairs, (len(outs), npairs)
print("ok:", len(outs), "outputs")
