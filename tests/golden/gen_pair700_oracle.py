#!/usr/bin/env python3
"""Generates tests/golden/pair700_oracle.json: CRC-32 and byte sum of the CPU oracle's result (oracle/orc_process_pair, full
L=5->1 loop, synthetic VGG19 seed 19) for the 700x700 bench pair (synth.image seeds 1000 / 1001), a 1000x1000 pair (BASELINE config 4) and a mixed-size pair.
Takes ~7 / ~15 / ~3 minutes on 8 cores.
The oracle is a restatement of the reference algorithm (see oracle/README.md); this fixture pins the GPU path to it at the
full BASELINE size, where running the oracle inside the test suite would be too slow."""
import json, os, sys, time, zlib
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, oracle_bind, synth
from caffemodel_io import synthetic_vgg19
orc = oracle_bind.load()
orc.l.orc_set_threads(min(32, os.cpu_count() or 1))
ws, bs = synthetic_vgg19(19)
out = {}
CASES = [("700", 700, 700, 700, 700), ("1000", 1000, 1000, 1000, 1000), ("mixed", 333, 517, 612, 401)]
only = sys.argv[1:]                                   # optional: names to (re)generate; others are kept from the existing file
path = os.path.join(HERE, "pair700_oracle.json")
if os.path.exists(path):
    out = json.load(open(path))
for (name, sh, sw, rh, rw) in CASES:
    if only and name not in only:
        continue
    src, ref = synth.image(1000, sh, sw), synth.image(1001, rh, rw)
    t = time.time(); exp = orc.process_pair(src, ref, ws, bs); dt = time.time() - t
    out[name] = {"src_seed": 1000, "ref_seed": 1001, "shape": [sh, sw, rh, rw], "crc32": zlib.crc32(exp.tobytes()),
                 "sum": int(exp.astype(np.uint64).sum()), "oracle_seconds": round(dt, 1)}
    print(name, out[name], flush=True)
    json.dump(out, open(path, "w"), indent=1)
