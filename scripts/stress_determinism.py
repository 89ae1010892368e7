"""Race hunt for the round-4 / round-5 kernels (round 5: hub passes of S1 and the votes, side-stream graph builds, per-run kNN searches — two of the four contexts run the
reference's demo photographs)
Round 4: (k_mg_down / k_mg_up / k_mg_mid with LDS-parked coefficients, k_pm_prop's per-wave LDS lists): the same pairs through nct_pair_run over and over,
alone and with several contexts in flight on the GPU; every result must have the CRC of the first. usage: python scripts/stress_determinism.py [rounds=12]"""
import sys, threading, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ws, bs = synthetic_vgg19(19)
shapes = [(700, 700, 700, 700), (333, 517, 612, 401), (1000, 1000, 1000, 1000), (96, 80, 72, 104)]
ctxs = [nct.Context(0) for _ in shapes]
ctxs[0].vgg19_load_raw(ws, bs)
for c in ctxs[1:]:
    c.vgg19_share_weights(ctxs[0])
import os, numpy as np
from PIL import Image
nat = lambda n: np.ascontiguousarray(np.asarray(Image.open(os.path.join("tests", "golden", "natural", n + ".png")).convert("RGB"))[..., ::-1])
for k, (c, (sh, sw, rh, rw)) in enumerate(zip(ctxs, shapes)):
    if k == 1: c.pair_upload(nat("in4"), nat("tar4"))          # natural photographs: hubs, long vote lists, isolated colours
    elif k == 3: c.pair_upload(nat("in1"), nat("tar1"))
    else: c.pair_upload(synth.image(1000, sh, sw), synth.image(1001, rh, rw))
prm = nct.Params.default()
ref = []
for c in ctxs:
    c.pair_run(prm); ref.append(zlib.crc32(c.pair_download().tobytes()))
print("reference CRCs", ref, flush=True)
bad = 0
for r in range(rounds):
    res = [None] * len(ctxs)
    def work(k):
        ctxs[k].pair_run(prm); res[k] = zlib.crc32(ctxs[k].pair_download().tobytes())
    if r % 2 == 0:                       # concurrently: four contexts in flight
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(ctxs))]
        [t.start() for t in th]; [t.join() for t in th]
    else:
        for k in range(len(ctxs)): work(k)
    ok = res == ref
    bad += 0 if ok else 1
    print("round", r, "concurrent" if r % 2 == 0 else "sequential", "OK" if ok else ("MISMATCH " + str(res)), flush=True)
print("mismatching rounds:", bad)
sys.exit(1 if bad else 0)
