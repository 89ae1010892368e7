"""Throughput probe: K pairs in flight on ONE GPU (K contexts, K host threads). Not part of the bench contract."""
import sys, os, tempfile, threading, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19, write_caffemodel
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ws, bs = synthetic_vgg19(19)
td = tempfile.mkdtemp(); path = os.path.join(td, "v.caffemodel"); write_caffemodel(path, ws, bs, fmt="v1")
ctxs = []
for k in range(K):
    c = nct.Context(0); c.vgg19_load_caffemodel(path)
    c.pair_upload(synth.image(1000 + 2 * k, 700, 700), synth.image(1001 + 2 * k, 700, 700))
    ctxs.append(c)
prm = nct.Params.default()
for c in ctxs: c.pair_run(prm)            # warm-up (arena allocation)
bar = threading.Barrier(K + 1)
def work(c):
    bar.wait()
    for _ in range(steps): c.pair_run(prm)
    bar.wait()
ths = [threading.Thread(target=work, args=(c,)) for c in ctxs]
for t in ths: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
for t in ths: t.join()
print("K=%d pairs in flight: %.2f pairs/s (%.1f ms per pair-slot)" % (K, K * steps / dt, 1e3 * dt / steps))
