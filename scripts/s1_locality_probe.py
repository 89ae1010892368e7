"""How much of the finest-level S1 operator is gather locality? Times nct_local_color_transfer (layer 4, 700x700) with the real kNN
graph and with a fake graph whose neighbours are the next 8 pixels in memory (same degree, perfectly local gathers)."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
ctx = nct.Context(0)
S = 700
src = synth.image(1000, S, S); guide = synth.image(1001, S, S)
lab = ctx.bgr2lab(src)
labels = np.zeros((44, 44), np.int32)
ids, ws = ctx.knn_graph(lab, labels, 1, 16)
n = S * S
fake = ((np.arange(n)[:, None] + np.arange(1, 9)[None, :]) % n).astype(np.int32)
perm = np.random.default_rng(0).permutation(n).astype(np.int32)
rnd = perm[((np.arange(n)[:, None] + np.arange(1, 9)[None, :]) % n)]          # same structure, random targets
err = np.random.default_rng(1).random((S, S)).astype(np.float32)
for name, g in (("real kNN graph", ids), ("local fake graph", fake), ("random fake graph", rnd)):
    ctx.local_color_transfer(err, src, guide, src, g, ws, 4)
    t = time.perf_counter()
    for _ in range(3): ctx.local_color_transfer(err, src, guide, src, g, ws, 4)
    print("%-20s %.1f ms per call" % (name, (time.perf_counter() - t) / 3 * 1e3))
