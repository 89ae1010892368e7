"""Per-right-hand-side iteration counts of the WLS solves of a 700x700 pair (instrumented run prints the max only)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
ctx = nct.Context(0)
S = 700
src = synth.image(1000, S, S); guide = synth.image(1001, S, S)
lab = ctx.bgr2lab(src)
err = np.random.default_rng(1).random((S, S)).astype(np.float32)
sizes = [44, 88, 175, 350, 700]
img = src
pyr = [src]
while pyr[-1].shape[0] > 44:
    nh = (pyr[-1].shape[0] + 1) // 2
    pyr.append(ctx.resize_u8c3(pyr[-1], nh, nh))
gpyr = [guide]
while gpyr[-1].shape[0] > 44:
    nh = (gpyr[-1].shape[0] + 1) // 2
    gpyr.append(ctx.resize_u8c3(gpyr[-1], nh, nh))
for layer, (s_l, g_l) in enumerate(zip(reversed(pyr), reversed(gpyr))):
    h = s_l.shape[0]
    ids, ws = ctx.knn_graph(ctx.bgr2lab(s_l), np.zeros((44, 44), np.int32), 1, 1 << layer)
    e = np.random.default_rng(layer).random((h, h)).astype(np.float32)
    out, st = ctx.local_color_transfer(e, s_l, g_l, src, ids, ws, layer, want_stages=True)
    print("layer", layer, "size", h, "wls iters per rhs", st["wls_iters"].tolist())
