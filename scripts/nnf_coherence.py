"""How coherent is the finest-level NNF on the synthetic bench data? (fraction of pixels whose right/down neighbour maps to the
shifted match). Random features (the roofline fixture) vs conv1_1 features of the synthetic bench pair."""
import sys, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19
ctx = nct.Context(0)
S = 700
def coherence(nnf):
    x = (nnf & 0xFFF).astype(np.int64); y = (nnf >> 12).astype(np.int64)
    cx = ((x[:, 1:] == x[:, :-1] + 1) & (y[:, 1:] == y[:, :-1])).mean()
    cy = ((y[1:, :] == y[:-1, :] + 1) & (x[1:, :] == x[:-1, :])).mean()
    return cx, cy
def run(name, fa, fb):
    a, b = ctx.feat_normalize(fa), ctx.feat_normalize(fb)
    nnf0 = ctx.nnf_init(S, S, S, S)
    nnf, d = ctx.patchmatch(a, b, nnf0, iters=10, rs_max=32, seed=5)
    print("%-28s coherence x %.3f y %.3f  mean dist %.4f" % (name, *coherence(nnf.reshape(S, S)), float(d.mean())))
run("random features (fixture)", synth.features(11, 64, S, S), synth.features(12, 64, S, S))
ws, bs = synthetic_vgg19(19)
ctx.vgg19_load_raw(ws, bs)
fa = ctx.vgg19_features(synth.image(1000, S, S), 1)[0]
fb = ctx.vgg19_features(synth.image(1001, S, S), 1)[0]
run("conv1_1 of the bench pair", fa, fb)
