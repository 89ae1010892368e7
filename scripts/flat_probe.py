import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
def run(name, src, ref):
    c.pair_upload(src, ref); c.pair_run(); tm = c.pair_run(want_timing=True)
    print("%-28s total %.1f pm %.1f vote %.1f nonlocal %.1f (levels %s) wls %.1f iters %s" % (name, tm["total_ms"], tm["patchmatch_ms"], tm["vote_ms"], tm["nonlocal_ms"], [round(x,1) for x in tm["nonlocal_level_ms"]], tm["wls_ms"], list(tm["wls_iters"])), "hub blocks", [c.counter(nct.CTR_S1_HUB_BLOCKS_L0 + l) for l in range(5)], flush=True)
S = 700
a, b = synth.image(1000, S, S), synth.image(1001, S, S)
run("synthetic", a, b)
a2, b2 = a.copy(), b.copy(); a2[:105] = 0; a2[-105:] = 0; b2[:105] = 0; b2[-105:] = 0
run("letterboxed (30% black)", a2, b2)
a3 = a.copy(); a3[:, :350] = (200, 180, 90)
run("half one colour", a3, b)
run("whole image one colour", np.full((S, S, 3), (60, 120, 200), np.uint8), np.full((S, S, 3), (200, 90, 40), np.uint8))
