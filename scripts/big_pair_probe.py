import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19
ctx = nct.Context(0)
ws, bs = synthetic_vgg19(19); ctx.vgg19_load_raw(ws, bs)
src, ref = synth.image(5, 1500, 2000), synth.image(6, 1700, 1300)
t = time.time(); a = ctx.process_pair(src, ref); t1 = time.time() - t
t = time.time(); b = ctx.process_pair(src, ref); t2 = time.time() - t
print("1500x2000 pair: %.2f s / %.2f s, deterministic:" % (t1, t2), np.array_equal(a, b), "changed:", float(np.abs(a.astype(int) - src.astype(int)).mean()))
