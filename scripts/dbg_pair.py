import sys, os, time, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, oracle_bind, synth
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
ctx = nct.Context(0); ctx.vgg19_load_raw(ws, bs)
src, ref = synth.image(1000, 64, 64), synth.image(1001, 64, 64)
t = time.time(); got, tm = ctx.process_pair(src, ref, want_timing=True); print("gpu pair %.2fs" % (time.time() - t), tm, flush=True)
orc = oracle_bind.load()
t = time.time(); exp, lv = orc.process_pair(src, ref, ws, bs, want_levels=True); print("oracle pair %.2fs" % (time.time() - t), flush=True)
d = np.abs(got.astype(int) - exp.astype(int)); print("Linf", d.max(), "frac", (d > 0).mean(), flush=True)
