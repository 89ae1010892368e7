import sys, zlib, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
c.pair_upload(synth.image(1000, 700, 700), synth.image(1001, 700, 700))
prm = nct.Params.default()
c.pair_run(prm); c.pair_run(prm)
lv, tot = [], []
for _ in range(9):
    tm = c.pair_run(prm, want_timing=True)
    lv.append(list(tm["pm_level_ms"])); tot.append(tm["total_ms"])
print("pm_level_ms", [round(statistics.median(x[k] for x in lv), 3) for k in range(5)], "pm %.2f total %.2f crc %08x" % (sum(statistics.median(x[k] for x in lv) for k in range(5)), statistics.median(tot), zlib.crc32(c.pair_download().tobytes())))
