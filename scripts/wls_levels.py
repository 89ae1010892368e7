"""WLS stage time per level + total for one 700x700 pair in flight (median of N), CRC of the result. usage: [NCT_WLS_FORECAST=0] python scripts/wls_levels.py [runs=7]"""
import sys, zlib, statistics
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 7
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
c.pair_upload(synth.image(1000, 700, 700), synth.image(1001, 700, 700))
prm = nct.Params.default()
c.pair_run(prm); c.pair_run(prm)
lv, tot, single = [], [], []
for _ in range(runs):
    tm = c.pair_run(prm, want_timing=True)
    lv.append(list(tm["wls_level_ms"])); tot.append(tm["wls_ms"]); single.append(tm["total_ms"])
med = [round(statistics.median(x[k] for x in lv), 3) for k in range(5)]
print("wls_level_ms", med, "wls_ms %.2f total %.2f iters %s crc %08x" % (statistics.median(tot), statistics.median(single), list(tm["wls_iters"]), zlib.crc32(c.pair_download().tobytes())), flush=True)
