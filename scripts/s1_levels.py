"""S1 stage time per level (stream events, one 700x700 pair in flight, median of N runs) + the CRC of the result, for the library NCT_LIB selects.
usage: [NCT_LIB=...] [NCT_S1_HUB_HINT=0] python scripts/s1_levels.py [runs=7]"""
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
    lv.append(list(tm["nonlocal_level_ms"])); tot.append(tm["nonlocal_ms"]); single.append(tm["total_ms"])
med = [round(statistics.median(x[k] for x in lv), 3) for k in range(5)]
print("nonlocal_level_ms", med, "nonlocal_ms %.2f total %.2f crc %08x" % (statistics.median(tot), statistics.median(single), zlib.crc32(c.pair_download().tobytes())), flush=True)
