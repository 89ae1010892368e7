"""How many completeness sources a target pixel of the BDS vote has (per pyramid level of the 700x700 bench pair): the inverse of the R->S field, summed over the 9 taps.
usage: python scripts/vote_sources_hist.py"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
c.pair_upload(synth.image(1000, 700, 700), synth.image(1001, 700, 700))
lv = c.pair_run_levels((700, 700), (700, 700))
for l, bnn in enumerate(lv["bnn"]):
    ah, aw = lv["ann"][l].shape
    x = (bnn & 0xFFF).astype(np.int64); y = ((bnn >> 12) & 0xFFF).astype(np.int64)          # nnf_x / nnf_y (nct_device.h)
    cnt = np.zeros((ah + 2, aw + 2), np.int64)
    np.add.at(cnt, (y.ravel() + 1, x.ravel() + 1), 1)                                      # sources whose correspondence is (x, y)
    tot = np.zeros((ah, aw), np.int64)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            tot += cnt[1 - dy:1 - dy + ah, 1 - dx:1 - dx + aw]
    t = tot.ravel()
    print("level %d (%dx%d): mean %.2f  p50 %d p90 %d p99 %d max %d  >32: %.3f %%  targets with 0: %.1f %%" % (l, aw, ah, t.mean(), np.percentile(t, 50), np.percentile(t, 90), np.percentile(t, 99), t.max(), 100.0 * (t > 32).mean(), 100.0 * (t == 0).mean()), flush=True)
