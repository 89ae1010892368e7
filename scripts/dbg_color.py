import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, oracle_bind, synth
from test_gpu_color import _level_case
orc = oracle_bind.load(); ctx = nct.Context(0)
for case in [(48, 48, 12, 12, (3, 3), 4, 2), (40, 56, 20, 28, (5, 7), 4, 3), (32, 32, 32, 32, (2, 2), 16, 4)]:
    H, W, h, w, grid, samples, layer = case
    err, s_lvl, g_lvl, s_full, ids, ws = _level_case(20 + layer, H, W, h, w, grid, samples, orc)
    go, gs = ctx.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    oo, os_ = orc.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    for k in ("ab_local", "ab_nonlocal", "ab_up", "roughness", "ab_wls"):
        d = np.abs(gs[k] - os_[k]); print(case[:4], k, "maxabs", d.max(), "maxrel", (d / (np.abs(os_[k]) + 1e-12)).max(), "scale", np.abs(os_[k]).max())
    print("cg", gs["cg_iters"], os_["cg_iters"], "wls", gs["wls_iters"], os_["wls_iters"])
    d = np.abs(go.astype(int) - oo.astype(int)); print("out maxdiff", d.max(), "frac", (d > 0).mean())
