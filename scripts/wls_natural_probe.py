"""Why do the WLS solves of natural pairs need 2-3x the PCG iterations of the synthetic pair? Per level: PCG iterations of the six right-hand sides, share of pixels whose
roughness is 1e-6 (a*Lab+b outside [0,1]: ColorTransfer.cpp:476-486 — those rows have almost no data term), size of the largest such connected region, share of exactly flat
L neighbours (|dL| = 0: the largest edge weights lambda / 1e-4).   usage: python scripts/wls_natural_probe.py [case ...]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python")); sys.path.insert(0, os.path.join(REPO, "scripts"))
import numpy as np, nct, synth
from scipy import ndimage
from caffemodel_io import synthetic_vgg19
from natural_report import CASES, load_bgr
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
for name in (sys.argv[1:] or ["in0_tar0_2", "in1_tar1_2", "synthetic"]):
    if name == "synthetic":
        src, ref, bds = synth.image(1000, 452, 680), synth.image(1001, 600, 960), 2.0
    else:
        s, r, bds = CASES[name]; src, ref = load_bgr(s), load_bgr(r)
    prm = nct.Params.default(); prm.bds_weight = bds
    c.pair_upload(src, ref)
    lv = c.pair_run_levels(src.shape, ref.shape, prm, want_color=True)
    H, W = src.shape[:2]
    L = c.bgr2lab(src)[..., 0].astype(int)
    flat = ((np.diff(L, axis=1) == 0).mean() + (np.diff(L, axis=0) == 0).mean()) / 2
    print(f"## {name}: {W}x{H}, share of neighbour pairs with |dL| = 0: {flat:.3f}")
    for l, col in enumerate(lv["color"]):
        rough = col["roughness"].reshape(H, W)
        low = rough < 0.5
        lab, nl = ndimage.label(low)
        biggest = int(np.bincount(lab.ravel())[1:].max()) if nl else 0
        print(f"level {l}: wls iters per rhs {col['wls_iters'].tolist()}  roughness 1e-6 on {low.mean():.3f} of the pixels, {nl} regions, largest {biggest} px")
