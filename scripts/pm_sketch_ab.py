"""A/B of the exact sketch pre-rejection of far random samples (k_pm_sketch.hip): one pair through nct_pair_run per setting, in one process (the switches are read
when a context is created). Prints PatchMatch per level, the pair total, the CRC of the result (must not move) and, from a counted run, tested / rejected samples.
usage: python scripts/pm_sketch_ab.py [size | natural case e.g. in4_tar4_2] [runs]"""
import os, sys, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np
import nct, synth
from caffemodel_io import synthetic_vgg19
NAT = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else None
S = int(sys.argv[1]) if len(sys.argv) > 1 and not NAT else 700
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ws, bs = synthetic_vgg19(19)
if NAT:
    from PIL import Image
    a, b, bds = NAT.split("_")
    load = lambda nme: np.ascontiguousarray(np.asarray(Image.open(os.path.join("tests", "golden", "natural", nme + ".png")).convert("RGB"))[..., ::-1])
    src, ref = load(a), load(b)
else:
    src, ref, bds = synth.image(1000, S, S), synth.image(1001, S, S), 2.0
for sk, mag in ((0, 8), (1, 16), (1, 8), (1, 4), (1, 2), (0, 8), (1, 8)):
    os.environ["NCT_PM_SKETCH"] = str(sk); os.environ["NCT_PM_SKETCH_MAG"] = str(mag)
    with nct.Context(0) as c:
        c.vgg19_load_raw(ws, bs)
        prm = nct.Params.default(); prm.bds_weight = float(bds)
        c.pair_upload(src, ref)
        tms = [c.pair_run(prm, want_timing=True) for _ in range(runs)]
        out = c.pair_download()
        tms = tms[1:]
        pm = np.median([t["pm_level_ms"] for t in tms], axis=0)
        tot = np.median([t["total_ms"] for t in tms])
        prm.flags |= nct.FLAG_COUNT_EVALS
        c.pair_run(prm, want_timing=True)
        tested, rej = c.counter(nct.CTR_PM_SKETCH_TESTED), c.counter(nct.CTR_PM_SKETCH_REJECTED)
        print(f"sketch {sk} mag {mag:2d}: pair {tot:6.2f} ms  pm {pm.sum():6.2f} = {[round(float(x), 2) for x in pm]}  crc {zlib.crc32(out.tobytes()):08x}  tested {tested} rejected {rej} ({100.0 * rej / max(tested, 1):.1f} %)", flush=True)
