"""One bench pair (700x700 by default) through nct_pair_run, `runs` times — the workload rocprofv3 is pointed at for per-kernel traces and
PMC passes of the pipeline's own kernels (k_pm_step<1, MODE> ... as the pipeline launches them).
usage: python scripts/pair_only.py [size | natural case, e.g. in4_tar4_2] [runs]      env: FEAT16=1 -> NCT_FLAG_FEAT16"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import nct, synth
from caffemodel_io import synthetic_vgg19
NAT = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else None
S = int(sys.argv[1]) if len(sys.argv) > 1 and not NAT else 700
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ws, bs = synthetic_vgg19(19)
c = nct.Context(0)
c.vgg19_load_raw(ws, bs)
prm = nct.Params.default()
if NAT:                       # the reference's demo inputs (tests/golden/natural/*.png: natural photographs), e.g. in4_tar4_2 = in4.png, tar4.png, bds 2
    import numpy as np
    from PIL import Image
    a, b, bds = NAT.split("_")
    load = lambda nme: np.ascontiguousarray(np.asarray(Image.open(os.path.join("tests", "golden", "natural", nme + ".png")).convert("RGB"))[..., ::-1])
    c.pair_upload(load(a), load(b)); prm.bds_weight = float(bds)
else:
    c.pair_upload(synth.image(1000, S, S), synth.image(1001, S, S))
if os.environ.get("FEAT16"):
    prm.flags |= nct.FLAG_FEAT16
for i in range(runs):
    tm = c.pair_run(prm, want_timing=True)
    print("run", i, "total ms %.2f" % tm["total_ms"], "pm per level", [round(x, 2) for x in tm["pm_level_ms"]], flush=True)
