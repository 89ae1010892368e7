"""Per-level CRCs of the GPU path on the reference's demo pairs (tests/golden/natural/*.png), in the record format of tests/golden/gen_natural.py — so that a fixture the CPU
oracle is still computing can be compared as soon as it exists (python scripts/natural_crcs.py check <json> compares a dump with the committed / generated .npz files).
usage: python scripts/natural_crcs.py [case ...] > out.json        |  python scripts/natural_crcs.py check out.json"""
import os, sys, json, zlib
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python"); sys.path.insert(0, "scripts")
import numpy as np
from natural_report import CASES, load_bgr
crc = lambda a: zlib.crc32(np.ascontiguousarray(a).tobytes())
if len(sys.argv) > 2 and sys.argv[1] == "check":
    dump = json.load(open(sys.argv[2])); bad = 0
    for name, d in dump.items():
        f = os.path.join("tests", "golden", "natural", f"pair_{name}.npz")
        if not os.path.exists(f):
            print(name, "no fixture yet"); continue
        g = np.load(f)
        ok = int(g["crc_canonical"]) == d["crc"] and all(int(g["level_crc_canonical"][l]) == d["level_crc_result"][l] for l in range(5))
        for k in ("ann", "bnn", "annd", "bnnd", "guide", "err"):
            ok = ok and all(int(g["level_crc_" + k][l]) == d["level_crc_" + k][l] for l in range(5))
        print(name, "GPU == oracle at every level" if ok else "MISMATCH", "| exact-S2 vs canonical differing bytes:", int(g["idx"].size) if "idx" in g else "n/a")
        bad += 0 if ok else 1
    sys.exit(1 if bad else 0)
import nct
from caffemodel_io import synthetic_vgg19
ws, bs = synthetic_vgg19(19)
out = {}
with nct.Context(0) as c:
    c.vgg19_load_raw(ws, bs)
    for name in (sys.argv[1:] or list(CASES)):
        s, r, bds = CASES[name]; src, ref = load_bgr(s), load_bgr(r)
        prm = nct.Params.default(); prm.bds_weight = bds
        c.pair_upload(src, ref)
        lv = c.pair_run_levels(src.shape, ref.shape, prm)
        got = c.pair_download()
        rec = {"crc": crc(got), "sum": int(got.astype(np.uint64).sum()), "level_crc_result": [crc(lv["result"][l]) for l in range(5)], "shape": list(src.shape[:2] + ref.shape[:2]),
               "hub_blocks": [c.counter(nct.CTR_S1_HUB_BLOCKS_L0 + l) for l in range(5)]}
        for k in ("ann", "bnn", "annd", "bnnd", "guide", "err"):
            rec["level_crc_" + k] = [crc(lv[k][l]) for l in range(5)]
        tm = c.pair_run(prm, want_timing=True)
        rec["ms"] = tm["total_ms"]; rec["wls_iters"] = list(tm["wls_iters"])
        out[name] = rec
print(json.dumps(out))
