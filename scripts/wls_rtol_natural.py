"""S2 stopping tolerance on the natural fixtures: the GPU result at NCT_WLS_RTOL = r against the oracle's EXACT-S2 image (rebuilt from tests/golden/natural/pair_<case>.npz), with the
PCG iterations and the WLS stage time. One process per tolerance (the context reads the variable at creation).   usage: python scripts/wls_rtol_natural.py [case ...]"""
import os, sys, subprocess, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, json, zlib
sys.path.insert(0, os.path.join(%(repo)r, "tests")); sys.path.insert(0, os.path.join(%(repo)r, "neural-color-transfer_amd", "python")); sys.path.insert(0, os.path.join(%(repo)r, "scripts"))
import numpy as np, nct
from caffemodel_io import synthetic_vgg19
from natural_report import CASES, load_bgr
ws, bs = synthetic_vgg19(19)
c = nct.Context(0); c.vgg19_load_raw(ws, bs)
out = {}
for name in %(cases)r:
    g = np.load(os.path.join(%(repo)r, "tests", "golden", "natural", "pair_%%s.npz" %% name))
    s, r, bds = CASES[name]; src, ref = load_bgr(s), load_bgr(r)
    prm = nct.Params.default(); prm.bds_weight = bds
    c.pair_upload(src, ref); c.pair_run(prm); tm = c.pair_run(prm, want_timing=True); got = c.pair_download()
    canon_ok = zlib.crc32(got.tobytes()) == int(g["crc_canonical"])
    exact = None
    # the exact-S2 image = canonical + delta, and the canonical image is what the default rtol (3e-8) gives: rebuild it from the fixture's CRC-checked run only when this run IS canonical
    out[name] = {"wls_ms": round(tm["wls_ms"], 2), "iters": list(tm["wls_iters"]), "crc": zlib.crc32(got.tobytes()), "canonical": canon_ok}
    np.save(os.path.join(%(tmp)r, "%%s_%%s.npy" %% (name, os.environ.get("NCT_WLS_RTOL", "3e-8"))), got)
print(json.dumps(out))
'''
cases = sys.argv[1:] or ["in0_tar0_2", "in1_tar1_2", "in4_tar4_2"]
import tempfile, numpy as np
tmp = tempfile.mkdtemp()
res = {}
for rtol in ("3e-8", "1e-7", "1e-8", "1e-9", "1e-10"):           # 3e-8 = the default
    r = subprocess.run([sys.executable, "-c", CODE % {"repo": REPO, "cases": cases, "tmp": tmp}], env=dict(os.environ, NCT_WLS_RTOL=rtol), capture_output=True, text=True)
    res[rtol] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
for name in cases:
    g = np.load(os.path.join(REPO, "tests", "golden", "natural", "pair_%s.npz" % name))
    canon = np.load(os.path.join(tmp, "%s_3e-8.npy" % name))
    assert res["3e-8"][name]["canonical"], "the default tolerance must reproduce the canonical fixture"
    exact = canon.astype(np.int16).reshape(-1); exact[g["idx"]] += g["delta"]; exact = exact.astype(np.uint8).reshape(canon.shape)
    print("##", name, "(exact-S2 oracle vs canonical: %d differing bytes)" % g["idx"].size)
    for rtol in res:
        got = np.load(os.path.join(tmp, "%s_%s.npy" % (name, rtol)))
        d = got.astype(int) - exact.astype(int)
        mse = (d.astype(float) ** 2).reshape(-1, 3).mean(0).max()
        print("rtol %-6s iterations %-22s wls %.2f ms   vs exact-S2 image: %d differing bytes, L-inf %d, PSNR(min channel) %s" % (rtol, res[rtol][name]["iters"], res[rtol][name]["wls_ms"], int((d != 0).sum()), int(np.abs(d).max()), "identical" if mse == 0 else "%.2f dB" % (10 * np.log10(255.0 ** 2 / mse))))
