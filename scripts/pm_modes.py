"""PatchMatch evaluation modes (k_patchmatch.hip) inside the real pipeline: per-level kernel time, candidate evaluations and
accepted candidates on the 700x700 bench pair, for the exact fp32 path (row rejection) and the opt-in -feat16 mode (FEAT16=1 in
the child), for the default library and every build under lib/variants/.
usage: python scripts/pm_modes.py [size]"""
import os, sys, json, zlib, subprocess
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np

S = int(sys.argv[1]) if len(sys.argv) > 1 else 700
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import nct, synth
    from caffemodel_io import synthetic_vgg19
    ws, bs = synthetic_vgg19(19)
    c = nct.Context(0)
    c.vgg19_load_raw(ws, bs)
    src, ref = synth.image(1000, S, S), synth.image(1001, S, S)
    c.pair_upload(src, ref)
    prm = nct.Params.default()
    if os.environ.get("FEAT16"):
        prm.flags |= nct.FLAG_FEAT16
    if os.environ.get("LATENCY"):
        prm.flags |= nct.FLAG_LATENCY
    c.pair_run(prm)
    tms = [c.pair_run(prm, want_timing=True) for _ in range(3)]
    tm = min(tms, key=lambda t: t["total_ms"])
    prm.flags |= nct.FLAG_COUNT_EVALS
    tc = c.pair_run(prm, want_timing=True)
    out = c.pair_download()
    print(json.dumps({"total_ms": tm["total_ms"], "pm_ms": tm["patchmatch_ms"], "pm_level_ms": tm["pm_level_ms"], "evals": tc["pm_level_evals"],
                      "accepted": tc["pm_level_accepted"], "crc": zlib.crc32(out.tobytes()), "wls_iters": tm["wls_iters"],
                      "stages": {k: tm[k] for k in ("vgg_ms", "cluster_ms", "patchmatch_ms", "vote_ms", "knn_ms", "color_ms", "nonlocal_ms", "wls_ms", "other_ms")}}))
    sys.exit(0)

import glob
libs = [("default", None)] + [(os.path.basename(p)[7:-3], p) for p in sorted(glob.glob("neural-color-transfer_amd/lib/variants/libnct_*.so"))]
for lname, lpath in libs:
    for name, env in ((("fp32", {}),) if os.environ.get("ONLY_FP32") else (("fp32", {}), ("feat16", {"FEAT16": "1"}))):
        e = dict(os.environ); e.update(env)
        if lpath:
            e["NCT_LIB"] = os.path.abspath(lpath)
        r = subprocess.run([sys.executable, __file__, str(S), "child"], env=e, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(lname, name, "FAILED", r.stderr[-800:]); continue
        d = json.loads(line[0])
        print(f"{lname:8s} {name:7s} total {d['total_ms']:7.2f} ms  PM {d['pm_ms']:6.2f} ms  per level {[round(x, 2) for x in d['pm_level_ms']]}  crc {d['crc']}")
        print(f"{'':16s} evals {d['evals']}  accepted {d['accepted']}  wls_iters {d['wls_iters']}")
        print(f"{'':16s} stages {json.dumps({k: round(v, 2) for k, v in d['stages'].items()})}", flush=True)
