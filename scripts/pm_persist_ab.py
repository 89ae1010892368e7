"""PatchMatch as one persistent launch per level (NCT_PM_PERSIST=1, k_pm_level) against one launch per step (0), inside the real pipeline: per-level kernel time, pair time,
launches, result CRC — each configuration in its own process, alternating, best of `runs`.   usage: python scripts/pm_persist_ab.py [size=700] [runs=3] [inflight=1]"""
import os, sys, json, zlib, subprocess
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
S = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 700
if "child" in sys.argv:
    import nct, synth
    from caffemodel_io import synthetic_vgg19
    S = int(sys.argv[sys.argv.index("child") + 1])
    ws, bs = synthetic_vgg19(19)
    c = nct.Context(0)
    c.vgg19_load_raw(ws, bs)
    c.pair_upload(synth.image(1000, S, S), synth.image(1001, S, S))
    prm = nct.Params.default()
    prm.flags |= nct.FLAG_LATENCY
    c.pair_run(prm)
    tms = [c.pair_run(prm, want_timing=True) for _ in range(int(os.environ.get("RUNS", "3")))]
    tm = min(tms, key=lambda t: t["total_ms"])
    out = c.pair_download()
    print(json.dumps({"total_ms": tm["total_ms"], "pm_ms": tm["patchmatch_ms"], "pm_level_ms": tm["pm_level_ms"], "launches": tm["pm_level_launches"], "crc": zlib.crc32(out.tobytes())}))
    sys.exit(0)
runs = sys.argv[2] if len(sys.argv) > 2 else "3"
for rep in range(2):
    for name, env in (("per-step", {"NCT_PM_PERSIST": "0"}), ("persistent", {"NCT_PM_PERSIST": "1"})) + tuple(
            ("persistent wgs=%s" % w, {"NCT_PM_PERSIST": "1", "NCT_PM_PERSIST_WGS": w}) for w in os.environ.get("WGS", "").split(",") if w) + tuple(
            (v.split("=")[0], {"NCT_PM_PERSIST": v.split("=")[1], "NCT_LIB": v.split("=")[2]}) for v in os.environ.get("VARIANTS", "").split(",") if v):     # name=persist=path
        e = dict(os.environ); e.update(env); e["RUNS"] = runs
        if "NCT_LIB" in env: e["NCT_LIB"] = os.path.abspath(env["NCT_LIB"])
        r = subprocess.run([sys.executable, __file__, "child", str(S)], env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", r.stdout[-400:], r.stderr[-1200:]); continue
        d = json.loads(line[0])
        print(f"{name:22s} total {d['total_ms']:7.2f} ms  PM {d['pm_ms']:6.2f} ms  per level {[round(x, 2) for x in d['pm_level_ms']]}  launches {d['launches']}  crc {d['crc']}", flush=True)
