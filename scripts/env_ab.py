"""A/B of environment switches inside the real pipeline: the bench pair through nct_pair_run in a fresh process per configuration, alternating, best of `runs`: pair ms, stage ms,
PatchMatch / S1 / S2 per level, result CRC.   usage: python scripts/env_ab.py <size> <runs> NAME=VALUE[,NAME=VALUE...] [more configurations ...]   ("-" = no variables)"""
import os, sys, json, zlib, subprocess
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
if sys.argv[1] == "child":
    import nct, synth
    from caffemodel_io import synthetic_vgg19
    S = int(sys.argv[2])
    ws, bs = synthetic_vgg19(19)
    c = nct.Context(0)
    c.vgg19_load_raw(ws, bs)
    c.pair_upload(synth.image(1000, S, S), synth.image(1001, S, S))
    prm = nct.Params.default()
    if os.environ.get("LATENCY"): prm.flags |= nct.FLAG_LATENCY
    c.pair_run(prm)
    tms = [c.pair_run(prm, want_timing=True) for _ in range(int(sys.argv[3]))]
    tm = min(tms, key=lambda t: t["total_ms"])
    out = c.pair_download()
    print(json.dumps({"total_ms": tm["total_ms"], "pm_ms": tm["patchmatch_ms"], "s1_ms": tm["nonlocal_ms"], "s2_ms": tm["wls_ms"], "vgg_ms": tm["vgg_ms"], "pm_level_ms": tm["pm_level_ms"],
                      "crc": zlib.crc32(out.tobytes())}))
    sys.exit(0)
S, runs, cfgs = sys.argv[1], sys.argv[2], sys.argv[3:]
for rep in range(2):
    for cfg in cfgs:
        e = dict(os.environ)
        if cfg != "-":
            for kv in cfg.split(","):
                k, v = kv.split("="); e[k] = v
        r = subprocess.run([sys.executable, __file__, "child", S, runs], env=e, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(cfg, "FAILED", r.stdout[-300:], r.stderr[-800:]); continue
        d = json.loads(line[0])
        print(f"{cfg:34s} total {d['total_ms']:7.2f}  vgg {d['vgg_ms']:5.2f}  PM {d['pm_ms']:6.2f} {[round(x, 2) for x in d['pm_level_ms']]}  S1 {d['s1_ms']:5.2f}  S2 {d['s2_ms']:5.2f}  crc {d['crc']}", flush=True)
