"""The reference's own demo inputs (tests/golden/natural/*.png, natural photographs) through the stage clock, next to a tests/synth.py pair of the SAME sizes
(VERDICT r4 item 1): per-stage and per-level times (stream events, one pair in flight, median of N runs), PCG / PatchMatch counters, the kNN in-degree
distribution per level (what the S1 kernels are sized on) and the number of completeness sources per target of the BDS vote. Markdown on stdout.
usage: python scripts/natural_report.py [runs=5] [case ...]"""
import os, sys, statistics, zlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))
import numpy as np, nct, synth
from PIL import Image
from caffemodel_io import synthetic_vgg19
CASES = {"in1_tar1_2": ("in1", "tar1", 2.0), "in4_tar4_2": ("in4", "tar4", 2.0), "in4_tar4_0": ("in4", "tar4", 0.0), "in4_tar4_8": ("in4", "tar4", 8.0),
         "in0_tar0_2": ("in0", "tar0", 2.0),
         # round 6: the remaining lines of demo/example/pairs.txt (3, 4, 6, 8) — all nine demo pairs are fixtures now
         "in2_tar2_2": ("in2", "tar2", 2.0), "in3_tar3_2": ("in3", "tar3", 2.0), "in4_tar4_1": ("in4", "tar4", 1.0), "in4_tar4_4": ("in4", "tar4", 4.0)}


def load_bgr(name):
    return np.ascontiguousarray(np.asarray(Image.open(os.path.join(REPO, "tests", "golden", "natural", name + ".png")).convert("RGB"))[..., ::-1])


def stage_times(c, src, ref, prm, runs):
    c.pair_upload(src, ref)
    c.pair_run(prm); c.pair_run(prm)
    tms = [c.pair_run(prm, want_timing=True) for _ in range(runs)]
    med = lambda k: statistics.median(t[k] for t in tms)
    medl = lambda k: [round(statistics.median(t[k][l] for t in tms), 3) for l in range(5)]
    out = {k: round(med(k), 2) for k in ("total_ms", "vgg_ms", "cluster_ms", "patchmatch_ms", "vote_ms", "knn_ms", "nonlocal_ms", "wls_ms", "other_ms")}
    for k in ("pm_level_ms", "vote_level_ms", "nonlocal_level_ms", "wls_level_ms"):
        out[k] = medl(k)
    out["wls_iters"] = list(tms[-1]["wls_iters"]); out["launches"] = list(tms[-1]["pm_level_launches"])
    out["crc"] = "%08x" % zlib.crc32(c.pair_download().tobytes())
    return out


def graph_stats(c, src, ref, prm):
    c.pair_upload(src, ref)
    lv = c.pair_run_levels(src.shape, ref.shape, prm, want_color=True)
    labels = lv["labels"]; nl = int(labels.max()) + 1
    rows = []
    for l, (ah, aw, bh, bw) in enumerate(lv["dims"]):
        # the level's source image: progressive bilinear pyramid (main.cu:104-108) of the ORIGINAL source — the graph is built on it (main.cu:351-359)
        im = src
        for (h2, w2, _, _) in reversed(lv["dims"][l:-1]):
            im = c.resize_u8c3(im, h2, w2)
        lab = c.bgr2lab(im)
        ids, w = c.knn_graph(lab, labels, nl, 1 << l)
        n = ids.shape[0]
        real = ids != np.arange(n)[:, None]                                     # padded self edges carry weight 0
        deg = np.bincount(ids[real], minlength=n)
        ncol = np.unique(lab.reshape(-1, 3), axis=0).shape[0]
        bnn = lv["bnn"][l]
        x = (bnn & 0xFFF).astype(np.int64); y = ((bnn >> 12) & 0xFFF).astype(np.int64)
        cnt = np.zeros((ah + 2, aw + 2), np.int64); np.add.at(cnt, (y.ravel() + 1, x.ravel() + 1), 1)
        tot = np.zeros((ah, aw), np.int64)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                tot += cnt[1 - dy:1 - dy + ah, 1 - dx:1 - dx + aw]
        t = tot.ravel()
        rows.append(dict(level=l, size="%dx%d" % (aw, ah), pixels=n, colours=ncol, px_per_colour=round(n / ncol, 2), deg_max=int(deg.max()), deg_p99=int(np.percentile(deg, 99)),
                         deg_gt64=int((deg > 64).sum()), deg_gt512=int((deg > 512).sum()), deg_gt4096=int((deg > 4096).sum()), padded=int((~real).sum()),
                         vote_mean=round(float(t.mean()), 2), vote_p99=int(np.percentile(t, 99)), vote_max=int(t.max()), vote_gt72=int((t > 72).sum())))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    runs = int(args.pop(0)) if args and args[0].isdigit() else 5
    ws, bs = synthetic_vgg19(19)
    c = nct.Context(0); c.vgg19_load_raw(ws, bs)
    print("device:", c.device_name(), " runs per case:", runs, "\n")
    for name in (args or list(CASES)):
        s, r, bds = CASES[name]
        src, ref = load_bgr(s), load_bgr(r)
        prm = nct.Params.default(); prm.bds_weight = bds
        nat = stage_times(c, src, ref, prm, runs)
        syn = stage_times(c, synth.image(1000, *src.shape[:2]), synth.image(1001, *ref.shape[:2]), prm, runs)
        print("## %s  (source %dx%d, reference %dx%d, bds %.1f)\n" % (name, src.shape[1], src.shape[0], ref.shape[1], ref.shape[0], bds))
        print("| stage | natural ms | synthetic (same sizes) ms | ratio |\n|---|---|---|---|")
        for k in ("total_ms", "vgg_ms", "cluster_ms", "patchmatch_ms", "vote_ms", "knn_ms", "nonlocal_ms", "wls_ms", "other_ms"):
            print("| %s | %.2f | %.2f | %.2f |" % (k, nat[k], syn[k], nat[k] / max(syn[k], 1e-9)))
        for k in ("pm_level_ms", "vote_level_ms", "nonlocal_level_ms", "wls_level_ms", "wls_iters"):
            print("| %s | %s | %s | |" % (k, nat[k], syn[k]))
        print("| result crc | %s | %s | |\n" % (nat["crc"], syn["crc"]))
        rows = graph_stats(c, src, ref, prm)
        keys = list(rows[0].keys())
        print("| " + " | ".join(keys) + " |\n|" + "---|" * len(keys))
        for row in rows:
            print("| " + " | ".join(str(row[k]) for k in keys) + " |")
        print(flush=True)
