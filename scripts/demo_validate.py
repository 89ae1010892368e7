#!/usr/bin/env python3
"""Real-weights validation harness: runs the reference's own demo batch (demo/example/pairs.txt:1-9, incl. the BDS
sweep 0/1/2/4/8 on in4/tar4) through the CLI and scores every output against the result PNG the ORIGINAL binary produced
(demo/example/res/*.png — the only artefacts of the reference program that exist), per line: PSNR (min over channels), mean SSIM
(8x8 uniform windows on luma), L-inf and mean absolute difference.

Needs the Oxford weights, which are not in the repository and cannot be downloaded here:
    <model_dir>/vgg19/VGG_ILSVRC_19_layers.caffemodel          (default model_dir: $NCT_MODEL_DIR)
The demo inputs and reference results are read where they lie (default: $NCT_DEMO_DIR or /root/reference/demo/example); nothing is copied.
Expect tens of dB, not 50: the reference's PatchMatch is racy and its RNG unseeded, so its own output is not
reproducible run to run; this is a qualitative gate (SSIM), not the parity gate.

`--weights synthetic` is the DRY RUN of the same harness for boxes without the Oxford file (the GPU test box): it builds a stand-in demo directory with the demo's
GEOMETRY — ten PNGs of the demo's sizes and colour types (five of them RGBA, colour type 6: in0, in2, in4, tar0, tar4), the 9-line pairs.txt with the `in/` prefix
and the BDS sweep 0/1/2/4/8 — from tests/synth.py, a synthetic VGG19 caffemodel + deploy prototxt, and a res/ directory computed through the library's Python
binding on the alpha-dropped arrays; the CLI run must then reproduce res/ exactly (PSNR 99 = identical), which exercises decode -> alpha drop -> shrink ->
CLI -> output naming -> scoring end to end. Nothing from /root/reference is used or copied in that mode.

usage: python scripts/demo_validate.py [--model-dir D] [--demo-dir D] [--out D] [--json report.json] [--min-ssim 0.0] [--weights oxford|synthetic]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(REPO, "neural-color-transfer_amd", "bin", "neural_color_transfer")


def psnr_min_channel(a, b):
    mse = ((a.astype(np.float64) - b.astype(np.float64)) ** 2).reshape(-1, a.shape[-1]).mean(0)
    return float(min(99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m) for m in mse))


def ssim_luma(a, b, win=8):
    """Mean SSIM of the BT.601 luma planes over win x win uniform windows (K1 = 0.01, K2 = 0.03, L = 255)."""
    from scipy.ndimage import uniform_filter
    w = np.array([0.114, 0.587, 0.299])                     # BGR order
    x, y = a.astype(np.float64) @ w, b.astype(np.float64) @ w
    mx, my = uniform_filter(x, win), uniform_filter(y, win)
    vx, vy = uniform_filter(x * x, win) - mx * mx, uniform_filter(y * y, win) - my * my
    cxy = uniform_filter(x * y, win) - mx * my
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    s = ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
    h = win // 2
    return float(s[h:-h, h:-h].mean())


def score(got_bgr, ref_bgr):
    d = np.abs(got_bgr.astype(int) - ref_bgr.astype(int))
    return {"psnr_min_channel_db": psnr_min_channel(got_bgr, ref_bgr), "ssim_luma": ssim_luma(got_bgr, ref_bgr), "linf": int(d.max()), "mean_abs": float(d.mean())}


# the demo batch's geometry (demo/example/in/*.png: width x height, PNG colour type) and its pairs.txt lines
DEMO_GEOMETRY = {"in0": (680, 452, "RGBA"), "in1": (700, 528, "RGB"), "in2": (520, 600, "RGBA"), "in3": (700, 466, "RGB"), "in4": (700, 505, "RGBA"),
                 "tar0": (960, 600, "RGBA"), "tar1": (700, 393, "RGB"), "tar2": (352, 520, "RGB"), "tar3": (700, 466, "RGB"), "tar4": (800, 533, "RGBA")}
DEMO_PAIRS = [("in0", "tar0", 2.0), ("in1", "tar1", 2.0), ("in2", "tar2", 2.0), ("in3", "tar3", 2.0)] + [("in4", "tar4", b) for b in (0.0, 1.0, 2.0, 4.0, 8.0)]


def build_synthetic_demo(root, scale=1.0, gpu=0):
    """stand-in demo directory + model directory under `root` (see the module docstring); returns (demo_dir, model_dir). scale < 1 shrinks every image (tests)."""
    from PIL import Image
    sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "neural-color-transfer_amd", "python"))
    import nct, synth
    from caffemodel_io import synthetic_vgg19, write_caffemodel, write_deploy_prototxt
    demo, mdir = os.path.join(root, "example"), os.path.join(root, "model", "vgg19")
    os.makedirs(os.path.join(demo, "in")); os.makedirs(os.path.join(demo, "res")); os.makedirs(mdir)
    ws, bs = synthetic_vgg19(19)
    write_caffemodel(os.path.join(mdir, "VGG_ILSVRC_19_layers.caffemodel"), ws, bs, fmt="v1")
    write_deploy_prototxt(os.path.join(mdir, "VGG_ILSVRC_19_layers_deploy.prototxt"))
    imgs = {}
    for i, (name, (w, h, mode)) in enumerate(sorted(DEMO_GEOMETRY.items())):
        w, h = max(24, int(w * scale)), max(24, int(h * scale))
        bgr = synth.image(700 + i, h, w)
        imgs[name] = bgr
        rgb = bgr[..., ::-1].copy()
        if mode == "RGBA":      # cv::imread drops the alpha plane (does not blend): any alpha content must leave the result unchanged
            alpha = synth.image(800 + i, h, w)[..., :1]
            Image.fromarray(np.concatenate([rgb, alpha], -1), "RGBA").save(os.path.join(demo, "in", name + ".png"))
        else:
            Image.fromarray(rgb, "RGB").save(os.path.join(demo, "in", name + ".png"))
    with open(os.path.join(demo, "pairs.txt"), "w") as f:
        for s_, t_, b in DEMO_PAIRS:
            f.write("in/%s.png in/%s.png %.1f\n" % (s_, t_, b))
    with nct.Context(gpu) as c:
        c.vgg19_load_raw(ws, bs)
        for s_, t_, b in DEMO_PAIRS:
            prm = nct.Params.default(); prm.bds_weight = b
            out = c.process_pair(imgs[s_], imgs[t_], prm)
            Image.fromarray(out[..., ::-1].copy(), "RGB").save(os.path.join(demo, "res", "%s_%s_%2.2f.png" % (s_, t_, b)))
    return demo, os.path.join(root, "model")


def main():
    from PIL import Image
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default=os.environ.get("NCT_MODEL_DIR", ""))
    ap.add_argument("--demo-dir", default=os.environ.get("NCT_DEMO_DIR", "/root/reference/demo/example"))
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--min-ssim", type=float, default=0.0)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--weights", default="oxford", choices=("oxford", "synthetic"), help="synthetic = the dry run on a generated stand-in demo directory")
    ap.add_argument("--scale", type=float, default=1.0, help="[dry run] shrink the stand-in images by this factor")
    args = ap.parse_args()
    if args.weights == "synthetic":
        root = tempfile.mkdtemp(prefix="nct_demo_dry_")
        args.demo_dir, args.model_dir = build_synthetic_demo(root, args.scale, args.gpu)
        args.min_ssim = max(args.min_ssim, 0.999999)
        print(f"demo_validate: dry run on {args.demo_dir} (synthetic weights, the demo's geometry; res/ from the library binding)")
    model = os.path.join(args.model_dir, "vgg19", "VGG_ILSVRC_19_layers.caffemodel")
    if not args.model_dir or not os.path.isfile(model):
        print(f"demo_validate: {model or '<model_dir>/vgg19/VGG_ILSVRC_19_layers.caffemodel'} not found — supply the Oxford VGG19 weights (NCT_MODEL_DIR)")
        return 2
    pairs = os.path.join(args.demo_dir, "pairs.txt")
    if not os.path.isfile(pairs):
        print(f"demo_validate: {pairs} not found (NCT_DEMO_DIR)")
        return 2
    out = args.out or tempfile.mkdtemp(prefix="nct_demo_")
    r = subprocess.run([CLI, "-m", args.model_dir, "-i", args.demo_dir, "-o", out, "-g", str(args.gpu), "-inflight", "2"], capture_output=True, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:], r.stderr[-2000:])
        return 1
    report, worst = [], 1.0
    for line in open(pairs):
        f = line.split()
        if len(f) != 3:
            continue
        name = "%s_%s_%2.2f.png" % (os.path.splitext(os.path.basename(f[0]))[0], os.path.splitext(os.path.basename(f[1]))[0], float(f[2]))
        got_p, ref_p = os.path.join(out, name), os.path.join(args.demo_dir, "res", name)
        if not os.path.isfile(got_p) or not os.path.isfile(ref_p):
            report.append({"name": name, "error": "missing " + ("output" if not os.path.isfile(got_p) else "reference result")})
            continue
        got = np.asarray(Image.open(got_p).convert("RGB"))[..., ::-1]
        ref = np.asarray(Image.open(ref_p).convert("RGB"))[..., ::-1]
        if got.shape != ref.shape:
            report.append({"name": name, "error": f"shape {got.shape} vs reference {ref.shape}"})
            continue
        s = score(got, ref); s["name"] = name
        worst = min(worst, s["ssim_luma"])
        report.append(s)
        print("%-24s PSNR %6.2f dB  SSIM %.4f  L-inf %3d  mean|d| %.2f" % (name, s["psnr_min_channel_db"], s["ssim_luma"], s["linf"], s["mean_abs"]))
    if args.json:
        json.dump(report, open(args.json, "w"), indent=1)
    bad = [x for x in report if "error" in x]
    for x in bad:
        print(x["name"], "ERROR:", x["error"])
    if args.weights == "synthetic" and any(x.get("linf", 1) != 0 for x in report):
        print("dry run: the CLI's files differ from the library's results")
        return 1
    return 1 if bad or worst < args.min_ssim or len(report) != 9 else 0


if __name__ == "__main__":
    sys.exit(main())
