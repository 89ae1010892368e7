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

usage: python scripts/demo_validate.py [--model-dir D] [--demo-dir D] [--out D] [--json report.json] [--min-ssim 0.0]"""
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


def main():
    from PIL import Image
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default=os.environ.get("NCT_MODEL_DIR", ""))
    ap.add_argument("--demo-dir", default=os.environ.get("NCT_DEMO_DIR", "/root/reference/demo/example"))
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--min-ssim", type=float, default=0.0)
    ap.add_argument("--gpu", type=int, default=0)
    args = ap.parse_args()
    model = os.path.join(args.model_dir, "vgg19", "VGG_ILSVRC_19_layers.caffemodel")
    if not args.model_dir or not os.path.isfile(model):
        print(f"demo_validate: {model or '<model_dir>/vgg19/VGG_ILSVRC_19_layers.caffemodel'} not found — supply the Oxford VGG19 weights (NCT_MODEL_DIR)")
        return 2
    pairs = os.path.join(args.demo_dir, "pairs.txt")
    if not os.path.isfile(pairs):
        print(f"demo_validate: {pairs} not found (NCT_DEMO_DIR)")
        return 2
    out = args.out or tempfile.mkdtemp(prefix="nct_demo_")
    r = subprocess.run([CLI, "-m", args.model_dir, "-i", args.demo_dir, "-o", out, "-g", str(args.gpu)], capture_output=True, text=True)
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
    return 1 if bad or worst < args.min_ssim else 0


if __name__ == "__main__":
    sys.exit(main())
