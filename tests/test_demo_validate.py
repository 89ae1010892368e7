"""the real-weights validation harness. The scoring code is tested on CPU; the end-to-end run needs the Oxford VGG19
caffemodel ($NCT_MODEL_DIR/vgg19/VGG_ILSVRC_19_layers.caffemodel) and the reference's demo directory ($NCT_DEMO_DIR), neither of which
exists on the GPU box, so that test skips itself unless both are supplied."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "scripts"))


def test_scores():
    import demo_validate as dv
    import synth
    a = synth.image(3, 64, 80)
    s = dv.score(a, a)
    assert s["linf"] == 0 and s["psnr_min_channel_db"] == 99.0 and abs(s["ssim_luma"] - 1.0) < 1e-12
    b = a.copy(); b[..., 0] = np.clip(b[..., 0].astype(int) + 10, 0, 255)
    s = dv.score(a, b)
    assert 27.5 < s["psnr_min_channel_db"] < 29.0 and s["linf"] == 10 and 0.9 < s["ssim_luma"] < 1.0      # 20 log10(255/10) = 28.13 dB
    rng = np.random.default_rng(0)
    s = dv.score(a, rng.integers(0, 256, a.shape).astype(np.uint8))
    assert s["ssim_luma"] < 0.2 and s["psnr_min_channel_db"] < 12


def test_missing_model_is_reported(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "demo_validate.py"), "--model-dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 2 and "VGG_ILSVRC_19_layers.caffemodel" in r.stdout


@pytest.mark.gpu
def test_demo_batch_against_reference_results(tmp_path):
    md, dd = os.environ.get("NCT_MODEL_DIR", ""), os.environ.get("NCT_DEMO_DIR", "/root/reference/demo/example")
    if not os.path.isfile(os.path.join(md, "vgg19", "VGG_ILSVRC_19_layers.caffemodel")) or not os.path.isfile(os.path.join(dd, "pairs.txt")):
        pytest.skip("Oxford VGG19 caffemodel ($NCT_MODEL_DIR) or the reference demo directory ($NCT_DEMO_DIR) not supplied")
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "demo_validate.py"), "--model-dir", md, "--demo-dir", dd, "--out", str(tmp_path / "out"),
                        "--json", str(tmp_path / "report.json"), "--min-ssim", "0.5"], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_demo_dry_run_on_the_demo_geometry(tmp_path):
    """SURVEY §8(f)-2 readiness: the whole harness on a generated stand-in of the reference's demo batch — the demo's image sizes and PNG colour types (five RGBA
    inputs: alpha must be dropped like cv::imread does), `in/` sub-directory, the 9-line pairs.txt with the BDS sweep 0/1/2/4/8, `<src>_<ref>_<bds>.png` naming —
    with synthetic weights; the CLI's files must equal the library's results (L-inf 0). The real-weights run stays gated on $NCT_MODEL_DIR."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "demo_validate.py"), "--weights", "synthetic", "--scale", "0.5", "--out", str(tmp_path / "out"),
                        "--json", str(tmp_path / "report.json")], capture_output=True, text=True)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    import json
    rep = json.load(open(tmp_path / "report.json"))
    assert len(rep) == 9 and all(x["linf"] == 0 for x in rep)
    assert sorted(x["name"] for x in rep)[:2] == ["in0_tar0_2.00.png", "in1_tar1_2.00.png"] and "in4_tar4_8.00.png" in [x["name"] for x in rep]


def test_demo_geometry_matches_the_reference_demo():
    """the stand-in's sizes / colour types are the demo's (checked where the reference is mounted)"""
    import demo_validate as dv
    d = "/root/reference/demo/example/in"
    if not os.path.isdir(d):
        pytest.skip("reference demo not mounted")
    from PIL import Image
    for name, (w, h, mode) in dv.DEMO_GEOMETRY.items():
        im = Image.open(os.path.join(d, name + ".png"))
        assert im.size == (w, h) and im.mode == mode, name
    lines = [l.split() for l in open("/root/reference/demo/example/pairs.txt") if l.strip()]
    assert [("in/%s.png" % a, "in/%s.png" % b, c) for a, b, c in dv.DEMO_PAIRS] == [(l[0], l[1], float(l[2])) for l in lines]
