"""Generates tests/golden/demo_res_dark_stats.json: darkest-pixel statistics of the reference's own result images
(/root/reference/demo/example/res/*.png — the only artefacts of the original binary). They arbitrate which form of OpenCV's
Lab2RGB_f the reference's CV_Lab2BGR (ColorTransfer.cpp:1469) ran: the plain-cube form cannot output a pixel whose brightest
channel is below 9 for ANY 8-bit Lab input (tests/test_oracle_color.py evaluates all 2^24), the piecewise form can.
Only statistics are stored (counts, the darkest triples); runs only where /root/reference is mounted.
    python tests/golden/gen_demo_res_dark_stats.py"""
import glob
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
out = {"generator": "tests/golden/gen_demo_res_dark_stats.py", "source": "demo/example/res/*.png of the reference repository", "images": {}}
for f in sorted(glob.glob("/root/reference/demo/example/res/*.png")):
    im = np.asarray(Image.open(f).convert("RGB")).astype(int)
    mx = im.max(-1)
    order = np.argsort(mx.ravel())[:4]
    out["images"][os.path.basename(f)] = {"height": im.shape[0], "width": im.shape[1], "min_of_brightest_channel": int(mx.min()),
                                          "pixels_brightest_channel_below_9": int((mx < 9).sum()), "pixels_brightest_channel_below_5": int((mx < 5).sum()),
                                          "pixels_pure_black": int((mx == 0).sum()), "darkest_rgb": [[int(v) for v in im.reshape(-1, 3)[i]] for i in order]}
json.dump(out, open(os.path.join(HERE, "demo_res_dark_stats.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
