#!/usr/bin/env python3
"""Generates tests/golden/knn_nanoflann.npz with the reference's OWN KD-tree library: oracle/_ref/ref_nanoflann_knn (built by
`make -C oracle _ref` from oracle/ref_nanoflann_knn.cpp against /root/reference/.../ColorTransfer/Flann/nanoflann.hpp, which is
included where it lies) is run on the Lab colours of a seeded synthetic image exactly the way ColorTransfer::findSubKNNs drives it
(one cluster, k+1 = 9 results per point, Euclidean kdtree_distance). The fixture holds inputs (8-bit Lab image) and outputs
(neighbour indices and distances); tests/test_oracle_color.py::test_knn_matches_reference_nanoflann checks the oracle against it.
Needs /root/reference (this container only)."""
import os, struct, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np, oracle_bind, synth
orc = oracle_bind.load()
exe = os.path.join(REPO, "oracle", "_ref", "ref_nanoflann_knn")
subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "_ref"], check=True)
assert os.path.exists(exe), "oracle/_ref/ref_nanoflann_knn missing (is /root/reference mounted?)"
K = 8
out = {}
for name, (seed, h, w, quant) in {"smooth": (41, 48, 56, 1), "flat": (42, 40, 40, 16)}.items():
    img = synth.image(seed, h, w)
    if quant > 1:
        img = (img // quant) * quant                       # many exactly repeated colours: ties and duplicate points
    lab = orc.bgr2lab(img)
    pts = np.ascontiguousarray(lab.reshape(-1, 3).astype(np.float64) * (1.0 / 255.0))     # Mat::convertTo(CV_64F, 1/255)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<ii", pts.shape[0], K)); f.write(pts.tobytes())
        subprocess.run([exe, fin, fout], check=True)
        raw = open(fout, "rb").read()
    n, m = pts.shape[0], K + 1
    ids = np.frombuffer(raw[: n * m * 4], np.int32).reshape(n, m)
    ds = np.frombuffer(raw[n * m * 4:], np.float64).reshape(n, m)
    out[name + "_lab"] = lab; out[name + "_ids"] = ids.copy(); out[name + "_dist"] = ds.copy()
    print(name, lab.shape, "distinct colours", len(np.unique(lab.reshape(-1, 3), axis=0)), "max dist", ds.max())
np.savez_compressed(os.path.join(HERE, "knn_nanoflann.npz"), **out)
