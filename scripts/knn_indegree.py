"""In-degree distribution of the kNN graph per pyramid level (synthetic bench pair): explains the S1 operator's tail latency."""
import sys, os, tempfile
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19, write_caffemodel
ctx = nct.Context(0)
ws_, bs_ = synthetic_vgg19(19)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "v.caffemodel"); write_caffemodel(p, ws_, bs_, fmt="v1"); ctx.vgg19_load_caffemodel(p)
img = synth.image(1000, 700, 700)
feats = ctx.vgg19_features(img, 5)
labels, nl = ctx.cluster_features(feats[4])
print("labels", labels.shape, "nlabels", nl, "cluster sizes", np.bincount(labels.reshape(-1), minlength=nl))
imgs = [img]
while imgs[-1].shape[0] > 44:
    nh = (imgs[-1].shape[0] + 1) // 2
    imgs.append(ctx.resize_u8c3(imgs[-1], nh, nh))
for l, im in enumerate(reversed(imgs)):
    lab = ctx.bgr2lab(im)
    ids, ws = ctx.knn_graph(lab, labels, nl, 1 << l)
    deg = np.bincount(ids.reshape(-1), minlength=ids.shape[0])
    top = np.argsort(deg)[-3:]
    print(im.shape[:2], "in-degree max", deg.max(), "p99", np.percentile(deg, 99), ">16:", (deg > 16).sum(), ">64:", (deg > 64).sum(), ">512:", (deg > 512).sum(), "top", top, deg[top],
          "self edges", (ids == np.arange(ids.shape[0])[:, None]).sum())
