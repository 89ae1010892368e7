"""Does a CLUSTER-MAJOR pixel order confine the S1 operator's gathers to an XCD's L2? The kNN edges of a pixel stay inside the (dilated) k-means clusters it belongs to, and the
clusters are spatial blobs: if threads and vectors are ordered (cluster, raster) and every XCD takes a contiguous range of that order (NCT_S1_XCD=1), an XCD gathers from ~1/8 of the
23.5 MB vector. Probe (like s1_sorted_probe.py): the finest-level solve of the bench pair's real graph, and of the same graph with image, graph and weights renumbered cluster-major
(a different system — the raster terms then couple neighbours of the new order — with the permuted solver's gather pattern, minus its scattered raster gathers: an optimistic bound).
usage: [NCT_S1_XCD=1] python scripts/s1_cluster_probe.py"""
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
from caffemodel_io import synthetic_vgg19
ws_, bs_ = synthetic_vgg19(19)
ctx = nct.Context(0); ctx.vgg19_load_raw(ws_, bs_)
S = 700
src = synth.image(1000, S, S); ref = synth.image(1001, S, S)
ctx.pair_upload(src, ref)
lv = ctx.pair_run_levels(src.shape, ref.shape, want_color=True)
labels = lv["labels"]; nl = int(labels.max()) + 1
lab = ctx.bgr2lab(src)
ids, ws = ctx.knn_graph(lab, labels, nl, 16)
n = S * S
guide = lv["guide"][4]; err = lv["err"][4]
own = np.repeat(np.repeat(labels, 16, 0), 16, 1)[:S, :S].reshape(-1)
print("cluster sizes", np.bincount(own, minlength=nl), "share of kNN edges inside the pixel's own cluster: %.3f" % (own[ids.reshape(-1)] == np.repeat(own, 8)).mean())


def run(name, src_, guide_, g, w_, e_):
    ctx.local_color_transfer(e_, src_, guide_, src_, g, w_, 4)
    t = time.perf_counter()
    for _ in range(5): ctx.local_color_transfer(e_, src_, guide_, src_, g, w_, 4)
    print("%-70s %.2f ms per call (NCT_S1_XCD=%s)" % (name, (time.perf_counter() - t) / 5 * 1e3, os.environ.get("NCT_S1_XCD", "0")), flush=True)


run("real kNN graph, raster order", src, guide, ids, ws, err)
for name, key in (("cluster-major (own cluster, raster)", own.astype(np.uint64) * np.uint64(n) + np.arange(n, dtype=np.uint64)),):
    perm = np.argsort(key, kind="stable").astype(np.int64)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    g2 = inv[ids[perm].astype(np.int64)].astype(np.int32)
    run(name, np.ascontiguousarray(src.reshape(-1, 3)[perm].reshape(S, S, 3)), np.ascontiguousarray(guide.reshape(-1, 3)[perm].reshape(S, S, 3)),
        np.ascontiguousarray(g2), np.ascontiguousarray(ws[perm]), np.ascontiguousarray(err.reshape(-1)[perm].reshape(S, S)))
