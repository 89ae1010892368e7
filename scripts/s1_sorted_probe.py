"""Upper bound on what re-ordering the S1 THREADS by colour could buy (storing only the gathered vector in colour order lost): the finest-level solve on the real
kNN graph, and on the same graph after renumbering the pixels in colour order (image, graph and weights permuted together, so that thread t owns the t-th colour; the four
raster-neighbour terms then couple adjacent colours instead of adjacent pixels — a different system with the memory access pattern of the fully permuted solver, minus its four
scattered raster gathers: an optimistic bound). usage: python scripts/s1_sorted_probe.py"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, "neural-color-transfer_amd/python")
import numpy as np, nct, synth
ctx = nct.Context(0)
S = 700
src = synth.image(1000, S, S); guide = synth.image(1001, S, S)
lab = ctx.bgr2lab(src)
labels = np.zeros((44, 44), np.int32)
ids, ws = ctx.knn_graph(lab, labels, 1, 16)
n = S * S
err = np.random.default_rng(1).random((S, S)).astype(np.float32)


def morton(lab3, shift):
    v = (lab3.reshape(-1, 3).astype(np.uint32) >> shift)
    out = np.zeros(n, np.uint64)
    for b in range(8 - shift):
        for c in range(3):
            out |= ((v[:, c].astype(np.uint64) >> b) & 1) << (3 * b + c)
    return out


def run(name, src_, guide_, g, w_, e_):
    ctx.local_color_transfer(e_, src_, guide_, src_, g, w_, 4)
    t = time.perf_counter()
    for _ in range(3): ctx.local_color_transfer(e_, src_, guide_, src_, g, w_, 4)
    print("%-44s %.1f ms per call" % (name, (time.perf_counter() - t) / 3 * 1e3), flush=True)


run("real kNN graph, raster threads", src, guide, ids, ws, err)
for name, key in (("colour order: Morton of 4-unit Lab cells", morton(lab, 2) * np.uint64(n) + np.arange(n, dtype=np.uint64)),
                  ("colour order: Morton of full Lab", morton(lab, 0) * np.uint64(n) + np.arange(n, dtype=np.uint64)),
                  ("colour order: lexicographic L, a, b", (lab.reshape(-1, 3).astype(np.uint64) @ np.array([65536, 256, 1], np.uint64)) * np.uint64(n) + np.arange(n, dtype=np.uint64))):
    perm = np.argsort(key, kind="stable").astype(np.int64)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    g2 = inv[ids[perm].astype(np.int64)].astype(np.int32)
    dist = np.abs(g2.astype(np.int64) - np.arange(n)[:, None])
    run(name + " (median |t - t_nb| = %d, p90 %d)" % (np.median(dist), np.percentile(dist, 90)),
        np.ascontiguousarray(src.reshape(-1, 3)[perm].reshape(S, S, 3)), np.ascontiguousarray(guide.reshape(-1, 3)[perm].reshape(S, S, 3)),
        np.ascontiguousarray(g2), np.ascontiguousarray(ws[perm]), np.ascontiguousarray(err.reshape(-1)[perm].reshape(S, S)))
