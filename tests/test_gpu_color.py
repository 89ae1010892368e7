"""GPU parity for the colour stage (A1/OpenCV restatements, C1, K1, T1, T2, S1, U1, S2) vs the CPU oracle.
Bars: bit-exact everywhere — u8 images, labels, neighbour ids, local statistics, and the fp64 solver stages too: the operation order
of both solvers is specified (oracle/orc_color_canon.c, oracle/orc_wls_mg.c), so GPU and oracle perform the same IEEE operations."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def test_lab_conversions_exact(ctx, oracle):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (50000, 3)).astype(np.uint8)
    lab = ctx.bgr2lab(x)
    assert np.array_equal(lab, oracle.bgr2lab(x))
    allc = np.stack(np.meshgrid(np.arange(0, 256, 5), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing="ij"), -1).reshape(-1, 3).astype(np.uint8)
    assert np.array_equal(ctx.bgr2lab(allc), oracle.bgr2lab(allc))
    # Lab -> BGR, both forms of Lab2RGB_f: the float path is the same sequence of IEEE operations on both sides (-ffp-contract=off, the
    # spline table built by the same code), so the bytes are identical — on random Lab triples, on a 5-step lattice and on EVERY triple of the dark / saturated
    # corner where the two forms part (L_u8 <= 24, all a, all b)
    import nct
    dark = np.stack(np.meshgrid(np.arange(0, 25), np.arange(256), np.arange(256), indexing="ij"), -1).reshape(-1, 3).astype(np.uint8)
    for form in (nct.LAB2BGR_PIECEWISE, nct.LAB2BGR_CUBE):
        for t in (lab, allc, dark):
            assert np.array_equal(ctx.lab2bgr(t, form=form), oracle.lab2bgr(t, form=form)), form
    assert np.array_equal(ctx.lab2bgr(lab), oracle.lab2bgr(lab, form=0))                      # the default is the piecewise form
    assert not np.array_equal(ctx.lab2bgr(dark, form=0), ctx.lab2bgr(dark, form=1))


@pytest.mark.parametrize("dims", [(700, 700, 350, 350), (175, 175, 88, 88), (113, 170, 57, 85), (60, 47, 31, 23), (30, 40, 30, 40), (452, 680, 226, 340)])
def test_resize_u8_exact(ctx, oracle, dims):
    sh, sw, dh, dw = dims
    img = synth.image(5, sh, sw)
    assert np.array_equal(ctx.resize_u8c3(img, dh, dw), oracle.resize_u8c3(img, dh, dw))


@pytest.mark.parametrize("dims", [(44, 44, 700, 700), (88, 88, 175, 175), (12, 17, 100, 90), (29, 43, 452, 680)])
def test_resize_f64_exact(ctx, oracle, dims):
    sh, sw, dh, dw = dims
    src = np.random.default_rng(1).random((sh, sw, 3))
    g, o = ctx.resize_f64c3(src, dh, dw), oracle.resize_f64c3(src, dh, dw)
    assert np.array_equal(g.view(np.uint64), o.view(np.uint64))


@pytest.mark.parametrize("shape", [(512, 16, 16), (512, 44, 44), (64, 9, 11), (512, 3, 3)])
def test_kmeans_labels_exact(ctx, oracle, shape):
    f = synth.features(3, *shape) * np.float32(5.0)
    for seed in (1, 99):
        gl, gn = ctx.cluster_features(f, 10, 11, seed)
        ol, on = oracle.cluster_features(f, 10, 11, seed)
        assert gn == on and np.array_equal(gl, ol)


@pytest.mark.parametrize("shape,distinct", [((512, 44, 44), 4), ((512, 30, 30), 1), ((64, 20, 20), 12), ((512, 70, 70), 7)])
def test_kmeans_few_distinct_vectors(ctx, oracle, shape, distinct):
    """Centre selection walks the seeded permutation and skips candidates that duplicate an earlier centre (squared distance < 1e-16). Round 6: k_km_init no longer performs
    the n dependent swaps of the shuffle — it derives the first 64 entries of the SAME permutation by walking the swaps backwards — and falls back to the full shuffle only
    when more than 64 positions are consumed. Maps with a handful of distinct vectors (a flat photograph) consume hundreds: both paths, the one-label exit (fewer distinct
    vectors than K) and a map beyond the LDS list (70 x 70 > 4096 pixels) must give the oracle's labels."""
    C, h, w = shape
    rng = np.random.default_rng(17)
    protos = (rng.random((distinct, C), dtype=np.float32) + np.float32(0.05)) * np.float32(3.0)
    which = rng.integers(0, distinct, size=h * w)
    f = np.ascontiguousarray(protos[which].T.reshape(C, h, w))
    for seed in (1, 7):
        gl, gn = ctx.cluster_features(f, 10, 11, seed)
        ol, on = oracle.cluster_features(f, 10, 11, seed)
        assert gn == on and np.array_equal(gl, ol)


def test_kmeans_blobs(ctx, oracle):
    rng = np.random.default_rng(5)
    blobs = rng.standard_normal((7, 64)).astype(np.float32) * 3
    pts = np.abs(np.concatenate([blobs[i] + 0.3 * rng.standard_normal((36, 64)).astype(np.float32) for i in range(7)])) + 0.1
    f = np.ascontiguousarray(pts.T.reshape(64, 12, 21))
    gl, gn = ctx.cluster_features(f)
    ol, on = oracle.cluster_features(f)
    assert gn == on == 10 and np.array_equal(gl, ol)


# the last two cases reach the 8-unit cells (n >= 12 000) and the 2-unit cells with the two-stage start table (n >= 100 000) of k_knn_grid; a third of each image is one flat colour
@pytest.mark.parametrize("case", [(12, 14, 6, 7, 2), (32, 32, 16, 16, 2), (64, 60, 16, 15, 4), (40, 40, 5, 5, 8), (120, 110, 15, 14, 8), (320, 324, 20, 21, 16)])
def test_knn_graph(ctx, oracle, case):
    h, w, lh, lw, samples = case
    img = synth.image(9, h, w)
    img[: h // 3, : w // 3] = (90, 120, 40)              # a flat region: many exact-distance ties
    lab = oracle.bgr2lab(img)
    rng = np.random.default_rng(3)
    labels = rng.integers(0, 4, (lh, lw)).astype(np.int32)
    labels[lh // 2:, :] = 4
    gi, gw = ctx.knn_graph(lab, labels, 5, samples)
    oi, ow = oracle.knn_graph(lab, labels, 5, samples)
    assert np.array_equal(gi, oi)
    assert np.allclose(gw, ow, rtol=1e-14, atol=0)


@pytest.mark.parametrize("runs", ["0", "1"])
def test_knn_graph_both_search_forms(oracle, runs, monkeypatch):
    """K1 has two search forms that must give the same graph: every entry for itself (short runs of equal (cluster, colour) keys: the synthetic pairs) or one search per run
    (natural photographs: 17 pixels per colour in in4.png); the device picks by the entries-per-run ratio, NCT_KNN_RUNS forces one. Both on an image with a third of one
    colour AND isolated colours (the bounded-ring fallback), at the 16-lane (coarse) and one-thread (fine) sizes."""
    import nct
    monkeypatch.setenv("NCT_KNN_RUNS", runs)
    with nct.Context(0) as c:
        for (h, w, lh, lw, samples) in ((64, 60, 16, 15, 4), (200, 180, 25, 23, 8), (320, 324, 20, 21, 16)):
            img = synth.image_flat(9, h, w)
            img[h // 2:h // 2 + 3, : w // 2] = (255, 0, 255)                # colours far from everything else: the ring search gives way to the cluster pass
            img[0, 0] = (0, 255, 0)
            lab = oracle.bgr2lab(img)
            labels = (np.arange(lh * lw).reshape(lh, lw) % 4).astype(np.int32)
            gi, gw = c.knn_graph(lab, labels, 4, samples)
            oi, ow = oracle.knn_graph(lab, labels, 4, samples)
            assert np.array_equal(gi, oi), (runs, h, w)
            assert np.allclose(gw, ow, rtol=1e-14, atol=0)


def _level_case(seed, H, W, h, w, nlab_grid, samples, oracle, flat=False):
    mk = synth.image_flat if flat else synth.image
    s_full = mk(seed, H, W)
    s_lvl = oracle.resize_u8c3(s_full, h, w) if (h, w) != (H, W) else s_full
    g_lvl = oracle.resize_u8c3(mk(seed + 1, H, W), h, w)
    lh, lw = nlab_grid
    labels = (np.arange(lh * lw).reshape(lh, lw) % 3).astype(np.int32)
    ids, ws = oracle.knn_graph(oracle.bgr2lab(s_lvl), labels, 3, samples)
    err = -np.random.default_rng(seed).random((h, w)).astype(np.float32)
    return err, s_lvl, g_lvl, s_full, ids, ws


# the last case (136 896 pixels at the level) runs the forms the bandwidth-bound levels use: workgroup-shared in-edge gathers and unfused scalar steps in S1, 48x8 V-cycle tiles in S2,
# 2-unit kNN cells
# "flat" cases (round 5): images with regions of one colour (synth.image_flat) — kNN hubs with in-degrees of hundreds to ten thousands, i.e. the in-edge blocks beyond a pixel's first
# 64 that k_s1_hub sums (3 / 40 / 400 / 14 000 pixels of one colour at the level), in the fused (<= 512 workgroups) and unfused, thread-walk and shared-gather forms of the operator
@pytest.mark.parametrize("case", [(48, 48, 12, 12, (3, 3), 4, 2), (40, 56, 20, 28, (5, 7), 4, 3), (32, 32, 32, 32, (2, 2), 16, 4), (372, 368, 372, 368, (23, 23), 16, 4),
                                  (96, 96, 48, 48, (3, 3), 16, 3, "flat"), (160, 160, 80, 80, (5, 5), 16, 2, "flat"), (64, 64, 64, 64, (4, 4), 16, 4, "flat"), (372, 368, 372, 368, (23, 23), 16, 4, "flat")])
def test_local_color_transfer_stages(ctx, oracle, case):
    H, W, h, w, grid, samples, layer = case[:7]
    flat = len(case) > 7
    err, s_lvl, g_lvl, s_full, ids, ws = _level_case(20 + layer, H, W, h, w, grid, samples, oracle, flat)
    if flat:
        deg = np.bincount(ids.reshape(-1), minlength=ids.shape[0])
        assert deg.max() > 3 * 64, f"the flat case is meant to have in-edge lists of several blocks (max in-degree {deg.max()})"
    go, gs = ctx.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    oo, os_ = oracle.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    # T1: closed-form statistics — same fp64 expression order => bit exact
    assert np.array_equal(gs["ab_local"].view(np.uint64), os_["ab_local"].view(np.uint64))
    # S1: the truncated CG in the canonical operation order (oracle/orc_color_canon.c mirrors the kernels' operator order and reduction
    # trees; exp/pow are the shared IEEE-basic-op implementations): iteration cap reached on both sides, iterates bit-identical
    assert gs["cg_iters"].tolist() == os_["cg_iters"].tolist() == [50 if layer == 4 else 100] * 3
    assert np.array_equal(gs["ab_nonlocal"].view(np.uint64), os_["ab_nonlocal"].view(np.uint64))
    assert np.array_equal(gs["ab_up"].view(np.uint64), os_["ab_up"].view(np.uint64))
    assert np.array_equal(gs["roughness"], os_["roughness"])
    # S2: the multigrid-preconditioned single-reduction PCG, mirrored operation for operation by oracle/orc_wls_mg.c: same iteration
    # counts, bit-identical solution (agreement of that solver with the EXACT solve is a separate test: test_oracle_color.py and
    # test_gpu_pipeline.py::test_full_size_pair_vs_exact_s2_oracle)
    assert gs["wls_iters"].tolist() == os_["wls_iters"].tolist()
    assert np.array_equal(gs["ab_wls"].view(np.uint64), os_["ab_wls"].view(np.uint64))
    # A1: 8-bit output
    assert np.array_equal(go, oo)


@pytest.mark.parametrize("case", [(40, 56, 20, 28, (5, 7), 4, 3), (100, 141, 100, 141, (5, 7), 8, 1), (372, 368, 372, 368, (23, 23), 16, 4), (372, 368, 372, 368, (23, 23), 16, 0, "flat")])
def test_local_color_transfer_block_step(ctx, oracle, case, monkeypatch):
    """S2's block step (round 5: alternating line solves in 32 x 16 blocks on the finest level, first in the pre- and last in the post-smoother; k_mg_block / mg_block_step):
    iteration counts and solution bit-identical to the oracle's mirror at sizes that leave partial blocks at the right and bottom edges, in the 32 x 14 (>= 100 000 pixels)
    and 16 x 8 tile forms of the leg that follows the step — and fewer iterations than the cycle without it, which stays selectable (NCT_S2_LINES=0 / orc_set_mg_lines(0))
    and bit-identical to ITS mirror (the split 3 + 3 solve: test_gpu_pipeline.py::test_pair_block_step_matches_oracle)."""
    import nct
    H, W, h, w, grid, samples, layer = case[:7]
    err, s_lvl, g_lvl, s_full, ids, ws = _level_case(30 + layer, H, W, h, w, grid, samples, oracle, len(case) > 7)
    go, gs = ctx.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    oo, os_ = oracle.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    assert gs["wls_iters"].tolist() == os_["wls_iters"].tolist()
    assert np.array_equal(gs["ab_wls"].view(np.uint64), os_["ab_wls"].view(np.uint64))
    assert np.array_equal(go, oo)
    monkeypatch.setenv("NCT_S2_LINES", "0")
    oracle.l.orc_set_mg_lines(0)
    try:
        with nct.Context(0) as c:
            g0, gs0 = c.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
        o0, os0 = oracle.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer, want_stages=True)
    finally:
        oracle.l.orc_set_mg_lines(1)
    assert gs0["wls_iters"].tolist() == os0["wls_iters"].tolist()
    assert np.array_equal(gs0["ab_wls"].view(np.uint64), os0["ab_wls"].view(np.uint64))
    assert np.array_equal(g0, o0)
    assert max(os_["wls_iters"]) < max(os0["wls_iters"]), (os_["wls_iters"], os0["wls_iters"])


def test_local_color_transfer_with_nan_matching_error(ctx, oracle):
    """T2 on a matching-error map with NaNs (dead feature pixels: see test_patchmatch_dead_feature_pixels): the extremes ignore them (ColorTransfer.cpp:1311-1320 compares
    with < and >) and their confidence weight is 1e-6 (the reference's max() is the Windows macro: a NaN first operand yields the second) — on the GPU as in the oracle;
    everything downstream stays finite and bit-identical."""
    err, s_lvl, g_lvl, s_full, ids, ws = _level_case(23, 40, 56, 20, 28, (5, 7), 4, oracle)
    err = err.copy(); err[3:6, 4:9] = np.nan; err[0, 0] = np.nan; err[19, 27] = np.nan
    go, gs = ctx.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, 3, want_stages=True)
    oo, os_ = oracle.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, 3, want_stages=True)
    for k in ("ab_local", "ab_nonlocal", "ab_up", "ab_wls"):
        assert np.isfinite(os_[k]).all() and np.array_equal(gs[k].view(np.uint64), os_[k].view(np.uint64)), k
    assert np.array_equal(go, oo)
    clean = np.where(np.isnan(err), np.float32(-0.5), err)           # a map without NaNs gives another result: the NaN pixels really entered with weight 1e-6
    assert not np.array_equal(oracle.local_color_transfer(clean, s_lvl, g_lvl, s_full, ids, ws, 3), oo)
