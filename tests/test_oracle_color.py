import os
"""CPU tests of the colour-stage oracle: known answers for the OpenCV restatements, brute-force/independent checks of
k-means, kNN, the CG recurrence and the WLS solve (vs scipy sparse direct solve; MKL PARDISO fixture where available)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import synth


def test_bgr2lab_known_colours(oracle):
    """8-bit CV_BGR2Lab reference values (L*255/100, a+128, b+128) for the sRGB primaries, white, black, mid-grey."""
    cols = np.array([[255, 255, 255], [0, 0, 0], [0, 0, 255], [0, 255, 0], [255, 0, 0], [128, 128, 128]], np.uint8)
    lab = oracle.bgr2lab(cols)
    assert lab.tolist() == [[255, 128, 128], [0, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20], [137, 128, 128]]


def test_lab_roundtrip_close(oracle):
    rng = np.random.default_rng(0)
    x = rng.integers(30, 226, (20000, 3)).astype(np.uint8)
    y = oracle.lab2bgr(oracle.bgr2lab(x))
    d = np.abs(x.astype(int) - y.astype(int))
    assert d.mean() < 1.0 and np.percentile(d, 99) <= 4
    g = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)      # greys survive the round trip within 1 LSB
    assert np.abs(oracle.lab2bgr(oracle.bgr2lab(g)).astype(int) - g).max() <= 1


LAB_PIN = np.array([[0, 128, 128], [5, 128, 128], [12, 130, 120], [20, 128, 128], [21, 128, 128], [10, 160, 90], [60, 128, 128], [137, 128, 128],
                    [200, 250, 250], [40, 10, 240], [128, 128, 250], [128, 20, 128]], np.uint8)


def _lab2bgr_numpy(lab, form):
    """float64 evaluation of the two Lab2RGB_f forms, no tables (exact sRGB gamma): the independent pin of both restatements."""
    L = lab[:, 0].astype(np.float64) * 100 / 255; a = lab[:, 1].astype(np.float64) - 128; b = lab[:, 2].astype(np.float64) - 128
    if form == 1:        # plain cubes, no clipping before the gamma table
        fy = (L + 16) / 116
        y, X, Z = fy ** 3, (fy + a / 500) ** 3 * 0.950456, (fy - b / 200) ** 3 * 1.088754
    else:                # piecewise: CIE's linear branch below L* = 8 / f = 6/29, clip to [0, 1]
        fy = np.where(L <= 0.008856 * 903.3, 7.787 * (L / 903.3) + 16 / 116, (L + 16) / 116)
        y = np.where(L <= 0.008856 * 903.3, L / 903.3, fy ** 3)
        finv = lambda f: np.where(f <= 7.787 * 0.008856 + 16 / 116, (f - 16 / 116) / 7.787, f ** 3)
        X, Z = finv(a / 500 + fy) * 0.950456, finv(fy - b / 200) * 1.088754
    M = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    rgb = np.stack([X, y, Z], 1) @ M.T
    far_out = rgb[:, ::-1] < -0.02      # cube form only: far below the gamut the table's first cubic is extrapolated hundreds of steps and its tiny cubic
    #                                     term takes over (-> a saturated byte, 0 or 255, decided by the spline coefficients): not modelled here
    if form == 0:
        rgb = np.clip(rgb, 0, 1)
    srgb = np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * np.maximum(rgb, 0) ** (1 / 2.4) - 0.055)
    return np.clip(np.rint(srgb[:, ::-1] * 255), 0, 255), (far_out if form == 1 else np.zeros_like(far_out))


def test_lab2bgr_both_forms_pinned(oracle):
    """CV_Lab2BGR on 8-bit input (ColorTransfer.cpp:1469), both forms of OpenCV's Lab2RGB_f: the 2.4.x cube form (default: the reference
    links OpenCV 2.4.10) and the 3.x piecewise form. Independent float64 evaluation for dark (L_u8 <= 20), grey, saturated and out-of-gamut triples, +-1 LSB for
    the spline-interpolated gamma table, plus hand-computed values where the two forms must differ."""
    for form in (0, 1):
        got = oracle.lab2bgr(LAB_PIN, form=form).astype(int)
        exp, far_out = _lab2bgr_numpy(LAB_PIN, form)
        assert np.abs(got - exp)[~far_out].max() <= 1, (form, got.tolist(), exp.tolist())
        assert np.isin(got[far_out], (0, 255)).all()
        assert got[7].tolist() == [128, 128, 128]
    pw, cube = oracle.lab2bgr(LAB_PIN, form=0).astype(int), oracle.lab2bgr(LAB_PIN, form=1).astype(int)
    # hand-computed. L_u8 = 0 (L* = 0): piecewise Y = 0 -> (0, 0, 0); cube fY = 16/116, Y = (16/116)^3 = 0.0026241 = R = G = B on the grey axis, below the sRGB
    # toe: 12.92 * 0.0026241 * 255 = 8.65 -> 9. L_u8 = 5 (L* = 1.961): piecewise Y = 1.961/903.3 = 0.0021708 -> 12.92 * Y * 255 = 7.15 -> 7; cube
    # ((1.961+16)/116)^3 = 0.0037120 -> 12.23 -> 12. L_u8 = 21 (L* = 8.235 > 8): both ((8.235+16)/116)^3 = 0.0091190 -> 1.055 Y^(1/2.4) - 0.055 = 0.0940 -> 24.
    assert pw[0].tolist() == [0, 0, 0] and cube[0].tolist() == [9, 9, 9]
    assert pw[1].tolist() == [7, 7, 7] and cube[1].tolist() == [12, 12, 12]
    assert pw[4].tolist() == [24, 24, 24] and cube[4].tolist() == [24, 24, 24]
    assert pw[6].tolist() == cube[6].tolist() and pw[7].tolist() == cube[7].tolist()          # in-gamut greys above L* = 8: identical
    assert oracle.lab2bgr(LAB_PIN).tolist() == pw.tolist()                                       # the default is the piecewise form


def test_lab2bgr_default_form_is_the_one_the_reference_results_show(oracle):
    """Which Lab2RGB_f the reference's OpenCV ran is decided by its own artefacts: over ALL 2^24 8-bit Lab inputs the plain-cube form never outputs a pixel
    whose brightest channel is below 9 (Y >= (16/116)^3), the piecewise form does (Y = L/903.3 -> 0) — and the original binary's demo results
    (tests/golden/demo_res_dark_stats.json, statistics of demo/example/res/*.png) hold hundreds of such pixels, pure black included."""
    import json
    g = np.arange(256, dtype=np.uint8)
    lab = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    cube_floor = int(oracle.lab2bgr(lab, form=1).max(1).min())
    pw_floor = int(oracle.lab2bgr(lab, form=0).max(1).min())
    assert cube_floor == 9 and pw_floor == 0
    stats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_res_dark_stats.json")))["images"]
    below = {k: v["pixels_brightest_channel_below_9"] for k, v in stats.items()}
    assert below["in0_tar0_2.00.png"] > 1000 and stats["in0_tar0_2.00.png"]["pixels_pure_black"] > 0
    assert sum(1 for v in below.values() if v > 0) >= 7          # nearly every result image rules the cube form out


def test_resize_u8_area_and_linear(oracle):
    img = synth.image(3, 8, 12)
    half = oracle.resize_u8c3(img, 4, 6)         # exact 2x -> INTER_AREA: (a+b+c+d+2)>>2
    exp = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, exp.astype(np.uint8))
    same = oracle.resize_u8c3(img, 8, 12)
    assert np.array_equal(same, img)
    const = np.full((9, 11, 3), 77, np.uint8)    # bilinear of a constant is the constant (weights sum to 2048)
    assert np.array_equal(oracle.resize_u8c3(const, 5, 6), np.full((5, 6, 3), 77, np.uint8))
    ramp = np.tile(np.arange(0, 175, dtype=np.uint8)[None, :, None], (4, 1, 3))
    r = oracle.resize_u8c3(ramp, 4, 88)          # 175 -> 88 (ceil-pooled width): scale 1.9886
    fx = (np.arange(88) + 0.5) * (175 / 88) - 0.5
    assert np.abs(r[0, :, 0].astype(float) - np.clip(fx, 0, 174)).max() <= 1.0


def test_resize_f64_linear_exact_on_planes(oracle):
    yy, xx = np.mgrid[0:5, 0:7].astype(np.float64)
    src = np.stack([2 * xx + 1, 3 * yy - 2, xx + yy], -1)
    dst = oracle.resize_f64c3(src, 20, 28)
    sx = np.clip((np.arange(28) + 0.5) * 0.25 - 0.5, 0, 6)
    sy = np.clip((np.arange(20) + 0.5) * 0.25 - 0.5, 0, 4)
    assert np.allclose(dst[..., 0], np.broadcast_to(2 * sx + 1, (20, 28)), atol=1e-5)
    assert np.allclose(dst[..., 1], np.broadcast_to((3 * sy - 2)[:, None], (20, 28)), atol=1e-5)


def test_kmeans_invariants(oracle):
    """G9: <= K clusters, every pixel labelled once, assignment = nearest centroid of the final partition."""
    rng = np.random.default_rng(5)
    blobs = rng.standard_normal((6, 32)).astype(np.float32) * 3
    pts = np.abs(np.concatenate([blobs[i] + 0.3 * rng.standard_normal((40, 32)).astype(np.float32) for i in range(6)])) + 0.1
    f = np.ascontiguousarray(pts.T.reshape(32, 15, 16))
    labels, nl = oracle.cluster_features(f, K=10, iters=11, seed=1)
    assert nl == 10 and labels.min() >= 0 and labels.max() < 10
    l2, _ = oracle.cluster_features(f, K=10, iters=11, seed=1)
    assert np.array_equal(labels, l2)                      # deterministic
    # fewer than K points -> single label (root is a leaf, kmeans_index.h:705-710)
    l3, n3 = oracle.cluster_features(f[:, :1, :8], K=10)
    assert n3 == 1 and not l3.any()


def test_knn_matches_bruteforce(oracle):
    """G10: k smallest by (dist, id) per dilated cluster, merged — against an independent numpy brute force."""
    img = synth.image(9, 12, 14)
    lab = oracle.bgr2lab(img)
    labels = np.zeros((6, 7), np.int32); labels[:, 4:] = 1; labels[4:, :] = 2
    ids, ws = oracle.knn_graph(lab, labels, 3, samples=2)
    labd = lab.reshape(-1, 3).astype(np.float64) * (1.0 / 255.0)
    # membership by dilation
    mem = np.zeros((3, 6, 7), bool)
    for y in range(6):
        for x in range(7):
            l0 = labels[y, x]; mem[l0, y, x] = True
            for dy, dx in ((0, 1), (0, -1), (1, 0), (-1, 0)):
                yy, xx = y + dy, x + dx
                if 0 <= yy < 6 and 0 <= xx < 7 and labels[yy, xx] != l0:
                    mem[l0, yy, xx] = True
    pix_mem = np.repeat(np.repeat(mem, 2, 1), 2, 2)[:, :12, :14].reshape(3, -1)
    for i in (0, 37, 100, 167):
        cand = {}
        for l in range(3):
            if not pix_mem[l, i]:
                continue
            members = np.flatnonzero(pix_mem[l])
            d = np.sqrt(((labd[members] - labd[i]) ** 2).sum(1))
            order = sorted(zip(d, members))[:9]
            nons = [(dd, j) for dd, j in order if j != i][:8]
            for dd, j in nons:
                cand[j] = dd
        best = sorted((dd, j) for j, dd in cand.items())[:8]
        assert [j for _, j in best] == ids[i].tolist()
        assert np.allclose(ws[i], [np.exp(1 - dd / 3) for dd, _ in best])


def _level_inputs(seed, h, w, H, W):
    s_full = synth.image(seed, H, W)
    r_full = synth.image(seed + 1, H, W)
    return s_full, r_full


def test_wls_matches_scipy_direct(oracle):
    """S2: banded-Cholesky path and PCG path vs an independent scipy sparse LU of the same system."""
    oracle._decl_color()
    H, W = 14, 17
    rng = np.random.default_rng(2)
    lab = np.ascontiguousarray(oracle.bgr2lab(synth.image(4, H, W)).astype(np.float64) / 255.0)
    rough = np.where(rng.random(H * W) < 0.2, 1e-6, 1.0)
    a0 = rng.random((H * W, 3)); b0 = rng.random((H * W, 3)) - 0.5
    diag = np.empty(H * W); wx = np.empty(H * W); wy = np.empty(H * W)
    oracle.l.orc_wls_system(lab.reshape(-1), H, W, 0.37, 1.2, rough, diag, wx, wy)
    n = H * W
    M = sp.lil_matrix((n, n))
    for i in range(n):
        M[i, i] = diag[i]
        if i % W + 1 < W: M[i, i + 1] = -wx[i]; M[i + 1, i] = -wx[i]
        if i + W < n: M[i, i + W] = -wy[i]; M[i + W, i] = -wy[i]
    M = M.tocsc()
    # row sums: diag - offdiag = roughness (graph Laplacian + data term)
    assert np.allclose(np.asarray(M.sum(1)).ravel(), rough, rtol=1e-9, atol=1e-9)
    lu = spla.splu(M)
    for force in (0, 1):
        a, b = a0.copy(), b0.copy()
        oracle.l.orc_wls_solve(a.reshape(-1), b.reshape(-1), lab.reshape(-1), H, W, 0.37, 1.2, rough, force)
        for c in range(3):
            assert np.allclose(a[:, c], lu.solve(rough * a0[:, c]), rtol=1e-8, atol=1e-10)
            assert np.allclose(b[:, c], lu.solve(rough * b0[:, c]), rtol=1e-8, atol=1e-10)


def test_wls_matches_mkl_pardiso_fixture(oracle):
    """The reference's actual solver (MKL PARDISO, mtype 2, the iparm of SparseSolver_CPU.cpp:135-160) solved this system
    in the build container; the fixture holds its solution (tests/golden/gen_wls_pardiso.py)."""
    import os
    p = os.path.join(os.path.dirname(__file__), "golden", "wls_pardiso.npz")
    if not os.path.exists(p):
        pytest.skip("fixture not generated")
    oracle._decl_color()
    z = np.load(p)
    H, W = int(z["H"]), int(z["W"])
    a, b = z["a0"].copy(), z["b0"].copy()
    oracle.l.orc_wls_solve(a.reshape(-1), b.reshape(-1), np.ascontiguousarray(z["lab"]).reshape(-1), H, W, float(z["lamda"]), float(z["alpha"]),
                           np.ascontiguousarray(z["rough"]), 0)
    assert np.allclose(a, z["a_pardiso"], rtol=1e-9, atol=1e-11) and np.allclose(b, z["b_pardiso"], rtol=1e-9, atol=1e-11)


def test_level_transfer_stages(oracle):
    """Composed level: stage invariants (initial a = sigma ratio, CG iteration cap is what stops S1, roughness in
    {1e-6,1}, WLS residual small, identity transfer when G == S)."""
    H = W = 24; h = w = 12
    s_full = synth.image(11, H, W)
    s_lvl = oracle.resize_u8c3(s_full, h, w)
    g_lvl = oracle.resize_u8c3(synth.image(12, H, W), h, w)
    labels = np.zeros((3, 3), np.int32); labels[:, 2] = 1
    ids, ws = oracle.knn_graph(oracle.bgr2lab(s_lvl), labels, 2, samples=4)
    err = -np.random.default_rng(1).random((h, w)).astype(np.float32)
    out, st = oracle.local_color_transfer(err, s_lvl, g_lvl, s_full, ids, ws, layer=3, want_stages=True)
    assert st["cg_iters"].tolist() == [100, 100, 100]             # un-preconditioned CG never reaches 1e-12: the cap stops it
    assert set(np.unique(st["roughness"]).tolist()) <= {1e-6, 1.0}
    assert np.isfinite(st["ab_wls"]).all() and out.shape == (H, W, 3)
    # finest level (layer 4) caps at 50 and skips the resize
    idsf, wsf = oracle.knn_graph(oracle.bgr2lab(s_full), np.zeros((2, 2), np.int32), 1, samples=16)
    errf = -np.random.default_rng(2).random((H, W)).astype(np.float32)
    outf, stf = oracle.local_color_transfer(errf, s_full, synth.image(13, H, W), s_full, idsf, wsf, layer=4, want_stages=True)
    assert stf["cg_iters"].tolist() == [50, 50, 50]
    assert np.array_equal(stf["ab_up"], stf["ab_nonlocal"])
    # G == S: a = sigma/(sigma+eps) < 1 initially; the result must stay close to S (sanity of the whole chain)
    out_id = oracle.local_color_transfer(errf, s_full, s_full, s_full, idsf, wsf, layer=4)
    assert np.abs(out_id.astype(int) - s_full.astype(int)).mean() < 6.0


def _s1_inputs(oracle, seed=3, h=12, w=12):
    s = oracle.resize_u8c3(synth.image(seed, 48, 48), h, w)
    g = oracle.resize_u8c3(synth.image(seed + 1, 48, 48), h, w)
    labels = (np.arange(9).reshape(3, 3) % 3).astype(np.int32)
    ids, ws = oracle.knn_graph(oracle.bgr2lab(s), labels, 3, samples=4)
    err = -np.random.default_rng(seed).random((h, w)).astype(np.float32)
    return err, s, g, ids, ws


def test_truncated_cg_is_chaotic(oracle):
    """the reference's S1 (un-preconditioned CG stopped at its iteration cap) amplifies a 1e-15 relative
    perturbation of ONE kNN weight into O(1e-3..1e-2) coefficient changes — bit-level agreement needs identical arithmetic."""
    err, s, g, ids, ws = _s1_inputs(oracle)
    full = synth.image(3, 48, 48)
    _, st1 = oracle.local_color_transfer(err, s, g, full, ids, ws, layer=2, want_stages=True)
    ws2 = ws.copy(); ws2[5, 3] *= (1 + 1e-15)
    _, st2 = oracle.local_color_transfer(err, s, g, full, ids, ws2, layer=2, want_stages=True)
    d = np.abs(st1["ab_nonlocal"] - st2["ab_nonlocal"]).max()
    assert d > 1e-6, f"expected chaotic amplification, got {d}"
    assert np.array_equal(st1["ab_local"], st2["ab_local"])


def test_canonical_cg_matches_explicit_for_few_iterations(oracle):
    """The canonical-order matrix-free operator (orc_color_canon.c) and the literally assembled A^T(Ax) (orc_color.c) are the same
    linear operator: before the chaos sets in (few iterations) both CG variants agree to rounding."""
    import ctypes as C
    oracle._decl_color()
    err, s, g, ids, ws = _s1_inputs(oracle)
    h = w = 12; n = h * w
    lab_s = oracle.bgr2lab(s).reshape(-1, 3).astype(np.float64) * (1.0 / 255.0)
    lab_g = oracle.bgr2lab(g).reshape(-1, 3).astype(np.float64) * (1.0 / 255.0)
    _, st = oracle.local_color_transfer(err, s, g, synth.image(3, 48, 48), ids, ws, layer=2, want_stages=True)
    a0, b0 = st["ab_local"][0].copy(), st["ab_local"][1].copy()
    e = err.reshape(-1).astype(np.float64)
    wgt = np.maximum(1.0 - (e - e.min()) / (e.max() - e.min()), 1e-6)
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"); i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
    sig = [f64, f64, f64, f64, f64, i32, f64] + [C.c_int] * 4 + [C.c_float] * 3 + [C.c_double, C.c_double, i32, C.c_int]
    oracle.l.orc_nonlocal_solve.argtypes = sig
    oracle.l.orc_nonlocal_solve_explicit.argtypes = sig
    for maxit, tol in ((1, 1e-12), (3, 1e-10), (8, 1e-7)):
        res = []
        for fn in (oracle.l.orc_nonlocal_solve, oracle.l.orc_nonlocal_solve_explicit):
            a, b = a0.copy(), b0.copy(); it = np.zeros(3, np.int32)
            fn(a.reshape(-1), b.reshape(-1), lab_s.reshape(-1), lab_g.reshape(-1), wgt, ids.reshape(-1), ws.reshape(-1), 8, h, w, 2,
               0.125, 1.2, 16.0, 2.0, 8.0, it, maxit)
            assert it.tolist() == [maxit] * 3
            res.append((a, b))
        assert np.allclose(res[0][0], res[1][0], rtol=tol, atol=tol) and np.allclose(res[0][1], res[1][1], rtol=tol, atol=tol)
        assert not np.array_equal(res[0][0], a0)       # the solve moved the coefficients


def test_canonical_wls_matches_exact_solve(oracle):
    """S2: the canonical-order MG-PCG (mirror of the product, stopped at 1e-7 relative residual) agrees with the exact
    banded-Cholesky solve to ~1e-5 in the coefficients, i.e. far below one 8-bit quantisation step of the output."""
    err, s, g, ids, ws = _s1_inputs(oracle)
    full = synth.image(3, 48, 48)
    o1, s1 = oracle.local_color_transfer(err, s, g, full, ids, ws, layer=2, want_stages=True, s2_exact=False)
    o2, s2 = oracle.local_color_transfer(err, s, g, full, ids, ws, layer=2, want_stages=True, s2_exact=True)
    assert np.array_equal(s1["ab_up"], s2["ab_up"])
    assert np.allclose(s1["ab_wls"], s2["ab_wls"], rtol=2e-5, atol=2e-6)
    d = np.abs(o1.astype(int) - o2.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_knn_matches_reference_nanoflann(oracle):
    """K1 pinned by the reference's own KD-tree library: tests/golden/knn_nanoflann.npz holds what the vendored nanoflann.hpp
    (driven as in ColorTransfer::findSubKNNs: one cluster, k+1 results, Euclidean kdtree_distance) returns for two seeded Lab images
    — one smooth, one quantised to force exact ties and duplicate points (generator: tests/golden/gen_knn_nanoflann.py, driver
    oracle/ref_nanoflann_knn.cpp). The oracle's (dist, id) brute force must give the same distances bit for bit, and the same
    neighbour wherever the distance is not tied (tie order is the KD-tree's traversal order in the reference; ours is by id)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_nanoflann.npz"))
    K = 8
    for name in ("smooth", "flat"):
        lab, rid, rd = g[name + "_lab"], g[name + "_ids"], g[name + "_dist"]
        h, w = lab.shape[:2]
        n = h * w
        ids, ws = oracle.knn_graph(lab, np.zeros((2, 2), np.int32), 1, samples=max(h, w))      # one cluster, every pixel in it
        exact_ids = checked = 0
        for i in range(n):
            # findSubKNNs: drop the query itself from the k+1 results, keep the first k
            keep = [(rd[i, t], rid[i, t]) for t in range(K + 1) if rid[i, t] >= 0 and rid[i, t] != i][:K]
            ref_d = np.array([d for d, _ in keep])
            ref_w = np.exp(1.0 - ref_d / 3.0)
            assert len(keep) == K
            # weights come from orc_exp (IEEE-basic-op exp, <= 1 ulp from libm); distances are compared through them
            assert np.allclose(ws[i], ref_w, rtol=4e-16, atol=0), (name, i)
            for t in range(K):
                d = keep[t][0]
                tied = (t > 0 and keep[t - 1][0] == d) or (t + 1 < K and keep[t + 1][0] == d) or (t == K - 1 and rd[i, K] == d)
                if not tied:
                    checked += 1
                    exact_ids += int(ids[i, t] == keep[t][1])
        assert exact_ids == checked and checked > (n if name == "smooth" else 100), (name, exact_ids, checked)


def _wls_case(oracle, H, W, lam_factor, seed=3, rough_frac=0.1):
    lab = oracle.bgr2lab(synth.image(seed, H, W)).astype(np.float64) / 255.0
    rough = np.where(np.random.default_rng(1).random(H * W) < rough_frac, 1e-6, 1.0)
    return lab, rough, 0.024 * lam_factor


@pytest.mark.parametrize("dims", [(64, 64, 253.0), (45, 71, 16.0), (97, 33, 4.0)])
def test_mg_preconditioner_is_symmetric_positive_definite(oracle, dims):
    """S2 (round 4): PCG needs the preconditioner to be ONE symmetric positive definite linear map. The V-cycle of oracle/orc_wls_mg.c (= k_wls_mg.hip) is built so — restriction
    = transpose of the operator-dependent prolongation (the same fp32 P columns on both sides), post-smoother = adjoint of the pre-smoother (Chebyshev-weighted Jacobi sweeps of one
    operator commute), symmetric Galerkin stencils (forward couplings stored once), a symmetric coarsest solve — and this checks it from the outside on random vectors, independently
    of any mirror: <M^-1 a, b> = <a, M^-1 b> to fp32 accuracy, <M^-1 a, a> > 0, linearity; odd sizes included (the last row / column has no coarse partner)."""
    H, W, lf = dims
    lab, rough, lam = _wls_case(oracle, H, W, lf)
    rng = np.random.default_rng(7)
    n = H * W
    a = rng.standard_normal((n, 6)); b = rng.standard_normal((n, 6))
    smooth = np.add.outer(np.sin(np.arange(H) / 9.0), np.cos(np.arange(W) / 7.0)).reshape(n, 1) * np.ones((1, 6))     # a smooth vector exercises the coarse levels
    z, nl = oracle.wls_vcycle(lab, lam, 1.2, rough, np.stack([a, b, smooth, a + 2.0 * b]))
    assert nl >= 4
    za, zb, zs, zab = z
    for q in range(6):
        lhs, rhs = float(za[:, q] @ b[:, q]), float(a[:, q] @ zb[:, q])
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), np.linalg.norm(za[:, q]) * np.linalg.norm(b[:, q]) * 1e-2), (q, lhs, rhs)
        assert float(za[:, q] @ a[:, q]) > 0 and float(zs[:, q] @ smooth[:, q]) > 0
        lin = np.abs(zab[:, q] - (za[:, q] + 2.0 * zb[:, q])).max()
        assert lin <= 2e-5 * np.abs(zab[:, q]).max(), (q, lin)
    # the six right-hand sides share the operator and nothing else: a column of zeros stays zero
    a0 = a.copy(); a0[:, 2] = 0.0
    z0, _ = oracle.wls_vcycle(lab, lam, 1.2, rough, a0[None])
    assert np.all(z0[0][:, 2] == 0.0) and np.array_equal(z0[0][:, 0], za[:, 0])


def test_mg_galerkin_levels_are_diagonally_sane(oracle):
    """The Galerkin coarse operators P^T A P of the vertex-centred hierarchy: level sizes halve (rounding up) down to <= 64 unknowns; every diagonal is positive; level 0 is an
    M-matrix whose smallest row sum is the smallest data term; couplings of the wrong sign may appear on coarse levels (the operator-dependent P does not sum to one where the data
    term matters, so Galerkin rows need not be diagonally dominant), and the safe smoother diagonal max(d, (|d| + sum |w|) / 2) then exceeds d (positive
    definiteness itself is what test_mg_preconditioner_is_symmetric_positive_definite checks)."""
    H, W = 83, 120
    lab, rough, lam = _wls_case(oracle, H, W, 63.0)
    st = oracle.wls_hierarchy_stats(lab, lam, 1.2, rough)
    sizes = [int(v) for v in st[:, 0]]
    h, w, exp = H, W, []
    while True:
        exp.append(h * w)
        if h * w <= 64 or (h <= 8 and w <= 8):
            break
        h, w = (h + 1) // 2, (w + 1) // 2
    assert sizes == exp
    assert (st[:, 1] > 0).all()                                    # diagonals
    assert abs(st[0, 2] - rough.min()) <= 1e-9 * st[0, 1]           # level 0: row sum = data term
    assert st[0, 3] == 0 and st[0, 4] == 1.0                        # level 0: M-matrix, safe diagonal = diagonal
    assert (st[:, 4] >= 1.0).all() and (st[1:, 4] > 1.0).any()     # some coarse row is not diagonally dominant: the safe diagonal is in use


def test_mg_smoother_degree_changes_iterations_not_the_solution(oracle):
    """S2 preconditioner: the V-cycle's Chebyshev-weighted Jacobi smoother runs MG_NS sweeps per leg (k_wls_mg.hip / orc_wls_mg.c; the product's default is 3). The degree
    only changes how fast PCG converges — at the solver's tolerance the solutions of the 2-, 3- and 4-sweep cycles agree far below an 8-bit step and the 8-bit results are equal;
    more sweeps need fewer iterations."""
    err, s, g, ids, ws = _s1_inputs(oracle, h=24, w=24)
    full = synth.image(3, 96, 96)
    assert oracle.l.orc_get_mg_smoother() == 3
    res = {}
    try:
        for ns in (2, 3, 4):
            oracle.l.orc_set_mg_smoother(ns)
            res[ns] = oracle.local_color_transfer(err, s, g, full, ids, ws, layer=2, want_stages=True)
    finally:
        oracle.l.orc_set_mg_smoother(3)
    it = {ns: int(res[ns][1]["wls_iters"].max()) for ns in res}
    assert it[2] > it[3] >= it[4] > 3, it
    for ns in (2, 4):
        assert np.allclose(res[ns][1]["ab_wls"], res[3][1]["ab_wls"], rtol=0, atol=5e-6)
        assert np.array_equal(res[ns][0], res[3][0])


def test_oracle_pair_reproduces_the_committed_tiny_fixture(oracle):
    """Regression pin of the ORACLE itself (CPU only): the whole L=5->1 loop on the 64x56 / 48x64 pair of tests/golden/pair_exact_tiny.npz (gen_pair700_exact.py tiny) must
    reproduce the committed CRC of every level's intermediate result — with the canonical-order S2 solver and with the exact S2 solve, which agree byte for byte. A change of
    any oracle stage (the S2 hierarchy of round 4 included) that moved a single output byte would show here without a GPU; the full-size fixtures of the same generator are what
    the GPU path is held to."""
    import os, zlib
    from caffemodel_io import synthetic_vgg19
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pair_exact_tiny.npz"))
    sh, sw, rh, rw = (int(v) for v in g["shape"])
    ws, bs = synthetic_vgg19(19)
    src, ref = synth.image(1000, sh, sw), synth.image(1001, rh, rw)
    out, lv = oracle.process_pair(src, ref, ws, bs, want_levels=True)
    assert [zlib.crc32(lv[l].tobytes()) for l in range(5)] == [int(v) for v in g["level_crc_canonical"]]
    assert zlib.crc32(out.tobytes()) == int(g["crc_canonical"]) == int(g["crc_exact"]) and g["idx"].size == 0
    out_x, lv_x = oracle.process_pair(src, ref, ws, bs, want_levels=True, s2_exact=True)
    assert np.array_equal(out_x, out) and [zlib.crc32(lv_x[l].tobytes()) for l in range(5)] == [int(v) for v in g["level_crc_exact"]]


def test_s1_literal_recurrence_band_end_to_end(oracle):
    """the canonical S1 (what the GPU reproduces) against the LITERAL recurrence of SparseSolver_GPU.cu:132-159 on the assembled A
    (orc_set_s1_form(1)), everything else identical, end to end. tests/golden/s1_band.json holds the measured band of the 256x256 (and 700x700) pair
    (generator tests/golden/gen_s1_band.py); this test re-derives a small pair from scratch: the two forms differ (S1 is chaotic), and by no more than the
    recorded band says they may — tens of dB below bit identity, far above the 17 dB the transfer itself moves the image."""
    import json, os
    from caffemodel_io import synthetic_vgg19
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "s1_band.json")))
    c = fx["cases"]["pair256"]
    assert 40.0 < c["final"]["psnr_min_channel_db"] < 60.0 and c["final"]["linf"] <= 32
    assert c["levels_coarse_to_fine"][0]["psnr_min_channel_db"] > c["levels_coarse_to_fine"][4]["psnr_min_channel_db"]        # the distance accumulates over the levels
    assert c["colour_change_of_the_transfer"]["psnr_min_channel_db"] < 25.0
    ws, bs = synthetic_vgg19(19)
    src, ref = synth.image(1000, 64, 56), synth.image(1001, 48, 64)
    try:
        oracle.set_s1_form(1); lit = oracle.process_pair(src, ref, ws, bs)
    finally:
        oracle.set_s1_form(0)
    can = oracle.process_pair(src, ref, ws, bs)
    d = np.abs(lit.astype(int) - can.astype(int))
    mse = max(float((d.astype(np.float64) ** 2).mean()), 1e-12)
    psnr = 10 * np.log10(255.0 ** 2 / mse)
    assert d.max() > 0, "the two recurrences are expected to differ end to end (S1 is chaotic)"
    assert psnr > 38.0 and d.max() <= 48, (psnr, int(d.max()))
