"""CPU tests of the oracle's correspondence sub-path (N1, N2, P1, B1, B2): hand-derived known answers and
size-independent properties. The reference ships no vectors for these functions — see
oracle/README.md ("parity unpinned"); these tests pin the restatement against the arithmetic written in
GeneralizedPatchMatch.cu as read by hand."""
import numpy as np
import pytest
import synth


def X(v):
    return v & 0xFFF


def Y(v):
    return (v >> 12) & 0xFFF


def test_nnf_init_known_answers(oracle):
    # init_Ann_kernel (:527-544): bx = min(int(float(ax)/(aw-1)*(bw-1)), bw-1)
    nnf = oracle.nnf_init(4, 5, 7, 9)
    assert X(nnf[0, 0]) == 0 and Y(nnf[0, 0]) == 0
    assert X(nnf[3, 4]) == 8 and Y(nnf[3, 4]) == 6          # corners map to corners
    assert X(nnf[0, 2]) == 4                                 # 2/4*8 = 4
    assert X(nnf[0, 1]) == 2 and X(nnf[0, 3]) == 6
    assert Y(nnf[1, 0]) == 2 and Y(nnf[2, 0]) == 4           # 1/3*6 = 2, 2/3*6 = 4 (float: 0.6666667*6=4.0000002)
    # identical sizes -> identity
    n2 = oracle.nnf_init(16, 16, 16, 16)
    yy, xx = np.mgrid[0:16, 0:16]
    assert np.array_equal(X(n2), xx) and np.array_equal(Y(n2), yy)


def test_nnf_init_matches_numpy_restatement(oracle):
    for (ah, aw, bh, bw) in [(44, 44, 44, 44), (29, 43, 38, 60), (16, 16, 63, 63), (2, 2, 5, 3)]:
        nnf = oracle.nnf_init(ah, aw, bh, bw)
        ax = np.arange(aw, dtype=np.float32)
        ay = np.arange(ah, dtype=np.float32)
        bx = np.minimum((ax / np.float32(aw - 1) * np.float32(bw - 1)).astype(np.int32), bw - 1)
        by = np.minimum((ay / np.float32(ah - 1) * np.float32(bh - 1)).astype(np.int32), bh - 1)
        assert np.array_equal(X(nnf), np.broadcast_to(bx, (ah, aw)))
        assert np.array_equal(Y(nnf), np.broadcast_to(by[:, None], (ah, aw)))


@pytest.mark.parametrize("dims", [(44, 44, 88, 88), (88, 88, 175, 175), (29, 43, 57, 85), (57, 85, 113, 170), (16, 16, 32, 32)])
def test_nnf_upsample_identity_and_offsets(oracle, dims):
    ahh, awh, ah, aw = dims
    # identity half-res NNF (B same size as A) upsamples to (nearly) identity: bx = ax + (0)*ratio + 0.5 -> ax
    hy0, hx0 = np.mgrid[0:ahh, 0:awh].astype(np.uint32)
    half = (hy0 << 12) | hx0          # exact identity (init_Ann's float scaling is NOT always the identity)
    up = oracle.nnf_upsample(half, ah, aw, ah, aw)
    yy, xx = np.mgrid[0:ah, 0:aw]
    assert np.array_equal(X(up), xx) and np.array_equal(Y(up), yy)
    # a constant offset (+3,+2) at half res becomes (+3*ratio, +2*ratio) rounded, clamped to B
    hx = np.clip(X(half) + 3, 0, awh - 1)
    hy = np.clip(Y(half) + 2, 0, ahh - 1)
    up2 = oracle.nnf_upsample(((hy.astype(np.uint32) << 12) | hx.astype(np.uint32)), ah, aw, ah, aw)
    rx, ry = np.float32(aw) / np.float32(awh), np.float32(ah) / np.float32(ahh)
    cy, cx = ah // 2, aw // 2
    axh = int((cx + 0.5) / float(rx)); ayh = int((cy + 0.5) / float(ry))
    ex = int(float(np.float32(cx) + np.float32(hx[ayh, axh] - axh) * rx) + 0.5)
    ey = int(float(np.float32(cy) + np.float32(hy[ayh, axh] - ayh) * ry) + 0.5)
    assert X(up2[cy, cx]) == min(max(ex, 0), aw - 1) and Y(up2[cy, cx]) == min(max(ey, 0), ah - 1)
    assert X(up2).max() <= aw - 1 and Y(up2).max() <= ah - 1


def test_normalize_unit_norm_and_response(oracle):
    f = synth.features(1, 64, 9, 11)
    n, resp = oracle.feat_normalize(f, want_resp=True)
    nrm = np.sqrt((n.astype(np.float64) ** 2).sum(0))
    assert np.allclose(nrm, 1.0, atol=2e-6)
    ref = f / np.sqrt((f.astype(np.float64) ** 2).sum(0)).astype(np.float32)
    assert np.allclose(n, ref, rtol=3e-6, atol=1e-7)
    assert resp.min() == 0.0 and abs(resp.max() - 1.0) < 1e-6
    # zero pixel -> NaN, like the reference (no epsilon, GeneralizedPatchMatch.cu:276-277)
    f[:, 3, 4] = 0
    n2 = oracle.feat_normalize(f)
    assert np.isnan(n2[:, 3, 4]).all() and np.isfinite(n2[:, 0, 0]).all()


def _dist_numpy(a, b, ax, ay, bx, by):
    """dist_compute_single (:355-405) in float64 for cross-checking."""
    C, ah, aw = a.shape
    _, bh, bw = b.shape
    s, n = 0.0, 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if 0 <= ay + dy < ah and 0 <= ax + dx < aw and 0 <= by + dy < bh and 0 <= bx + dx < bw:
                s -= float(np.dot(a[:, ay + dy, ax + dx].astype(np.float64), b[:, by + dy, bx + dx].astype(np.float64)))
                n += 1
    return s / n


@pytest.mark.parametrize("C", [8, 64, 512])
def test_patchmatch_zero_iters_gives_init_distance(oracle, C):
    a = oracle.feat_normalize(synth.features(2, C, 7, 9))
    b = oracle.feat_normalize(synth.features(3, C, 6, 8))
    nnf0 = oracle.nnf_init(7, 9, 6, 8)
    nnf, d = oracle.patchmatch(a, b, nnf0, iters=0, rs_max=4, seed=1)
    assert np.array_equal(nnf, nnf0)
    for (ay, ax) in [(0, 0), (3, 4), (6, 8), (0, 8), (6, 0), (2, 7)]:     # all border-clipping cases
        ref = _dist_numpy(a, b, ax, ay, int(X(nnf0[ay, ax])), int(Y(nnf0[ay, ax])))
        assert abs(d[ay, ax] - ref) < 2e-6


def test_patchmatch_energy_and_validity(oracle):
    a = oracle.feat_normalize(synth.features(4, 32, 24, 28))
    b = oracle.feat_normalize(synth.features(5, 32, 20, 30))
    nnf0 = oracle.nnf_init(24, 28, 20, 30)
    _, d0 = oracle.patchmatch(a, b, nnf0, iters=0, rs_max=8, seed=7)
    prev = d0
    for it in (1, 2, 4):
        nnf, d = oracle.patchmatch(a, b, nnf0, iters=it, rs_max=8, seed=7)
        assert (d <= prev + 0).all(), "per-pixel energy must be monotone non-increasing in the iteration count"
        assert X(nnf).max() < 30 and Y(nnf).max() < 20
        # stored distance equals the distance of the stored match
        for (ay, ax) in [(0, 0), (11, 13), (23, 27)]:
            assert abs(d[ay, ax] - _dist_numpy(a, b, ax, ay, int(X(nnf[ay, ax])), int(Y(nnf[ay, ax])))) < 2e-6
        prev = d
    assert d.mean() < d0.mean() - 1e-3
    # determinism + seed sensitivity
    nnf_b, d_b = oracle.patchmatch(a, b, nnf0, iters=4, rs_max=8, seed=7)
    assert np.array_equal(nnf, nnf_b) and np.array_equal(d, d_b)
    nnf_c, _ = oracle.patchmatch(a, b, nnf0, iters=4, rs_max=8, seed=8)
    assert not np.array_equal(nnf, nnf_c)


def test_patchmatch_finds_planted_translation(oracle):
    """B is A shifted by (+5,+3): PatchMatch must recover the shift almost everywhere (interior)."""
    f = synth.features(6, 16, 40, 40, smooth=False)
    a = oracle.feat_normalize(f[:, 0:30, 0:30])
    b = oracle.feat_normalize(f[:, 3:33, 5:35])          # b(y,x) = f(y+3,x+5)  => a(y,x) matches b(y-3,x-5)
    nnf, d = oracle.patchmatch(a, b, oracle.nnf_init(30, 30, 30, 30), iters=10, rs_max=16, seed=3)
    yy, xx = np.mgrid[8:28, 8:28]
    ok = (X(nnf[8:28, 8:28]) == xx - 5) & (Y(nnf[8:28, 8:28]) == yy - 3)
    assert ok.mean() > 0.95
    assert np.allclose(d[8:28, 8:28][ok], -1.0, atol=1e-5)


def test_eval_count_matches_survey_model(oracle):
    """E = 1 + iters*(16 + R) is an upper bound (out-of-image propagation candidates are skipped), SURVEY §8d."""
    a = oracle.feat_normalize(synth.features(7, 8, 20, 20))
    b = oracle.feat_normalize(synth.features(8, 8, 20, 20))
    oracle.patchmatch(a, b, oracle.nnf_init(20, 20, 20, 20), iters=3, rs_max=4, seed=0)
    R = 3   # mags 4,2,1
    ub = 20 * 20 * (1 + 3 * (16 + R))
    assert 0.6 * ub < oracle.last_evals() <= ub


def test_feature_distance(oracle):
    a = oracle.feat_normalize(synth.features(9, 64, 5, 6))
    assert np.allclose(oracle.feature_distance(a, a), -1.0, atol=3e-6)
    b = oracle.feat_normalize(synth.features(10, 64, 5, 6))
    ref = -(a.astype(np.float64) * b).sum(0)
    assert np.allclose(oracle.feature_distance(a, b), ref, atol=3e-6)


def test_bds_vote_features_identity_maps(oracle):
    """With identity NNFs both votes average a pixel with itself: voted == input (interior and borders)."""
    f = synth.features(11, 16, 9, 10)
    iy, ix = np.mgrid[0:9, 0:10].astype(np.uint32)
    ident = (iy << 12) | ix
    out, pw = oracle.bds_vote_features(ident, ident, f, 1.0, 2.0, want_pw=True)
    assert np.allclose(out, f, rtol=2e-6)
    assert np.allclose(pw[4, 5], 9 * (1.0 / 90) + 9 * (2.0 / 90), rtol=1e-5)     # 9 coherence + 9 completeness taps
    assert np.allclose(pw[0, 0], 4 * (1.0 / 90) + 4 * (2.0 / 90), rtol=1e-5)     # corner: 4 valid taps each


def test_bds_vote_features_constant_field_and_zero_complete(oracle):
    f = np.full((8, 7, 6), 3.5, np.float32)
    ann = synth.random_nnf(1, 5, 4, 7, 6)
    bnn = synth.random_nnf(2, 7, 6, 5, 4)
    out = oracle.bds_vote_features(ann, bnn, f, 1.0, 2.0)
    assert np.allclose(out, 3.5, rtol=1e-6)
    # w_complete = 0: completeness adds exactly 0 weight and 0 value -> pure coherence average
    g = synth.features(12, 8, 7, 6)
    out0 = oracle.bds_vote_features(ann, bnn, g, 1.0, 0.0)
    ay, ax = 2, 1
    acc, n = np.zeros(8), 0
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if 0 <= ax + dx < 4 and 0 <= ay + dy < 5:
                v = ann[ay + dy, ax + dx]
                xp, yp = int(X(v)) - dx, int(Y(v)) - dy
                if 0 <= xp < 6 and 0 <= yp < 7:
                    acc += g[:, yp, xp]; n += 1
    assert np.allclose(out0[:, ay, ax], acc / n, rtol=1e-5)


def test_bds_vote_image_known_answers(oracle):
    a = synth.image(1, 6, 7)
    iy, ix = np.mgrid[0:6, 0:7].astype(np.uint32)
    ident = (iy << 12) | ix
    g = oracle.bds_vote_image(a, a, ident, ident, 1.0, 2.0)
    # identity maps reproduce the image up to the reference's TRUNCATING double->uchar store (v-eps -> v-1)
    diff = a.astype(int) - g.astype(int)
    assert diff.min() >= 0 and diff.max() <= 1
    b = synth.image(2, 5, 8)
    ann = synth.random_nnf(3, 6, 7, 5, 8)
    bnn = synth.random_nnf(4, 5, 8, 6, 7)
    g2 = oracle.bds_vote_image(a, b, ann, bnn, 1.0, 2.0)
    # brute-force restatement in python for two pixels
    wa, wb = 1.0 / (7 * 6), 2.0 / (8 * 5)
    for (ay, ax) in [(0, 0), (3, 4)]:
        acol, an = np.zeros(3, np.int64), 0
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                if 0 <= ax + dx < 7 and 0 <= ay + dy < 6:
                    v = ann[ay + dy, ax + dx]; xp, yp = int(X(v)) - dx, int(Y(v)) - dy
                    if 0 <= xp < 8 and 0 <= yp < 5:
                        acol += b[yp, xp]; an += 1
        bcol, bn = np.zeros(3, np.int64), 0
        for by in range(5):
            for bx in range(8):
                v = bnn[by, bx]; xp, yp = int(X(v)), int(Y(v))
                for dx in (-1, 0, 1):
                    for dy in (-1, 0, 1):
                        if 0 <= bx + dx < 8 and 0 <= by + dy < 5 and xp + dx == ax and yp + dy == ay:
                            bcol += b[by + dy, bx + dx]; bn += 1
        exp = ((acol * wa + bcol * wb) / (an * wa + bn * wb)).astype(np.uint8)
        assert np.array_equal(g2[ay, ax], exp)


def test_inplace_reference_schedule_fixture_reproduces(oracle):
    """oracle/orc_nnf_inplace.c — patchmatch_single under the reference's own in-place schedule (GeneralizedPatchMatch.cu:677-831, two legal
    interleavings) — reproduces the committed statistics of tests/golden/pm_inplace_band.json on the conv5_1-shaped case, improves on the
    initial field, and the product's schedule (orc_patchmatch) is not worse than either."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pm_inplace_band.json")))["cases"]["44x44x512"]
    a = oracle.feat_normalize(synth.features(fx["feature_seeds"][0], 512, 44, 44)); b = oracle.feat_normalize(synth.features(fx["feature_seeds"][1], 512, 44, 44))
    n0 = oracle.nnf_init(44, 44, 44, 44)
    means = {}
    for name, sched in (("reference_sequential", 1), ("reference_lockstep", 2)):
        nn, d = oracle.patchmatch_inplace(a, b, n0, iters=10, rs_max=fx["rs_max"], seed=fx["pm_seed"], schedule=sched)
        st = oracle.field_stats(d)
        assert np.allclose(st, fx[name]["stats"], rtol=0, atol=1e-7), (name, st, fx[name]["stats"])
        assert st[0] < fx["init"][0] - 0.01                                     # a real improvement over the scaled-identity field
        assert (nn & 0xFFF).max() < 44 and (nn >> 12).max() < 44
        means[name] = st[0]
    _, dj = oracle.patchmatch(a, b, n0, iters=10, rs_max=fx["rs_max"], seed=fx["pm_seed"])
    mj = oracle.field_stats(dj)[0]
    assert abs(mj - fx["product_jacobi"]["stats"][0]) < 1e-7
    assert mj <= min(means.values()) + 1e-4 and mj >= min(means.values()) * 1.01      # energies are negative: within 1 % below the better interleaving


def test_inplace_schedule_sequential_channel_sum_matches_definition(oracle):
    """dist_compute_single as written (:355-405): sequential `pixel_sum1 -= a*b` over channels, CHW, divided by the number of valid taps — checked against a
    float64 evaluation on a border query (4 valid taps) through a zero-iteration run."""
    a = oracle.feat_normalize(synth.features(3, 16, 5, 6)); b = oracle.feat_normalize(synth.features(4, 16, 7, 5))
    n0 = oracle.nnf_init(5, 6, 7, 5)
    _, d = oracle.patchmatch_inplace(a, b, n0, iters=0, rs_max=4, seed=1, schedule=1)
    bx, by = int(n0[0, 0] & 0xFFF), int(n0[0, 0] >> 12)
    s, cnt = 0.0, 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if 0 <= dy < 5 and 0 <= dx < 6 and 0 <= by + dy < 7 and 0 <= bx + dx < 5:
                s -= float(np.dot(a[:, dy, dx].astype(np.float64), b[:, by + dy, bx + dx].astype(np.float64))); cnt += 1
    assert cnt == 4 and abs(d[0, 0] - s / cnt) < 1e-6
