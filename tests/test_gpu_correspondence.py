"""GPU parity for the correspondence sub-path (N1, N2, P1, B1, B2): libnct (HIP, through the C ABI) vs the CPU oracle
on identical seeded inputs. Bars: BIT-EXACT for NNFs, u8 images and — because the fp32 summation order is part of
the specification (oracle/orc_nnf.c header) — also for every fp32 output of these functions."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("dims", [(44, 44, 44, 44), (29, 43, 38, 60), (16, 16, 63, 63), (2, 2, 5, 3), (175, 175, 175, 175)])
def test_nnf_init_bit_exact(ctx, oracle, dims):
    assert np.array_equal(ctx.nnf_init(*dims), oracle.nnf_init(*dims))


@pytest.mark.parametrize("dims", [(44, 44, 88, 88, 88, 88), (88, 88, 175, 175, 175, 175), (29, 43, 57, 85, 75, 120), (57, 85, 113, 170, 150, 240), (16, 16, 32, 32, 32, 32)])
def test_nnf_upsample_bit_exact(ctx, oracle, dims):
    ahh, awh, ah, aw, bh, bw = dims
    half = synth.random_nnf(5, ahh, awh, (bh + 1) // 2, (bw + 1) // 2)
    assert np.array_equal(ctx.nnf_upsample(half, ah, aw, bh, bw), oracle.nnf_upsample(half, ah, aw, bh, bw))


@pytest.mark.parametrize("shape", [(64, 33, 47), (512, 16, 16), (128, 40, 40), (256, 21, 19), (8, 9, 7), (20, 5, 5)])
def test_normalize_bit_exact(ctx, oracle, shape):
    f = synth.features(1, *shape)
    g, gr = ctx.feat_normalize(f, want_resp=True)
    o, orr = oracle.feat_normalize(f, want_resp=True)
    assert np.array_equal(bits(g), bits(o))
    assert np.array_equal(bits(gr), bits(orr))


def test_normalize_zero_pixel_is_nan_like_reference(ctx):
    f = synth.features(2, 64, 6, 6)
    f[:, 2, 3] = 0
    g = ctx.feat_normalize(f)
    assert np.isnan(g[:, 2, 3]).all() and np.isfinite(g[:, 0, 0]).all()


PM_CASES = [
    # C, ah, aw, bh, bw, iters, rs_max
    (64, 40, 44, 36, 48, 10, 32),     # ragged A/B sizes, conv1_1-like
    (512, 16, 16, 16, 16, 10, 16),    # config 1, level 5 (256x256 image)
    (512, 29, 43, 38, 60, 3, 43),     # demo pair geometry (in0/tar0), level 5
    (128, 33, 35, 31, 37, 4, 32),
    (256, 25, 27, 30, 22, 4, 10),
    (8, 19, 23, 17, 29, 5, 4),        # generic-C path
    (64, 6, 5, 3, 4, 2, 32),          # tiny: rs_max > max(bw,bh) clamp, every query on a border
    (64, 70, 70, 70, 70, 2, 0),       # rs_max = 0 -> no random search
]


@pytest.mark.parametrize("case", PM_CASES)
def test_patchmatch_bit_exact(ctx, oracle, case):
    C, ah, aw, bh, bw, iters, rs = case
    a = oracle.feat_normalize(synth.features(10 + C, C, ah, aw))
    b = oracle.feat_normalize(synth.features(20 + C, C, bh, bw))
    nnf0 = oracle.nnf_init(ah, aw, bh, bw)
    gn, gd = ctx.patchmatch(a, b, nnf0, iters=iters, rs_max=rs, seed=1234)
    on, od = oracle.patchmatch(a, b, nnf0, iters=iters, rs_max=rs, seed=1234)
    assert np.array_equal(gn, on), f"NNF mismatch at {np.argwhere(gn != on)[:5]}"
    assert np.array_equal(bits(gd), bits(od))


def test_patchmatch_from_random_nnf_and_other_seed(ctx, oracle):
    a = oracle.feat_normalize(synth.features(31, 64, 30, 30))
    b = oracle.feat_normalize(synth.features(32, 64, 28, 34))
    nnf0 = synth.random_nnf(9, 30, 30, 28, 34)
    for seed in (0, 0xDEADBEEF):
        gn, gd = ctx.patchmatch(a, b, nnf0, iters=3, rs_max=8, seed=seed)
        on, od = oracle.patchmatch(a, b, nnf0, iters=3, rs_max=8, seed=seed)
        assert np.array_equal(gn, on) and np.array_equal(bits(gd), bits(od))


def test_patchmatch_rejects_bad_arguments(ctx):
    import nct
    a = synth.features(1, 64, 8, 8)
    with pytest.raises(nct.NctError):
        ctx.patchmatch(a, a, np.zeros((8, 8), np.uint32), patch=5)
    with pytest.raises(nct.NctError):
        ctx.patchmatch(a[:6], a[:6], np.zeros((8, 8), np.uint32))      # C=6 not a multiple of 4


def test_patchmatch_full_size_properties(ctx):
    """BASELINE config-2 finest level (700x700x64): size-independent properties instead of an oracle run:
    determinism, valid coordinates, per-pixel energy monotone in iterations, planted-shift recovery."""
    f = synth.features(77, 64, 720, 720, smooth=False)
    a = f[:, 0:700, 0:700]
    b = f[:, 7:707, 11:711]
    an, bn = ctx.feat_normalize(a), ctx.feat_normalize(b)
    nnf0 = ctx.nnf_init(700, 700, 700, 700)
    n1, d1 = ctx.patchmatch(an, bn, nnf0, iters=2, rs_max=32, seed=5)
    n2, d2 = ctx.patchmatch(an, bn, nnf0, iters=2, rs_max=32, seed=5)
    assert np.array_equal(n1, n2) and np.array_equal(bits(d1), bits(d2))
    n0, d0 = ctx.patchmatch(an, bn, nnf0, iters=0, rs_max=32, seed=5)
    assert np.array_equal(n0, nnf0)
    assert (d1 <= d0).all()
    n10, d10 = ctx.patchmatch(an, bn, nnf0, iters=10, rs_max=32, seed=5)
    assert (d10 <= d1).all()
    assert (n10 & 0xFFF).max() < 700 and ((n10 >> 12) & 0xFFF).max() < 700
    yy, xx = np.mgrid[20:680, 20:680]
    ok = ((n10[20:680, 20:680] & 0xFFF) == xx - 11) & (((n10[20:680, 20:680] >> 12) & 0xFFF) == yy - 7)
    assert ok.mean() > 0.9
    assert np.allclose(d10[20:680, 20:680][ok], -1.0, atol=2e-5)


@pytest.mark.parametrize("shape", [(64, 23, 31), (512, 9, 8), (12, 7, 7)])
def test_feature_distance_bit_exact(ctx, oracle, shape):
    a = oracle.feat_normalize(synth.features(40, *shape))
    b = oracle.feat_normalize(synth.features(41, *shape))
    assert np.array_equal(bits(ctx.feature_distance(a, b)), bits(oracle.feature_distance(a, b)))


VOTE_CASES = [(64, 20, 24, 18, 27), (512, 9, 10, 11, 8), (128, 15, 15, 15, 15), (256, 8, 9, 10, 7), (24, 12, 13, 9, 14)]


@pytest.mark.parametrize("case", VOTE_CASES)
@pytest.mark.parametrize("kind", ["random", "collapsed"])
def test_bds_vote_features_bit_exact(ctx, oracle, case, kind):
    C, ah, aw, bh, bw = case
    pin = synth.features(50 + C, C, bh, bw) * np.float32(37.0)       # un-normalised activations
    ann = synth.random_nnf(1, ah, aw, bh, bw)
    bnn = synth.random_nnf(2, bh, bw, ah, aw)
    if kind == "collapsed":        # many R pixels matched to the same S pixel: long source lists
        bnn[:, : bw // 2] = (np.uint32(ah // 2) << 12) | np.uint32(aw // 2)
    for wc in (2.0, 0.0, 8.0):
        g, gw = ctx.bds_vote_features(ann, bnn, pin, 1.0, wc, want_pw=True)
        o, ow = oracle.bds_vote_features(ann, bnn, pin, 1.0, wc, want_pw=True)
        assert np.array_equal(bits(gw), bits(ow))
        assert np.array_equal(bits(g), bits(o))


@pytest.mark.parametrize("dims", [(20, 24, 18, 27), (44, 44, 44, 44), (7, 5, 9, 11)])
def test_bds_vote_image_bit_exact(ctx, oracle, dims):
    ah, aw, bh, bw = dims
    a, b = synth.image(1, ah, aw), synth.image(2, bh, bw)
    ann = synth.random_nnf(3, ah, aw, bh, bw)
    bnn = synth.random_nnf(4, bh, bw, ah, aw)
    for wc in (2.0, 0.0, 1.0, 4.0, 8.0):       # the demo's BDS sweep (demo/example/pairs.txt:5-9)
        assert np.array_equal(ctx.bds_vote_image(a, b, ann, bnn, 1.0, wc), oracle.bds_vote_image(a, b, ann, bnn, 1.0, wc))


# ---- the pipeline's instantiations: both directions fused per launch, grouped candidate evaluation ------------------------------
@pytest.mark.parametrize("C,ah,aw,bh,bw,rs", [(64, 37, 41, 33, 45, 8), (128, 30, 26, 28, 31, 16), (256, 21, 24, 23, 20, 8), (512, 14, 13, 12, 15, 4), (64, 9, 70, 80, 7, 32),
                                              (64, 120, 90, 100, 110, 32), (128, 70, 64, 66, 72, 32), (24, 20, 22, 21, 19, 8)])
def test_patchmatch_bidir_bit_exact(ctx, oracle, C, ah, aw, bh, bw, rs):
    """k_pm_step as the pipeline launches it (S->R and R->S fields in the same launches), plain and with the exact row-wise rejection,
    against the oracle's two single-direction runs with the same seeds: identical NNF and bit-identical distances in BOTH directions."""
    fa, fb = synth.features(21, C, ah, aw), synth.features(22, C, bh, bw)
    a, b = oracle.feat_normalize(fa), oracle.feat_normalize(fb)
    seed = 77
    o_ann, o_annd = oracle.patchmatch(a, b, oracle.nnf_init(ah, aw, bh, bw), iters=5, rs_max=rs, seed=seed)
    o_bnn, o_bnnd = oracle.patchmatch(b, a, oracle.nnf_init(bh, bw, ah, aw), iters=5, rs_max=rs, seed=seed ^ 0x5bd1e995)
    ctx.pm_bench_setup(fa, fb)                         # uploads, normalises, builds the fp16 shadow maps
    counts = []
    for mode in (0, 1):
        ms, cnt, ann, annd, bnn, bnnd = ctx.pm_bench_run_bidir(iters=5, rs_max=rs, seed=seed, pm_mode=mode, count=True, fetch=True, both=True)
        assert np.array_equal(ann, o_ann) and np.array_equal(bnn, o_bnn), f"mode {mode}: NNF differs from the oracle"
        assert np.array_equal(annd.view(np.uint32), o_annd.view(np.uint32)) and np.array_equal(bnnd.view(np.uint32), o_bnnd.view(np.uint32)), f"mode {mode}: distances differ"
        counts.append(cnt)
    assert counts[0] == counts[1] and counts[0][0] > 0 and counts[0][1] > 0           # same candidates, same acceptances


@pytest.mark.parametrize("C,ah,aw,bh,bw,rs", [(64, 37, 41, 33, 45, 8), (128, 30, 26, 28, 31, 16), (256, 21, 24, 23, 20, 8), (512, 14, 13, 12, 15, 4), (64, 120, 90, 100, 110, 32)])
def test_patchmatch_persistent_level_kernel_bit_exact(oracle, monkeypatch, C, ah, aw, bh, bw, rs):
    """Round 6: NCT_PM_PERSIST=1 runs a pyramid level as ONE persistent launch — k_pm_level: (step, tile) items from per-(step, XCD) ticket counters,
    steps ordered by per-tile flags instead of kernel boundaries, NNF words exchanged through agent-scope accesses. Opt-in (measured slower than the per-step launches on
    MI355X: DESIGN §9), but it must compute the same field: both directions, both evaluation modes, same evaluation counts as the oracle's schedule."""
    import nct
    monkeypatch.setenv("NCT_PM_PERSIST", "1")
    fa, fb = synth.features(21, C, ah, aw), synth.features(22, C, bh, bw)
    a, b = oracle.feat_normalize(fa), oracle.feat_normalize(fb)
    seed = 77
    o_ann, o_annd = oracle.patchmatch(a, b, oracle.nnf_init(ah, aw, bh, bw), iters=5, rs_max=rs, seed=seed)
    o_bnn, o_bnnd = oracle.patchmatch(b, a, oracle.nnf_init(bh, bw, ah, aw), iters=5, rs_max=rs, seed=seed ^ 0x5bd1e995)
    with nct.Context(0) as c:                              # the flag is read when a context is created
        c.pm_bench_setup(fa, fb)
        counts = []
        for mode in (0, 1):
            for rep in range(2):                           # twice: the control block (tickets, flags) is rebuilt per level, nothing may leak from the first run
                ms, cnt, ann, annd, bnn, bnnd = c.pm_bench_run_bidir(iters=5, rs_max=rs, seed=seed, pm_mode=mode, count=True, fetch=True, both=True)
                assert np.array_equal(ann, o_ann) and np.array_equal(bnn, o_bnn), f"mode {mode}: NNF differs from the oracle"
                assert np.array_equal(annd.view(np.uint32), o_annd.view(np.uint32)) and np.array_equal(bnnd.view(np.uint32), o_bnnd.view(np.uint32)), f"mode {mode}: distances differ"
            counts.append(cnt)
        assert counts[0] == counts[1] and counts[0][0] > 0


def _same_or_both_nan(a, b):
    """bit-identical where finite; NaN where the other is NaN (x86 and gfx950 produce different NaN payloads for 0/0: 0xFFC00000 vs 0x7FC00000)"""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


@pytest.mark.parametrize("C,ah,aw,bh,bw,rs", [(64, 37, 41, 33, 45, 8), (128, 30, 26, 28, 31, 16), (256, 21, 24, 23, 20, 8), (512, 14, 13, 12, 15, 4), (64, 120, 90, 100, 110, 32)])
def test_patchmatch_dead_feature_pixels(ctx, oracle, C, ah, aw, bh, bw, rs):
    """VERDICT r4 item 5 / SURVEY 9 quirk 3: `norm` has no epsilon (GeneralizedPatchMatch.cu:276-277), so an all-zero feature pixel (post-ReLU conv1_1 of a saturated flat
    region) becomes a NaN vector. The defined behaviour = IEEE comparisons, as in the reference's `d < dbest`: a distance that involves a NaN tap is NaN and never wins; a
    query whose OWN patch holds a NaN keeps the match it started with. The pipeline's instantiation with the exact row rejection (pm_mode 1: its Cauchy-Schwarz bound
    assumes unit vectors) must give the oracle's field on such maps: identical NNFs, distances bit-identical where finite and NaN where the oracle's are NaN."""
    fa, fb = synth.features(31, C, ah, aw), synth.features(32, C, bh, bw)
    fa[:, 5:9, 7:12] = 0; fa[:, ah - 1, aw - 1] = 0; fb[:, 3:8, 2:6] = 0; fb[:, 0, 0] = 0; fb[:, bh // 2, bw // 2] = 0
    a, b = oracle.feat_normalize(fa), oracle.feat_normalize(fb)
    assert np.isnan(a[:, 6, 8]).all() and np.isnan(b[:, 0, 0]).all()
    seed = 91
    o_ann, o_annd = oracle.patchmatch(a, b, oracle.nnf_init(ah, aw, bh, bw), iters=5, rs_max=rs, seed=seed)
    o_bnn, o_bnnd = oracle.patchmatch(b, a, oracle.nnf_init(bh, bw, ah, aw), iters=5, rs_max=rs, seed=seed ^ 0x5bd1e995)
    assert np.isnan(o_annd).any() and np.isfinite(o_annd).any()
    ctx.pm_bench_setup(fa, fb)
    for mode in (0, 1):
        ms, cnt, ann, annd, bnn, bnnd = ctx.pm_bench_run_bidir(iters=5, rs_max=rs, seed=seed, pm_mode=mode, count=True, fetch=True, both=True)
        assert np.array_equal(ann, o_ann) and np.array_equal(bnn, o_bnn), f"mode {mode}: NNF differs from the oracle on maps with dead pixels"
        assert _same_or_both_nan(annd, o_annd) and _same_or_both_nan(bnnd, o_bnnd), f"mode {mode}: distances differ"


def test_patchmatch_fp16_mode_close(ctx):
    """Opt-in reduced-precision mode (NCT_FLAG_FEAT16): fp16 candidate tiles, fp32 accumulate. Not bit-identical by definition; the
    match energies stay within the fp16 rounding bound of the fp32 field's."""
    for C, ah, aw, bh, bw in ((64, 48, 52, 50, 46), (128, 40, 36, 38, 44), (256, 24, 28, 26, 22), (512, 16, 18, 17, 15)):
        ctx.pm_bench_setup(synth.features(31, C, ah, aw), synth.features(32, C, bh, bw))
        _, _, ann, annd = ctx.pm_bench_run_bidir(iters=6, rs_max=16, seed=5, pm_mode=1, fetch=True)
        _, _, ann16, annd16 = ctx.pm_bench_run_bidir(iters=6, rs_max=16, seed=5, pm_mode=2, fetch=True)
        assert np.abs(annd16.astype(np.float64).mean() - annd.astype(np.float64).mean()) < 2e-3, C
        assert (ann16 == ann).mean() > 0.8, C



@pytest.mark.parametrize("name", ["44x44x512", "175x175x256", "256x256x64", "350x350x128"])
def test_patchmatch_energy_within_reference_schedule_band(ctx, oracle, name):
    """SURVEY §8c G4 / DESIGN §4 divergence 1+2, quantified: the product runs PatchMatch as double-buffered Jacobi steps with a counter RNG and a 16-lane fp32
    tree; the reference runs ONE racy in-place launch with sequential channel sums and column-shared cuRAND streams (GeneralizedPatchMatch.cu:677-831). The
    committed fixture holds the energy statistics of the reference's own schedule under two legal interleavings (oracle/orc_nnf_inplace.c, generated by
    tests/golden/gen_pm_inplace_band.py). Band asserted here, on the GPU result of the same inputs:
      * mean annd not worse than the better of the two interleavings (+1e-4) and within 1 % of both;
      * every stored percentile (5/25/50/75/95) within 1 % of both interleavings;
      * the GPU field equals the fixture's product_jacobi statistics (the bit-exact oracle mirror) to 1e-6."""
    import json, os
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pm_inplace_band.json")))["cases"][name]
    C, ah, aw, bh, bw = fx["C"], fx["ah"], fx["aw"], fx["bh"], fx["bw"]
    a = oracle.feat_normalize(synth.features(fx["feature_seeds"][0], C, ah, aw)); b = oracle.feat_normalize(synth.features(fx["feature_seeds"][1], C, bh, bw))
    nnf, d = ctx.patchmatch(a, b, ctx.nnf_init(ah, aw, bh, bw), iters=fx["iters"], rs_max=fx["rs_max"], seed=fx["pm_seed"])
    st = oracle.field_stats(d)
    assert np.allclose(st, fx["product_jacobi"]["stats"], rtol=0, atol=1e-6)
    seq, lock = np.array(fx["reference_sequential"]["stats"]), np.array(fx["reference_lockstep"]["stats"])
    assert st[0] <= min(seq[0], lock[0]) + 1e-4, (st[0], seq[0], lock[0])
    for ref in (seq, lock):
        assert (np.abs(st - ref) <= 0.01 * np.abs(ref)).all(), (st, ref)
    assert float(np.mean(d < oracle.patchmatch(a, b, ctx.nnf_init(ah, aw, bh, bw), iters=0, rs_max=fx["rs_max"], seed=fx["pm_seed"])[1])) > 0.95
