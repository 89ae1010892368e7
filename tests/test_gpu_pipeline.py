"""End-to-end parity of the per-pair hot loop (D3): nct_process_pair on the GPU vs orc_process_pair on the CPU, same
synthetic VGG19 weights, same images. Every stage is specified bit-for-bit (canonical summation orders, counter RNG,
deterministic elementary functions), so the 8-bit result is expected to be IDENTICAL; the stated tolerance is the
north-star bar: PSNR >= 50 dB per channel (min over channels) and L-inf reported."""
import numpy as np
import pytest
import synth
from caffemodel_io import synthetic_vgg19

pytestmark = pytest.mark.gpu


def psnr_min_channel(a, b):
    d = (a.astype(np.float64) - b.astype(np.float64)) ** 2
    mse = d.reshape(-1, 3).mean(0)
    return float(min(99.0 if m == 0 else 10 * np.log10(255.0 ** 2 / m) for m in mse))


@pytest.fixture(scope="module")
def weights():
    return synthetic_vgg19(19)


@pytest.mark.parametrize("case", [((64, 64), (64, 64), 2.0), ((80, 56), (48, 72), 2.0), ((64, 64), (64, 64), 0.0), ((72, 72), (60, 90), 8.0)])
def test_pair_end_to_end_matches_oracle(ctx, oracle, weights, case):
    import nct
    (sh, sw), (rh, rw), bds = case
    ws, bs = weights
    ctx.vgg19_load_raw(ws, bs)
    src, ref = synth.image(1000, sh, sw), synth.image(1001, rh, rw)
    prm = nct.Params.default()
    prm.bds_weight = bds
    got = ctx.process_pair(src, ref, prm)
    exp, levels = oracle.process_pair(src, ref, ws, bs, dict(bds_weight=bds), want_levels=True)
    linf = int(np.abs(got.astype(int) - exp.astype(int)).max())
    p = psnr_min_channel(got, exp)
    print(f"case {case}: PSNR(min channel) = {p:.2f} dB, L-inf = {linf}")
    assert p >= 50.0, f"PSNR {p:.2f} dB < 50 dB, L-inf {linf}"
    # the transfer must actually do something (not the identity) and stay a valid image
    assert np.abs(exp.astype(int) - src.astype(int)).mean() > 1.0


def test_pair_is_deterministic_and_timing_sane(ctx, weights):
    ws, bs = weights
    ctx.vgg19_load_raw(ws, bs)
    src, ref = synth.image(7, 96, 96), synth.image(8, 96, 96)
    a = ctx.process_pair(src, ref)
    b, tm = ctx.process_pair(src, ref, want_timing=True)
    assert np.array_equal(a, b)
    assert tm["total_ms"] > 0 and abs(sum(tm[k] for k in ("vgg_ms", "cluster_ms", "patchmatch_ms", "vote_ms", "knn_ms", "color_ms", "other_ms")) - tm["total_ms"]) < 0.2 * tm["total_ms"] + 5
    assert all(1 <= it < 100000 for it in tm["wls_iters"])
    # split API == fused API
    ctx.pair_upload(src, ref); ctx.pair_run(); c = ctx.pair_download()
    assert np.array_equal(a, c)


def test_pair_argument_errors(ctx, weights):
    import nct
    ws, bs = weights
    ctx.vgg19_load_raw(ws, bs)
    small = synth.image(1, 8, 8)
    with pytest.raises(nct.NctError):
        ctx.process_pair(small, small)                      # below the minimum side
    with nct.Context(0) as c2:
        with pytest.raises(nct.NctError) as e:
            c2.process_pair(synth.image(1, 32, 32), synth.image(2, 32, 32))   # no weights loaded
        assert e.value.code == -5
