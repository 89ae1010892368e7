"""GPU parity for VGG19 (V1 caffemodel ingest, V2 features): f32-MFMA conv is a k-ordered fmaf chain, so the bar is
BIT-EXACT vs oracle/orc_vgg.c (which itself is pinned to Caffe's known answers and torch in tests/test_oracle_vgg.py)."""
import os
import numpy as np
import pytest
import synth
from caffemodel_io import synthetic_vgg19, write_caffemodel

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


CONV_CASES = [(3, 64, 17, 23), (64, 64, 40, 70), (64, 128, 33, 35), (128, 256, 20, 45), (256, 512, 11, 13), (512, 512, 9, 16),
              (4, 64, 2, 2), (64, 64, 180, 200), (128, 128, 5, 177), (64, 128, 2, 67), (6, 64, 130, 2), (32, 128, 131, 129)]       # the last one: the 128-cout x 128-px workgroup form (>= 128 workgroups)


@pytest.mark.parametrize("shape", CONV_CASES)
def test_conv3x3_bit_exact(ctx, oracle, shape):
    cin, cout, H, W = shape
    rng = np.random.default_rng(cin * 1000 + W)
    x = rng.standard_normal((cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    for relu in (True, False):
        g = ctx.conv3x3_relu(x, w, b, relu)
        o = oracle.conv3x3(x, w, b, relu)
        assert np.array_equal(bits(g), bits(o)), f"max abs diff {np.abs(g - o).max()}"


def test_conv_asymmetric_identity_check(ctx):
    """Transpose-detecting check (guide rule 16): delta kernels at asymmetric taps move the image by known offsets."""
    x = np.random.default_rng(0).standard_normal((2, 9, 11)).astype(np.float32)
    w = np.zeros((64, 2, 3, 3), np.float32)
    w[0, 0, 0, 2] = 1        # out0(y,x) = in0(y-1, x+1)
    w[5, 1, 2, 1] = 1        # out5(y,x) = in1(y+1, x)
    y = ctx.conv3x3_relu(x, w, np.zeros(64, np.float32), relu=False)
    exp0 = np.zeros((9, 11), np.float32); exp0[1:, :-1] = x[0, :-1, 1:]
    exp5 = np.zeros((9, 11), np.float32); exp5[:-1, :] = x[1, 1:, :]
    assert np.array_equal(y[0], exp0) and np.array_equal(y[5], exp5) and not y[1:5].any()


@pytest.mark.parametrize("shape", [(64, 700, 700), (3, 5, 7), (512, 88, 88), (128, 175, 175), (7, 2, 2), (16, 113, 170)])
def test_maxpool_exact(ctx, oracle, shape):
    x = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
    assert np.array_equal(bits(ctx.maxpool2x2(x)), bits(oracle.maxpool2x2(x)))


def test_caffe_pool_known_answer_on_gpu(ctx):
    """test_pooling_layer.cpp:56-99 uses kernel 2 / stride 1; the product only ships the VGG geometry (2x2/2, ceil), so
    the same input is checked against the hand-computed 2x2/2 ceil answer: [[9,5,8],[2,5,3]]."""
    x = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)[None]
    assert np.array_equal(ctx.maxpool2x2(x)[0], np.array([[9, 5, 8], [2, 5, 3]], np.float32))


@pytest.fixture(scope="module")
def weights():
    return synthetic_vgg19(19, bias_scale=0.05)


@pytest.mark.parametrize("hw", [(40, 52), (64, 64), (70, 45)])
def test_vgg19_features_bit_exact(ctx, oracle, weights, hw):
    ws, bs = weights
    ctx.vgg19_load_raw(ws, bs)
    img = synth.image(7, *hw)
    g = ctx.vgg19_features(img, 5)
    o = oracle.vgg19_features(img, ws, bs, 5)
    for t in range(5):
        assert g[t].shape == o[t].shape
        assert np.array_equal(bits(g[t]), bits(o[t])), f"tap {t + 1}: max abs diff {np.abs(g[t] - o[t]).max()}"
    # stopping at a shallower tap gives the same shallow taps
    g2 = ctx.vgg19_features(img, 2)
    assert len(g2) == 2 and np.array_equal(bits(g2[1]), bits(o[1]))


@pytest.mark.parametrize("fuse", ["1", "0"])
@pytest.mark.parametrize("hw", [(40, 52), (71, 45), (66, 96), (2, 2)])
def test_vgg19_pool_in_conv_epilogue_bit_exact(oracle, weights, monkeypatch, hw, fuse):
    """conv1_2 / 2_2 / 3_4 / 4_4 with the following 2x2 max-pool inside the conv epilogue (32 x 2 pixel tiles; forced on — the default only fuses where that tile
    shape fits the map, i.e. at 700^2 and 350^2 of the bench pair — and forced off): same bits as Caffe's conv + ceil-mode clipped pooling, odd sizes included."""
    import nct
    monkeypatch.setenv("NCT_CONV_POOL_FUSE", fuse)
    ws, bs = weights
    img = synth.image(11, *hw)
    with nct.Context(0) as c:
        c.vgg19_load_raw(ws, bs)
        g = c.vgg19_features(img, 5)
    o = oracle.vgg19_features(img, ws, bs, 5)
    for t in range(5):
        assert g[t].shape == o[t].shape and np.array_equal(bits(g[t]), bits(o[t])), f"tap {t + 1}"


@pytest.mark.parametrize("fmt,unpacked", [("v1", False), ("v2", False), ("v1", True)])
def test_caffemodel_ingest_matches_raw(ctx, oracle, weights, tmp_path, fmt, unpacked):
    ws, bs = weights
    if unpacked:      # unpacked repeated-float encoding is 5 bytes/value: keep the file small, swap in tiny conv stacks? no — same net, fewer layers not allowed
        pytest.skip("covered by the CPU-side reader test on a reduced file") if False else None
    path = os.path.join(tmp_path, f"vgg_{fmt}.caffemodel")
    write_caffemodel(path, ws, bs, fmt=fmt, unpacked=unpacked)
    ctx.vgg19_load_caffemodel(path)
    img = synth.image(8, 24, 30)
    g = ctx.vgg19_features(img, 5)
    o = oracle.vgg19_features(img, ws, bs, 5)
    for t in range(5):
        assert np.array_equal(bits(g[t]), bits(o[t]))


@pytest.mark.parametrize("fmt,unpacked,as_double", [("v1", False, False), ("v2", True, False), ("v1", False, True)])
def test_caffemodel_written_by_googles_encoder(ctx, oracle, weights, tmp_path, fmt, unpacked, as_double):
    """V1 against an independent writer: the file comes from google.protobuf over descriptors checked against the reference's caffe.proto
    (tests/caffe_pb.py) — V1 layers with legacy dims, V2 layers with unpacked `data`, V1 with `double_data` — with all the clutter of a real file; the features the
    library computes from it equal the oracle's on the arrays that went in."""
    pytest.importorskip("google.protobuf")
    import caffe_pb
    from caffemodel_io import VGG_NAMES
    ws, bs = weights
    path = os.path.join(tmp_path, "google_%s.caffemodel" % fmt)
    caffe_pb.write_vgg19(path, ws, bs, VGG_NAMES, fmt, unpacked_floats=unpacked, as_double=as_double)
    ctx.vgg19_load_caffemodel(path)
    img = synth.image(9, 26, 31)
    g = ctx.vgg19_features(img, 5)
    o = oracle.vgg19_features(img, ws, bs, 5)
    for t in range(5):
        assert np.array_equal(bits(g[t]), bits(o[t]))


def test_caffemodel_errors(ctx, weights, tmp_path):
    import nct
    ws, bs = weights
    with pytest.raises(nct.NctError) as e:
        ctx.vgg19_load_caffemodel(os.path.join(tmp_path, "missing.caffemodel"))
    assert e.value.code == -4
    # shape mismatch is fatal (net.cpp:780-791)
    bad = [w.copy() for w in ws]
    bad[2] = np.zeros((128, 32, 3, 3), np.float32)
    p = os.path.join(tmp_path, "bad.caffemodel")
    write_caffemodel(p, bad, bs)
    with pytest.raises(nct.NctError) as e:
        ctx.vgg19_load_caffemodel(p)
    assert "conv2_1" in str(e.value) and "mismatch" in str(e.value)
    # a missing conv layer is reported by name
    p2 = os.path.join(tmp_path, "short.caffemodel")
    write_caffemodel(p2, ws[:5], bs[:5])
    with pytest.raises(nct.NctError) as e:
        ctx.vgg19_load_caffemodel(p2)
    assert "conv3_2" in str(e.value)
    # truncated file
    data = open(p, "rb").read()
    p3 = os.path.join(tmp_path, "trunc.caffemodel")
    open(p3, "wb").write(data[: len(data) // 3])
    with pytest.raises(nct.NctError):
        ctx.vgg19_load_caffemodel(p3)


def test_features_before_weights_is_a_state_error():
    import nct
    with nct.Context(0) as c:
        with pytest.raises(nct.NctError) as e:
            c.vgg19_features(synth.image(1, 8, 8), 1)
        assert e.value.code == -5
