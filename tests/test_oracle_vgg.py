"""CPU tests pinning the oracle's VGG19 pieces: Caffe's own known-answer tests restated + torch-CPU cross-checks."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F
import synth
from caffemodel_io import synthetic_vgg19


def test_caffe_maxpool_known_answer_2x2_stride1(oracle):
    """code/src/caffe/test/test_pooling_layer.cpp:49-119 (TestForwardSquare): 3x5 input, kernel 2, stride 1 -> 2x4."""
    x = np.array([[1, 2, 5, 2, 3], [9, 4, 1, 4, 8], [1, 2, 5, 2, 3]], np.float32)[None]
    y = oracle.maxpool_generic(x, 2, 1)
    assert y.shape == (1, 2, 4)
    assert np.array_equal(y[0], np.array([[9, 5, 5, 8], [9, 5, 5, 8]], np.float32))


def test_pool_ceil_mode_shapes(oracle):
    """pooling_layer.cpp:90-93: pooled = ceil((n - 2)/2) + 1 (700->350->175->88->44, 1000->…->63, 452->…->29)."""
    for n, exp in [(700, 350), (175, 88), (88, 44), (1000, 500), (125, 63), (452, 226), (113, 57), (57, 29), (5, 3), (2, 1)]:
        assert oracle.l.orc_pool_out_size(n) == exp
    x = np.arange(2 * 5 * 7, dtype=np.float32).reshape(2, 5, 7)
    y = oracle.maxpool2x2(x)
    ref = F.max_pool2d(torch.from_numpy(x)[None], 2, 2, ceil_mode=True)[0].numpy()
    assert y.shape == (2, 3, 4) and np.array_equal(y, ref)


def test_conv_sobel_known_answer(oracle):
    """test_convolution_layer.cpp:498-590 idea: a Sobel x-gradient kernel on a horizontal ramp gives a constant interior
    (8 per unit slope) and zero response on a constant image; pad=1 borders see the zero padding."""
    ramp = np.tile(np.arange(8, dtype=np.float32), (6, 1))[None]
    sob = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    w = np.zeros((64, 1, 3, 3), np.float32); w[0, 0] = sob
    y = oracle.conv3x3(ramp, w, np.zeros(64, np.float32), relu=False)
    assert np.array_equal(y[0, 1:-1, 1:-1], np.full((4, 6), 8, np.float32))
    assert np.array_equal(y[0, 0, 1:-1], np.full(6, 6, np.float32))      # top row: one kernel row falls in the padding
    const = np.full((1, 6, 8), 3, np.float32)
    y2 = oracle.conv3x3(const, w, np.zeros(64, np.float32), relu=False)
    assert np.array_equal(y2[0, 1:-1, 1:-1], np.zeros((4, 6), np.float32))
    assert np.all(y[1:] == 0)


@pytest.mark.parametrize("shape", [(3, 64, 17, 23), (64, 64, 12, 9), (128, 256, 7, 8)])
def test_conv_vs_torch(oracle, shape):
    cin, cout, H, W = shape
    rng = np.random.default_rng(cin)
    x = rng.standard_normal((cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2 / (9 * cin))).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    y = oracle.conv3x3(x, w, b, relu=True)
    ref = F.relu(F.conv2d(torch.from_numpy(x)[None].double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1))[0].numpy()
    assert np.allclose(y, ref, atol=1e-4)


def test_preprocess_mean_subtraction(oracle):
    img = synth.image(3, 5, 6)
    p = oracle.vgg_preprocess(img)
    mean = np.array([103.939, 116.779, 123.68], np.float32)
    for c in range(3):
        assert np.array_equal(p[c], img[..., c].astype(np.float32) - mean[c])


def test_vgg19_forward_vs_torch(oracle):
    """Whole-net cross-check to conv5_1 on a 40x52 image (ceil pooling: 40x52 -> 20x26 -> 10x13 -> 5x7 -> 3x4)."""
    ws, bs = synthetic_vgg19(19, bias_scale=0.05)
    img = synth.image(5, 40, 52)
    taps = oracle.vgg19_features(img, ws, bs, 5)
    assert [t.shape for t in taps] == [(64, 40, 52), (128, 20, 26), (256, 10, 13), (512, 5, 7), (512, 3, 4)]
    x = torch.from_numpy(oracle.vgg_preprocess(img))[None].double()
    pool_after = {1, 3, 7, 11}
    tap_at = {0: 0, 2: 1, 4: 2, 8: 3, 12: 4}
    for i in range(13):
        x = F.relu(F.conv2d(x, torch.from_numpy(ws[i]).double(), torch.from_numpy(bs[i]).double(), padding=1))
        if i in tap_at:
            ref = x[0].numpy()
            got = taps[tap_at[i]]
            assert np.allclose(got, ref, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(ref).max())), f"tap {tap_at[i] + 1}"
            assert got.min() >= 0 and (got > 0).mean() > 0.2      # post-ReLU, informative
        if i in pool_after:
            x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    # stopping early gives identical shallow taps (quirk 9: the reference runs to pool5 regardless)
    t2 = oracle.vgg19_features(img, ws, bs, 2)
    assert np.array_equal(t2[0], taps[0]) and np.array_equal(t2[1], taps[1])


# ---- V1 (caffemodel ingest) against an INDEPENDENT writer: Google's protobuf encoder over descriptors built from caffe.proto's field numbers (tests/caffe_pb.py).
# nct_model_parse_caffemodel / nct_model_layer are host-only entry points of the C ABI: no device is needed to check what the reader leaves in the layers.
def _pb():
    return pytest.importorskip("google.protobuf") and __import__("caffe_pb")


@pytest.mark.skipif(not os.path.exists("/root/reference/code/src/caffe/proto/caffe.proto"), reason="reference not mounted")
def test_caffe_pb_descriptors_match_the_reference_proto_text():
    """every field the test writer uses (name, number, label, type, packed option) compared with the text of the reference's caffe.proto"""
    assert _pb().check_against_proto_text("/root/reference/code/src/caffe/proto/caffe.proto") >= 40


@pytest.mark.parametrize("fmt,unpacked,as_double", [("v1", False, False), ("v2", False, False), ("v1", True, False), ("v2", False, True), ("v1", True, True)])
def test_caffemodel_reader_against_googles_encoder(tmp_path, fmt, unpacked, as_double):
    """A VGG19-shaped net serialised by google.protobuf — V1 `layers` with legacy num/channels/height/width (the Oxford file's form) and V2 `layer` with BlobShape;
    `data` packed, not packed (tag + fixed32 per value), or carried in `double_data`; with everything a real file has beside the conv blobs (net name, input_dim,
    bottoms / tops, blobs_lr, convolution_param, a 1001-numbered string field, ReLU / pooling / softmax layers, fc6-fc8 WITH blobs) — must come out of
    nct_model_parse_caffemodel as exactly the arrays that went in."""
    import nct
    from caffemodel_io import VGG_NAMES
    pb = _pb()
    ws, bs = synthetic_vgg19(23, bias_scale=0.1)
    path = str(tmp_path / "g.caffemodel")
    pb.write_vgg19(path, ws, bs, VGG_NAMES, fmt, unpacked_floats=unpacked, as_double=as_double)
    m = nct.Model(path)
    for i in range(13):
        w, b = m.layer(i)
        assert np.array_equal(w.view(np.uint32), np.ascontiguousarray(ws[i], np.float32).view(np.uint32)), VGG_NAMES[i]
        assert np.array_equal(b.view(np.uint32), np.ascontiguousarray(bs[i], np.float32).view(np.uint32)), VGG_NAMES[i]
    with pytest.raises(nct.NctError):
        m.layer(13)
    m.close()


def test_own_writer_and_googles_encoder_agree(tmp_path):
    """tests/caffemodel_io.py (this project's writer, used by the GPU ingest tests) parsed BY GOOGLE's decoder gives the arrays back: the two readings of the
    wire format are the same one."""
    from caffemodel_io import write_caffemodel, VGG_NAMES
    pb = _pb()
    ws, bs = synthetic_vgg19(29, nlayers=4)
    for fmt in ("v1", "v2"):
        path = str(tmp_path / (fmt + ".caffemodel"))
        write_caffemodel(path, ws, bs, names=VGG_NAMES[:4], fmt=fmt)
        net = pb.messages()["NetParameter"]()
        net.ParseFromString(open(path, "rb").read())
        layers = [l for l in (net.layers if fmt == "v1" else net.layer) if l.name.startswith("conv")]
        assert [l.name for l in layers] == VGG_NAMES[:4]
        for l, w, b in zip(layers, ws, bs):
            assert np.array_equal(np.asarray(l.blobs[0].data, np.float32), np.asarray(w, np.float32).reshape(-1))
            assert np.array_equal(np.asarray(l.blobs[1].data, np.float32), np.asarray(b, np.float32).reshape(-1))
            dims = list(l.blobs[0].shape.dim) if fmt == "v2" else [l.blobs[0].num, l.blobs[0].channels, l.blobs[0].height, l.blobs[0].width]
            assert dims == list(w.shape)
