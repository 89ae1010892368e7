"""Test helper: write/read a minimal .caffemodel (protobuf wire format) without libprotobuf.
Schema restated from code/src/caffe/proto/caffe.proto: NetParameter{name=1, layers=2 (V1), layer=100 (V2)};
V1LayerParameter{bottom=2, top=3, name=4, type=5 (enum, CONVOLUTION=4), blobs=6}; LayerParameter{name=1, type=2, blobs=7};
BlobProto{num=1, channels=2, height=3, width=4, data=5 (packed float), shape=7{dim=1 packed int64}}."""
import struct
import numpy as np

VGG_NAMES = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4",
             "conv4_1", "conv4_2", "conv4_3", "conv4_4", "conv5_1", "conv5_2", "conv5_3", "conv5_4"]
VGG_CIN = [3, 64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512]
VGG_COUT = [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512]


def varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def tag(field, wt):
    return varint((field << 3) | wt)


def ld(field, payload):
    return tag(field, 2) + varint(len(payload)) + payload


def blob(arr, legacy_dims=None, shape_dims=None, unpacked=False):
    a = np.ascontiguousarray(arr, np.float32)
    out = b""
    if legacy_dims is not None:
        for f, d in zip((1, 2, 3, 4), legacy_dims):
            out += tag(f, 0) + varint(d)
    if unpacked:
        out += b"".join(tag(5, 5) + struct.pack("<f", float(x)) for x in a.reshape(-1))
    else:
        out += ld(5, a.tobytes())
    if shape_dims is not None:
        out += ld(7, ld(1, b"".join(varint(d) for d in shape_dims)))
    return out


def write_caffemodel(path, weights, biases, names=VGG_NAMES, fmt="v1", extra_layers=True, unpacked=False):
    """weights[i]: (Cout,Cin,3,3); biases[i]: (Cout,). fmt 'v1' = NetParameter.layers (like the Oxford VGG file), 'v2' = .layer."""
    body = ld(1, b"VGG_ILSVRC_19_layers")
    for i, (w, b) in enumerate(zip(weights, biases)):
        co, ci = w.shape[:2]
        if fmt == "v1":
            wb = blob(w, legacy_dims=(co, ci, 3, 3), unpacked=unpacked)
            bb = blob(b, legacy_dims=(1, 1, 1, co), unpacked=unpacked)
            layer = ld(2, b"x") + ld(3, names[i].encode()) + ld(4, names[i].encode()) + tag(5, 0) + varint(4) + ld(6, wb) + ld(6, bb)
            body += ld(2, layer)
            if extra_layers:   # in-place ReLU (type 18) without blobs: must be ignored
                body += ld(2, ld(4, ("relu" + names[i][4:]).encode()) + tag(5, 0) + varint(18))
        else:
            wb = blob(w, shape_dims=(co, ci, 3, 3))
            bb = blob(b, shape_dims=(co,))
            layer = ld(1, names[i].encode()) + ld(2, b"Convolution") + ld(7, wb) + ld(7, bb)
            body += ld(100, layer)
    if extra_layers:           # an fc layer with blobs under an unknown name: ignored by name matching (net.cpp:770-773)
        fcw = blob(np.ones((4, 8), np.float32), legacy_dims=(1, 1, 4, 8))
        if fmt == "v1":
            body += ld(2, ld(4, b"fc6") + tag(5, 0) + varint(14) + ld(6, fcw))
        else:
            body += ld(100, ld(1, b"fc6") + ld(2, b"InnerProduct") + ld(7, fcw))
    with open(path, "wb") as f:
        f.write(body)


def synthetic_vgg19(seed=19, nlayers=13, bias_scale=0.0):
    """He-normal N(0, 2/(9*Cin)) weights; conv1_1 additionally scaled for 0-255 inputs so activations stay O(1-10)."""
    rng = np.random.default_rng(seed)
    ws, bs = [], []
    for i in range(nlayers):
        std = np.sqrt(2.0 / (9 * VGG_CIN[i]))
        w = rng.standard_normal((VGG_COUT[i], VGG_CIN[i], 3, 3)).astype(np.float32) * np.float32(std)
        if i == 0:
            w *= np.float32(1.0 / 64.0)
        ws.append(np.ascontiguousarray(w))
        bs.append((rng.standard_normal(VGG_COUT[i]).astype(np.float32) * np.float32(bias_scale)) if bias_scale else np.zeros(VGG_COUT[i], np.float32))
    return ws, bs


def write_deploy_prototxt(path, v1=False, drop=None, num_output=None, extra_tail=True, input_layer=False, per_axis=False, relu_in_place=True, kernel_hw=None):
    """A deploy prototxt of the VGG19 topology in protobuf text format (own writer; the reference ships one under demo/model/vgg19/): `layer` messages with string
    types, or V1 `layers` with enum types. drop: name of a layer to leave out; num_output: {conv name: value} overrides — for the negative tests.
    input_layer: declare the input as `layer { type: "Input" }` (input_layer.cpp) instead of the legacy `input:` fields; per_axis: spell kernel / pad as kernel_h, kernel_w,
    pad_h, pad_w; relu_in_place=False: every ReLU writes a blob of its own; kernel_hw: {conv name: (h, w)} (non-square kernels, negative test)."""
    names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "conv4_1", "conv4_2", "conv4_3", "conv4_4",
             "conv5_1", "conv5_2", "conv5_3", "conv5_4"]
    cout = [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512]
    pool_after = {"conv1_2": "pool1", "conv2_2": "pool2", "conv3_4": "pool3", "conv4_4": "pool4", "conv5_4": "pool5"}
    kw = "layers" if v1 else "layer"
    ty = (lambda s_: {"Convolution": "CONVOLUTION", "ReLU": "RELU", "Pooling": "POOLING", "InnerProduct": "INNER_PRODUCT"}[s_]) if v1 else (lambda s_: '"%s"' % s_)
    out = ['name: "VGG_ILSVRC_19_layer"', 'input: "data"', "# comment line", "input_shape {", "  dim: 1", "  dim: 3", "  dim: 224", "  dim: 224", "}"]
    if input_layer and not v1:
        out = ['name: "VGG_ILSVRC_19_layer"', "layer {", '  name: "data"', '  type: "Input"', '  top: "data"', "  input_param { shape: { dim: 1 dim: 3 dim: 224 dim: 224 } }", "}"]
    cur = "data"
    for i, n in enumerate(names):
        if not extra_tail and i > 12:
            break
        no = (num_output or {}).get(n, cout[i])
        if n != drop:
            kh, kw_ = (kernel_hw or {}).get(n, (3, 3))
            geom = ["    pad_h: 1", "    pad_w: 1", f"    kernel_h: {kh}", f"    kernel_w: {kw_}"] if (per_axis or (kh, kw_) != (3, 3)) else ["    pad: 1", "    kernel_size: 3"]
            out += [f"{kw} {{", f'  bottom: "{cur}"', f'  top: "{n}"', f'  name: "{n}"', f"  type: {ty('Convolution')}", "  convolution_param {", f"    num_output: {no}"] + geom + ["  }", "}"]
            cur = n
        r = "relu" + n[4:]
        if r != drop:
            rt = cur if relu_in_place else r
            out += [f"{kw} {{", f'  bottom: "{cur}"', f'  top: "{rt}"', f'  name: "{r}"', f"  type: {ty('ReLU')}", "}"]
            cur = rt
        if n in pool_after and pool_after[n] != drop and (extra_tail or i < 12):
            pn = pool_after[n]
            out += [f"{kw} {{", f'  bottom: "{cur}"', f'  top: "{pn}"', f'  name: "{pn}"', f"  type: {ty('Pooling')}", "  pooling_param {", "    pool: MAX", "    kernel_size: 2", "    stride: 2", "  }", "}"]
            cur = pn
    if extra_tail:
        out += [f"{kw} {{", f'  bottom: "{cur}"', '  top: "fc6"', '  name: "fc6"', f"  type: {ty('InnerProduct')}", "  inner_product_param {", "    num_output: 4096", "  }", "}"]
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
