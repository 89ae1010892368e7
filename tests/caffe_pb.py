"""Test helper: .caffemodel files written by GOOGLE's protobuf encoder (google.protobuf, python-upb) from message descriptors built
programmatically — an independent writer for V1 (Classifier.cpp:16-17 -> Net::CopyTrainedLayersFrom), where tests/caffemodel_io.py is
this project's own reading of the wire format.

The descriptors restate the messages of the reference's code/src/caffe/proto/caffe.proto that a trained-weights file carries:
  BlobShape :5-8, BlobProto :10-23, NetParameter :64-100, LayerParameter :306-345 (the fields a weights file uses), V1LayerParameter :1283-1345,
  ConvolutionParameter (num_output / pad / kernel_size / stride) — enough for a reader to meet every wire type and several fields it must skip.
`check_against_proto_text(path)` re-reads the reference's caffe.proto as text and compares every field restated here (name, number, label, type, packed option)
with it; tests/test_oracle_vgg.py runs that check where /root/reference is mounted, so a mis-read field number cannot pass on both sides."""
import re
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

F = descriptor_pb2.FieldDescriptorProto
OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
_T = {"int32": F.TYPE_INT32, "int64": F.TYPE_INT64, "uint32": F.TYPE_UINT32, "float": F.TYPE_FLOAT, "double": F.TYPE_DOUBLE, "string": F.TYPE_STRING, "bool": F.TYPE_BOOL}

# message -> [(label, type, name, number, packed)]; a type that is not in _T is a message or enum of this file
SCHEMA = {
    "BlobShape": [(REP, "int64", "dim", 1, True)],
    "BlobProto": [(OPT, "BlobShape", "shape", 7, False), (REP, "float", "data", 5, True), (REP, "float", "diff", 6, True),
                  (REP, "double", "double_data", 8, True), (REP, "double", "double_diff", 9, True),
                  (OPT, "int32", "num", 1, False), (OPT, "int32", "channels", 2, False), (OPT, "int32", "height", 3, False), (OPT, "int32", "width", 4, False)],
    "ConvolutionParameter": [(OPT, "uint32", "num_output", 1, False), (OPT, "bool", "bias_term", 2, False), (REP, "uint32", "pad", 3, False),
                             (REP, "uint32", "kernel_size", 4, False), (REP, "uint32", "stride", 6, False)],
    "NetParameter": [(OPT, "string", "name", 1, False), (REP, "string", "input", 3, False), (REP, "BlobShape", "input_shape", 8, False),
                     (REP, "int32", "input_dim", 4, False), (OPT, "bool", "force_backward", 5, False), (OPT, "bool", "debug_info", 7, False),
                     (REP, "LayerParameter", "layer", 100, False), (REP, "V1LayerParameter", "layers", 2, False)],
    "LayerParameter": [(OPT, "string", "name", 1, False), (OPT, "string", "type", 2, False), (REP, "string", "bottom", 3, False), (REP, "string", "top", 4, False),
                       (REP, "float", "loss_weight", 5, False), (REP, "BlobProto", "blobs", 7, False), (REP, "bool", "propagate_down", 11, False),
                       (OPT, "ConvolutionParameter", "convolution_param", 106, False)],
    "V1LayerParameter": [(REP, "string", "bottom", 2, False), (REP, "string", "top", 3, False), (OPT, "string", "name", 4, False),
                         (OPT, "LayerType", "type", 5, False), (REP, "BlobProto", "blobs", 6, False), (REP, "string", "param", 1001, False),
                         (REP, "float", "blobs_lr", 7, False), (REP, "float", "weight_decay", 8, False), (REP, "float", "loss_weight", 35, False),
                         (OPT, "ConvolutionParameter", "convolution_param", 10, False)],
}
V1_LAYER_TYPE = {"NONE": 0, "CONVOLUTION": 4, "DROPOUT": 6, "INNER_PRODUCT": 14, "POOLING": 17, "RELU": 18, "SOFTMAX": 20}     # caffe.proto:1292-1333 (the ones VGG19 uses)


def _build(package, unpacked_floats):
    fd = descriptor_pb2.FileDescriptorProto(name=package + ".proto", package=package, syntax="proto2")
    for mname, fields in SCHEMA.items():
        m = fd.message_type.add(name=mname)
        if mname == "V1LayerParameter":
            e = m.enum_type.add(name="LayerType")
            for k, v in V1_LAYER_TYPE.items():
                e.value.add(name=k, number=v)
        for label, typ, name, number, packed in fields:
            f = m.field.add(name=name, number=number, label=label)
            if typ in _T:
                f.type = _T[typ]
            elif typ == "LayerType":
                f.type = F.TYPE_ENUM; f.type_name = "." + package + ".V1LayerParameter.LayerType"
            else:
                f.type = F.TYPE_MESSAGE; f.type_name = "." + package + "." + typ
            if packed and not (unpacked_floats and typ in ("float", "double")):
                f.options.packed = True
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName(package + "." + n)) for n in SCHEMA}


_CACHE = {}


def messages(unpacked_floats=False):
    """Message classes of the schema above. unpacked_floats: the same messages with `data` / `double_data` declared WITHOUT [packed = true] — Google's encoder then
    writes one tag + fixed32 per value, the encoding a proto2 parser must accept for a packed field as well (and that old Caffe writers produced)."""
    key = bool(unpacked_floats)
    if key not in _CACHE:
        _CACHE[key] = _build("caffe_unpacked" if key else "caffe", key)
    return _CACHE[key]


def check_against_proto_text(path):
    """Every field of SCHEMA against the text of the reference's caffe.proto: returns the number of fields compared; raises AssertionError on any difference."""
    txt = re.sub(r"//[^\n]*", "", open(path).read())
    n = 0
    for mname, fields in SCHEMA.items():
        mm = re.search(r"\bmessage\s+%s\s*\{" % mname, txt)
        assert mm, mname
        depth, i = 1, mm.end()
        while depth:                                     # the message body, nested braces included
            depth += {"{": 1, "}": -1}.get(txt[i], 0); i += 1
        body = txt[mm.end():i - 1]
        for label, typ, name, number, packed in fields:
            fm = re.search(r"\b(optional|repeated|required)\s+([\w.]+)\s+%s\s*=\s*(\d+)\s*(\[[^\]]*\])?\s*;" % re.escape(name), body)
            assert fm, (mname, name)
            assert fm.group(1) == ("repeated" if label == REP else "optional"), (mname, name, fm.group(1))
            assert fm.group(2) == typ, (mname, name, fm.group(2))
            assert int(fm.group(3)) == number, (mname, name, fm.group(3))
            assert bool(fm.group(4) and re.search(r"packed\s*=\s*true", fm.group(4))) == packed, (mname, name, fm.group(4))
            n += 1
    em = re.search(r"enum\s+LayerType\s*\{([^}]*)\}", txt[txt.index("message V1LayerParameter"):])
    vals = dict((k, int(v)) for k, v in re.findall(r"(\w+)\s*=\s*(\d+)\s*;", em.group(1)))
    for k, v in V1_LAYER_TYPE.items():
        assert vals[k] == v, k
    return n


def _fill_blob(bp, arr, dims_style, as_double):
    a = np.ascontiguousarray(arr, np.float32)
    if dims_style == "legacy":                            # num / channels / height / width (how the Oxford VGG file carries them)
        d = list(a.shape) if a.ndim == 4 else [1, 1, 1, a.size]
        bp.num, bp.channels, bp.height, bp.width = d
    else:
        bp.shape.dim.extend(int(x) for x in a.shape)
    if as_double:
        bp.double_data.extend(a.reshape(-1).astype(np.float64).tolist())
    else:
        bp.data.extend(a.reshape(-1).tolist())


def write_vgg19(path, weights, biases, names, fmt="v1", unpacked_floats=False, as_double=False, clutter=True):
    """Serialises a VGG19-shaped NetParameter with Google's encoder. fmt 'v1' = NetParameter.layers (V1LayerParameter, legacy blob dims: the Oxford file's shape),
    'v2' = NetParameter.layer (LayerParameter, BlobShape). clutter: everything a real file carries beside the conv blobs — net name, input declaration, bottoms /
    tops, learning-rate multipliers, convolution_param, ReLU / pooling / dropout / softmax layers without blobs, the three fully connected layers WITH blobs (by-name
    matching must ignore them, net.cpp:770-773), a `param` string in the >15 field-number range (two-byte tag)."""
    M = messages(unpacked_floats)
    net = M["NetParameter"]()
    if clutter:
        net.name = "VGG_ILSVRC_19_layers"
        net.input.append("data")
        net.input_dim.extend([10, 3, 224, 224])
        net.force_backward = False
    prev = "data"
    for i, (w, b) in enumerate(zip(weights, biases)):
        n = names[i]
        if fmt == "v1":
            L = net.layers.add()
            L.type = V1_LAYER_TYPE["CONVOLUTION"]
        else:
            L = net.layer.add()
            L.type = "Convolution"
        L.name = n
        if clutter:
            L.bottom.append(prev); L.top.append(n)
            L.convolution_param.num_output = int(w.shape[0]); L.convolution_param.pad.append(1); L.convolution_param.kernel_size.append(3)
            if fmt == "v1":
                L.blobs_lr.extend([1.0, 2.0]); L.weight_decay.extend([1.0, 0.0]); L.param.append("shared_" + n)
        _fill_blob(L.blobs.add(), w, "legacy" if fmt == "v1" else "shape", as_double)
        _fill_blob(L.blobs.add(), b, "legacy" if fmt == "v1" else "shape", as_double)
        prev = n
        if clutter:                                        # the in-place ReLU behind every conv, a pooling layer behind each block
            R = net.layers.add() if fmt == "v1" else net.layer.add()
            R.name = "relu" + n[4:]; R.bottom.append(n); R.top.append(n)
            if fmt == "v1":
                R.type = V1_LAYER_TYPE["RELU"]
            else:
                R.type = "ReLU"
            if n in ("conv1_2", "conv2_2", "conv3_4", "conv4_4", "conv5_4"):
                P = net.layers.add() if fmt == "v1" else net.layer.add()
                P.name = "pool" + n[4]; P.bottom.append(n); P.top.append(P.name)
                if fmt == "v1":
                    P.type = V1_LAYER_TYPE["POOLING"]
                else:
                    P.type = "Pooling"
                prev = P.name
    if clutter:
        rng = np.random.default_rng(5)
        for n, shape in (("fc6", (1, 1, 16, 24)), ("fc7", (1, 1, 16, 16)), ("fc8", (1, 1, 10, 16))):
            L = net.layers.add() if fmt == "v1" else net.layer.add()
            L.name = n
            if fmt == "v1":
                L.type = V1_LAYER_TYPE["INNER_PRODUCT"]
            else:
                L.type = "InnerProduct"
            _fill_blob(L.blobs.add(), rng.standard_normal(shape).astype(np.float32), "legacy" if fmt == "v1" else "shape", as_double)
            _fill_blob(L.blobs.add(), rng.standard_normal(shape[2]).astype(np.float32), "legacy" if fmt == "v1" else "shape", as_double)
        S = net.layers.add() if fmt == "v1" else net.layer.add()
        S.name = "prob"
        if fmt == "v1":
            S.type = V1_LAYER_TYPE["SOFTMAX"]
        else:
            S.type = "Softmax"
    with open(path, "wb") as f:
        f.write(net.SerializeToString())
    return net
