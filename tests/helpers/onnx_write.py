"""Test helper: write an ArcFace IResNet as an ONNX file without the onnx package (protobuf wire format by hand; field
numbers from the public onnx.proto).  Two export styles: every BatchNormalization kept as a node, or - as the PyTorch
exporter does for eval-mode models - each BN that FOLLOWS a conv / the fc folded into it (the block's pre-activation
bn1 cannot be folded and stays).  Tensor names are running numbers, as in real exports."""
import struct

import numpy as np

IRESNET_LAYERS = {"r18": [2, 2, 2, 2], "r34": [3, 4, 6, 3], "r50": [3, 4, 14, 3], "r100": [3, 13, 30, 3]}
EPS = 1e-5


def _varint(x):
    if x < 0:
        x += 1 << 64
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(field, payload):
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def _vi(field, x):
    return _varint(field << 3 | 0) + _varint(x)


def _tensor(name, a, raw=True):
    a = np.ascontiguousarray(a, dtype=np.float32)
    msg = b"".join(_vi(1, d) for d in a.shape) + _vi(2, 1) + _ld(8, name.encode())
    msg += _ld(9, a.tobytes()) if raw else _ld(4, a.tobytes())            # raw_data, or packed float_data
    return msg


def _attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7)


def _attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2)


def _attr_float(name, v):
    return _ld(1, name.encode()) + _varint(2 << 3 | 5) + struct.pack("<f", v) + _vi(20, 1)


class _Writer:
    def __init__(self):
        self.nodes, self.inits, self.n, self.flip = [], [], 0, False

    def name(self):
        self.n += 1
        return str(self.n)

    def init(self, a):
        nm = self.name()
        self.flip = not self.flip                              # alternate raw_data / float_data encodings
        self.inits.append(_tensor(nm, a, raw=self.flip))
        return nm

    def node(self, op, inputs, attrs=()):
        out = self.name()
        msg = b"".join(_ld(1, i.encode()) for i in inputs) + _ld(2, out.encode()) + _ld(4, op.encode())
        msg += b"".join(_ld(5, a) for a in attrs)
        self.nodes.append(msg)
        return out


def write_iresnet_onnx(path, state, arch, fold_bn):
    st = {k: np.asarray(v, dtype=np.float64) for k, v in state.items()}
    w = _Writer()

    def bn(x, prefix):
        ins = [x] + [w.init(st[f"{prefix}.{k}"]) for k in ("weight", "bias", "running_mean", "running_var")]
        return w.node("BatchNormalization", ins, [_attr_float("epsilon", EPS)])

    def conv_bn(x, cprefix, bprefix, k, stride):
        wt = st[cprefix + ".weight"]
        attrs = [_attr_ints("kernel_shape", [k, k]), _attr_ints("strides", [stride, stride]),
                 _attr_ints("pads", [k // 2] * 4), _attr_ints("dilations", [1, 1]), _attr_int("group", 1)]
        if fold_bn:
            s = st[bprefix + ".weight"] / np.sqrt(st[bprefix + ".running_var"] + EPS)
            t = st[bprefix + ".bias"] - st[bprefix + ".running_mean"] * s
            return w.node("Conv", [x, w.init(wt * s[:, None, None, None]), w.init(t)], attrs)
        return bn(w.node("Conv", [x, w.init(wt)], attrs), bprefix)

    def prelu(x, key):
        return w.node("PRelu", [x, w.init(st[key].reshape(-1, 1, 1))])

    x = "input.1"
    x = prelu(conv_bn(x, "conv1", "bn1", 3, 1), "prelu.weight")
    for li, n in enumerate(IRESNET_LAYERS[arch], start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            stride = 2 if bi == 0 else 1
            t = bn(x, p + ".bn1")
            t = prelu(conv_bn(t, p + ".conv1", p + ".bn2", 3, 1), p + ".prelu.weight")
            t = conv_bn(t, p + ".conv2", p + ".bn3", 3, stride)
            sc = conv_bn(x, p + ".downsample.0", p + ".downsample.1", 1, stride) if bi == 0 else x
            x = w.node("Add", [t, sc])
    x = bn(x, "bn2")
    x = w.node("Flatten", [x], [_attr_int("axis", 1)])
    if fold_bn:
        s = st["features.weight"] / np.sqrt(st["features.running_var"] + EPS)
        t = st["features.bias"] - st["features.running_mean"] * s
        x = w.node("Gemm", [x, w.init(st["fc.weight"] * s[:, None]), w.init(st["fc.bias"] * s + t)],
                   [_attr_float("alpha", 1.0), _attr_float("beta", 1.0), _attr_int("transB", 1)])
    else:
        x = w.node("Gemm", [x, w.init(st["fc.weight"]), w.init(st["fc.bias"])],
                   [_attr_float("alpha", 1.0), _attr_float("beta", 1.0), _attr_int("transB", 1)])
        x = bn(x, "features")
    graph = b"".join(_ld(1, n) for n in w.nodes) + _ld(2, b"iresnet") + b"".join(_ld(5, t) for t in w.inits)
    graph += _ld(11, _ld(1, b"input.1")) + _ld(12, _ld(1, x.encode()))
    model = _vi(1, 7) + _ld(2, b"tests/helpers/onnx_write.py") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 11))
    with open(path, "wb") as fh:
        fh.write(model)
