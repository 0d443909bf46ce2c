"""Read the recognition network of an insightface model pack straight from its ONNX file.

The reference gets its networks as ONNX files resolved by name (``FaceAnalysis(name="buffalo_l")``,
/root/reference/infrenceServer.py:412-416: ``~/.insightface/models/buffalo_l/w600k_r50.onnx`` is the ArcFace
IResNet-50).  Neither ``onnx`` nor ``onnxruntime`` exists in this image, and only two things are needed
from such a file - the graph's node list and its initialisers - so this module decodes the protobuf wire format
itself (ONNX ``ModelProto`` / ``GraphProto`` / ``NodeProto`` / ``TensorProto`` field numbers from the public
onnx.proto) and maps an IResNet graph onto the state-dict naming of weights.py by FOLLOWING THE DATA FLOW (tensor
names in ONNX exports are arbitrary numbers), not by name.

Both export styles are understood: BatchNormalization kept as nodes, or folded into the preceding Conv / Gemm by
the exporter (then the conv carries a bias and the state dict gets ``<conv>.bias`` and no BN entry; iresnet.py and
the oracle treat a missing BN as the identity).  The pre-activation ``bn1`` of a block can never be folded by an
exporter (zero padding sits between it and the conv) and must be present.

No file of this kind exists in the image: the mapping is tested on graphs written by tests/helpers/onnx_write.py
from seeded state dicts (both styles), not on a real pack - DESIGN.md says so.
"""
import struct

import numpy as np

# --------------------------------------------------------------------------- protobuf wire format (reader)


def _varint(buf, i):
    r = s = 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, i
        s += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value = int (varint / fixed) or memoryview (bytes)."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v = bytes(buf[i:i + 8]); i += 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]; i += ln
        elif wt == 5:
            v = bytes(buf[i:i + 4]); i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield f, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [v]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(x)
    return out


def _signed(x):
    return x - (1 << 64) if x >= 1 << 63 else x


_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 6: np.int32, 7: np.int64}


def _tensor(buf):
    dims, dtype, name, raw, floats, int64s, int32s, external = [], 1, "", None, [], [], [], False
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += [_signed(x) for x in _packed_varints(v, wt)]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:                                    # float_data, packed or one fixed32 per element
            floats.append(bytes(v))
        elif f == 5:                                    # int32_data: int32 payloads, and float16 as uint16 bit patterns
            int32s += [_signed(x) for x in _packed_varints(v, wt)]
        elif f == 7:
            int64s += [_signed(x) for x in _packed_varints(v, wt)]
        elif f == 14:                                   # data_location: 0 = DEFAULT (legal when written explicitly), 1 = EXTERNAL
            external = v == 1
    if external:
        raise ValueError(f"initializer {name!r} stores its data in an external file (not supported)")
    if dtype not in _DTYPES:
        raise ValueError(f"initializer {name!r}: unsupported ONNX data type {dtype}")
    if raw is not None:
        a = np.frombuffer(raw, dtype=np.dtype(_DTYPES[dtype]).newbyteorder("<"))
    elif floats:
        a = np.frombuffer(b"".join(floats), dtype="<f4")
    elif int32s:
        a = np.asarray(int32s, dtype=np.int64)
        a = a.astype(np.uint16).view(np.float16) if dtype == 10 else a.astype(np.int32)
    else:
        a = np.asarray(int64s, dtype=np.int64)
    return name, a.astype(_DTYPES[dtype]).reshape(dims)


def _attribute(buf):
    name, val = "", None
    ints, floats = [], []
    for f, wt, v in _fields(buf):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 8:
            ints += [_signed(x) for x in _packed_varints(v, wt)]
        elif f == 7:
            floats += [struct.unpack("<f", v)[0]] if wt == 5 else list(np.frombuffer(bytes(v), "<f4"))
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


class Node:
    __slots__ = ("op", "name", "inputs", "outputs", "attrs")

    def __init__(self):
        self.op, self.name, self.inputs, self.outputs, self.attrs = "", "", [], [], {}

    def __repr__(self):
        return f"{self.op}({', '.join(self.inputs)}) -> {', '.join(self.outputs)}"


def _node(buf):
    n = Node()
    for f, wt, v in _fields(buf):
        if f == 1:
            n.inputs.append(bytes(v).decode())
        elif f == 2:
            n.outputs.append(bytes(v).decode())
        elif f == 3:
            n.name = bytes(v).decode()
        elif f == 4:
            n.op = bytes(v).decode()
        elif f == 5:
            k, a = _attribute(v)
            n.attrs[k] = a
    return n


class OnnxGraph:
    """nodes (file order = topological), initialisers by name, graph input names that are not initialisers."""

    def __init__(self, nodes, initializers, inputs):
        self.nodes, self.initializers, self.inputs = nodes, initializers, inputs


def read_onnx(path):
    with open(path, "rb") as fh:
        buf = memoryview(fh.read())
    graph = None
    for f, wt, v in _fields(buf):
        if f == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f"{path}: no GraphProto (field 7) in the ModelProto")
    nodes, inits, inputs = [], {}, []
    for f, wt, v in _fields(graph):
        if f == 1:
            nodes.append(_node(v))
        elif f == 5:
            name, a = _tensor(v)
            inits[name] = a
        elif f == 11:                                   # ValueInfoProto: field 1 = name
            for g, _, w in _fields(v):
                if g == 1:
                    inputs.append(bytes(w).decode())
    # Constant nodes are initialisers in all but name
    for n in nodes:
        if n.op == "Constant" and "value" in n.attrs and n.outputs:
            inits[n.outputs[0]] = n.attrs["value"]
    return OnnxGraph([n for n in nodes if n.op != "Constant"], inits, [i for i in inputs if i not in inits])


# --------------------------------------------------------------------------- IResNet graph -> state dict

_ARCH_BY_BLOCKS = {(2, 2, 2, 2): "r18", (3, 4, 6, 3): "r34", (3, 4, 14, 3): "r50", (3, 13, 30, 3): "r100"}


def iresnet_state_from_onnx(path_or_graph):
    """-> (state dict of numpy float32 arrays in the naming of weights.py, arch).  Raises ValueError with the node
    it stopped at when the graph is not an ArcFace IResNet (conv-bn-prelu stem, pre-activation residual blocks,
    bn-flatten-fc-bn head)."""
    g = path_or_graph if isinstance(path_or_graph, OnnxGraph) else read_onnx(path_or_graph)
    init = g.initializers
    consumers = {}
    producer = {}
    for n in g.nodes:
        for t in n.inputs:
            consumers.setdefault(t, []).append(n)
        for t in n.outputs:
            producer[t] = n
    st = {}

    def f32(name):
        if name not in init:
            raise ValueError(f"tensor {name!r} is not an initialiser")
        return np.ascontiguousarray(init[name], dtype=np.float32)

    def only(t, op, what):
        c = [n for n in consumers.get(t, []) if n.op == op]
        if len(c) != 1:
            raise ValueError(f"expected one {op} reading {t!r} ({what}), found {[n.op for n in consumers.get(t, [])]}")
        return c[0]

    def maybe(t, op):
        c = [n for n in consumers.get(t, []) if n.op == op]
        return c[0] if len(c) == 1 else None

    def take_bn(n, prefix):
        eps = n.attrs.get("epsilon", 1e-5)
        if abs(eps - 1e-5) > 1e-9:
            raise ValueError(f"{prefix}: BatchNormalization epsilon {eps} (this engine folds with 1e-5)")
        for key, t in zip(("weight", "bias", "running_mean", "running_var"), n.inputs[1:5]):
            st[f"{prefix}.{key}"] = f32(t).reshape(-1)
        return n.outputs[0]

    def take_conv(n, prefix, k, stride):
        w = f32(n.inputs[1])
        ks = n.attrs.get("kernel_shape", list(w.shape[2:]))
        s = n.attrs.get("strides", [1, 1])
        p = n.attrs.get("pads", [0, 0, 0, 0])
        if list(ks) != [k, k] or list(s) != [stride, stride] or list(p) != [k // 2] * 4 or n.attrs.get("group", 1) != 1:
            raise ValueError(f"{prefix}: conv geometry kernel {ks} strides {s} pads {p} is not {k}x{k}/s{stride}/p{k // 2}")
        st[f"{prefix}.weight"] = w
        if len(n.inputs) > 2 and n.inputs[2]:
            st[f"{prefix}.bias"] = f32(n.inputs[2]).reshape(-1)         # a BatchNormalization folded in by the exporter
        return n.outputs[0]

    def conv_bn(t, conv_prefix, bn_prefix, k, stride, what):
        """Conv [+ BatchNormalization] reading tensor t -> output tensor."""
        c = only(t, "Conv", what)
        t = take_conv(c, conv_prefix, k, stride)
        b = maybe(t, "BatchNormalization")
        if b is not None:
            t = take_bn(b, bn_prefix)
        elif f"{conv_prefix}.bias" not in st:
            raise ValueError(f"{conv_prefix}: neither a BatchNormalization after it nor a folded bias")
        return t

    def take_prelu(t, key, what):
        n = only(t, "PRelu", what)
        st[key] = f32(n.inputs[1]).reshape(-1)
        return n.outputs[0]

    if len(g.inputs) != 1:
        raise ValueError(f"expected one graph input, found {g.inputs}")
    cur = g.inputs[0]
    cur = conv_bn(cur, "conv1", "bn1", 3, 1, "stem conv")
    cur = take_prelu(cur, "prelu.weight", "stem PReLU")
    stages, li, bi = [], 0, 0
    while True:
        bn = only(cur, "BatchNormalization", "block bn1 or the head's bn2")
        after = consumers.get(bn.outputs[0], [])
        if any(n.op in ("Flatten", "Reshape", "Gemm", "MatMul") for n in after):
            break                                                        # the head
        # a block whose shortcut is a conv opens a new stage
        has_sc = any(n.op == "Conv" and n is not None and n.attrs.get("kernel_shape", [0])[0] == 1
                     for n in consumers.get(cur, []))
        if has_sc or li == 0:
            if li:
                stages.append(bi)
            li, bi = li + 1, 0
        p = f"layer{li}.{bi}"
        stride = 2 if bi == 0 else 1
        t = take_bn(bn, p + ".bn1")
        t = conv_bn(t, p + ".conv1", p + ".bn2", 3, 1, p + ".conv1")
        t = take_prelu(t, p + ".prelu.weight", p + ".prelu")
        t = conv_bn(t, p + ".conv2", p + ".bn3", 3, stride, p + ".conv2")
        add = only(t, "Add", p + " residual add")
        other = [x for x in add.inputs if x != t]
        if len(other) != 1:
            raise ValueError(f"{p}: residual add {add!r} does not join two tensors")
        if other[0] != cur:
            sc = [n for n in consumers.get(cur, []) if n.op == "Conv"]
            if len(sc) != 1 or bi != 0:
                raise ValueError(f"{p}: the add's other input {other[0]!r} is neither the block input nor a shortcut conv of it")
            t2 = take_conv(sc[0], p + ".downsample.0", 1, stride)
            b = maybe(t2, "BatchNormalization")
            if b is not None:
                t2 = take_bn(b, p + ".downsample.1")
            if t2 != other[0]:
                raise ValueError(f"{p}: shortcut output {t2!r} is not the add's input {other[0]!r}")
        elif bi == 0:
            raise ValueError(f"{p}: first block of a stage without a shortcut conv")
        cur = add.outputs[0]
        bi += 1
    stages.append(bi)
    arch = _ARCH_BY_BLOCKS.get(tuple(stages))
    if arch is None:
        raise ValueError(f"blocks per stage {stages} match no known IResNet depth {sorted(_ARCH_BY_BLOCKS)}")
    # head: bn2 -> flatten -> fc (Gemm, or MatMul + Add) -> features BN (kept, or folded into the Gemm)
    t = take_bn(bn, "bn2")
    fl = [n for n in consumers.get(t, []) if n.op in ("Flatten", "Reshape")]
    if fl:
        t = fl[0].outputs[0]
    fc = [n for n in consumers.get(t, []) if n.op in ("Gemm", "MatMul")]
    if len(fc) != 1:
        raise ValueError(f"expected the fc Gemm / MatMul after the flatten, found {consumers.get(t, [])}")
    fc = fc[0]
    w = f32(fc.inputs[1])
    if fc.op == "Gemm":
        if fc.attrs.get("alpha", 1.0) != 1.0 or fc.attrs.get("beta", 1.0) != 1.0 or fc.attrs.get("transA", 0):
            raise ValueError("fc Gemm with alpha / beta / transA other than 1 / 1 / 0")
        if not fc.attrs.get("transB", 0):
            w = w.T
        bias = f32(fc.inputs[2]).reshape(-1) if len(fc.inputs) > 2 else np.zeros(w.shape[0], np.float32)
        t = fc.outputs[0]
    else:
        w = w.T
        t = fc.outputs[0]
        a = maybe(t, "Add")
        bias = np.zeros(w.shape[0], np.float32)
        if a is not None:
            bias = f32([x for x in a.inputs if x != t][0]).reshape(-1)
            t = a.outputs[0]
    if w.shape != (512, 512 * 49):
        raise ValueError(f"fc weight {w.shape}, expected (512, 25088)")
    st["fc.weight"], st["fc.bias"] = np.ascontiguousarray(w), bias
    b = maybe(t, "BatchNormalization")
    if b is not None:
        take_bn(b, "features")
    return st, arch
