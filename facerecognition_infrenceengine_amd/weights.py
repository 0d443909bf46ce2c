"""Model state dicts: seeded synthetic generators and checkpoint loading.

The reference obtains its networks by name from the network
(``FaceAnalysis(name="buffalo_l")``, /root/reference/infrenceServer.py:412-416),
which is impossible offline.  This module therefore provides
(a) loaders for real checkpoints in the public PyTorch naming (IResNet:
``conv1.weight, bn1.*, prelu.weight, layerN.i.{bn1,conv1,bn2,prelu,conv2,bn3,
downsample.0,downsample.1}, bn2.*, fc.*, features.*``; MTCNN P/R/O:
``conv*.{weight,bias}, prelu*.weight, conv4_1/conv4_2, dense*``) and (b) seeded
synthetic weights of the same shapes for parity tests and benchmarks.
Synthetic BN/PReLU parameters are deliberately non-trivial so folding bugs show.
"""
import math

import torch

IRESNET_LAYERS = {"r18": [2, 2, 2, 2], "r34": [3, 4, 6, 3], "r50": [3, 4, 14, 3],
                  "r100": [3, 13, 30, 3]}
IRESNET_WIDTHS = [64, 128, 256, 512]


def _bn(g, state, prefix, c, gamma=(0.8, 1.2)):
    state[prefix + ".weight"] = torch.empty(c).uniform_(gamma[0], gamma[1], generator=g)
    state[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
    state[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
    state[prefix + ".running_var"] = torch.empty(c).uniform_(0.8, 1.2, generator=g)


def _conv(g, cout, cin, kh, kw):
    std = math.sqrt(2.0 / (cin * kh * kw))
    return torch.randn(cout, cin, kh, kw, generator=g) * std


def synth_iresnet_state(arch="r100", seed=1234):
    """Seeded IResNet state dict (float32 CPU tensors)."""
    layers = IRESNET_LAYERS[arch]
    g = torch.Generator().manual_seed(seed)
    st = {}
    st["conv1.weight"] = _conv(g, 64, 3, 3, 3)
    _bn(g, st, "bn1", 64)
    st["prelu.weight"] = torch.empty(64).uniform_(0.15, 0.35, generator=g)
    cin = 64
    for li, (n, cout) in enumerate(zip(layers, IRESNET_WIDTHS), start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            _bn(g, st, p + ".bn1", cin)
            st[p + ".conv1.weight"] = _conv(g, cout, cin, 3, 3)
            _bn(g, st, p + ".bn2", cout)
            st[p + ".prelu.weight"] = torch.empty(cout).uniform_(0.15, 0.35, generator=g)
            st[p + ".conv2.weight"] = _conv(g, cout, cout, 3, 3)
            _bn(g, st, p + ".bn3", cout, gamma=(0.1, 0.2))
            if bi == 0:
                st[p + ".downsample.0.weight"] = _conv(g, cout, cin, 1, 1)
                _bn(g, st, p + ".downsample.1", cout)
            cin = cout
    _bn(g, st, "bn2", 512)
    st["fc.weight"] = torch.randn(512, 512 * 7 * 7, generator=g) * math.sqrt(1.0 / (512 * 49))
    st["fc.bias"] = torch.randn(512, generator=g) * 0.05
    _bn(g, st, "features", 512)
    return st


def _mt_conv(g, st, name, cout, cin, k):
    st[name + ".weight"] = _conv(g, cout, cin, k, k)
    st[name + ".bias"] = torch.randn(cout, generator=g) * 0.05


def _mt_fc(g, st, name, cout, cin, gain=1.0):
    st[name + ".weight"] = torch.randn(cout, cin, generator=g) * (gain * math.sqrt(1.0 / cin))
    st[name + ".bias"] = torch.randn(cout, generator=g) * 0.05


def _mt_prelu(g, st, name, c):
    st[name + ".weight"] = torch.empty(c).uniform_(0.1, 0.4, generator=g)


# Synthetic face-logit offsets: with untrained weights the heads would fire on
# ~half of all cells; these shift the 'face' logit so the cascade keeps a
# realistic fraction at the standard 0.6/0.7/0.7 thresholds (calibrated on the
# bench's synthetic frames; see DESIGN.md "Synthetic weights").
SYNTH_FACE_LOGIT_BIAS = {"pnet": -0.3, "rnet": -0.6, "onet": 1.75}


def synth_mtcnn_states(seed=4321, face_logit_bias=None):
    """(pnet, rnet, onet) seeded state dicts."""
    fb = dict(SYNTH_FACE_LOGIT_BIAS)
    if face_logit_bias:
        fb.update(face_logit_bias)
    g = torch.Generator().manual_seed(seed)
    p = {}
    _mt_conv(g, p, "conv1", 10, 3, 3); _mt_prelu(g, p, "prelu1", 10)
    _mt_conv(g, p, "conv2", 16, 10, 3); _mt_prelu(g, p, "prelu2", 16)
    _mt_conv(g, p, "conv3", 32, 16, 3); _mt_prelu(g, p, "prelu3", 32)
    _mt_conv(g, p, "conv4_1", 2, 32, 1); _mt_conv(g, p, "conv4_2", 4, 32, 1)
    p["conv4_2.weight"] *= 0.2
    p["conv4_1.bias"] = torch.tensor([0.0, fb["pnet"]])
    r = {}
    _mt_conv(g, r, "conv1", 28, 3, 3); _mt_prelu(g, r, "prelu1", 28)
    _mt_conv(g, r, "conv2", 48, 28, 3); _mt_prelu(g, r, "prelu2", 48)
    _mt_conv(g, r, "conv3", 64, 48, 2); _mt_prelu(g, r, "prelu3", 64)
    _mt_fc(g, r, "dense4", 128, 576); _mt_prelu(g, r, "prelu4", 128)
    _mt_fc(g, r, "dense5_1", 2, 128, 2.0); _mt_fc(g, r, "dense5_2", 4, 128, 0.2)
    r["dense5_1.bias"] = torch.tensor([0.0, fb["rnet"]])
    o = {}
    _mt_conv(g, o, "conv1", 32, 3, 3); _mt_prelu(g, o, "prelu1", 32)
    _mt_conv(g, o, "conv2", 64, 32, 3); _mt_prelu(g, o, "prelu2", 64)
    _mt_conv(g, o, "conv3", 64, 64, 3); _mt_prelu(g, o, "prelu3", 64)
    _mt_conv(g, o, "conv4", 128, 64, 2); _mt_prelu(g, o, "prelu4", 128)
    _mt_fc(g, o, "dense5", 256, 1152); _mt_prelu(g, o, "prelu5", 256)
    _mt_fc(g, o, "dense6_1", 2, 256, 2.0); _mt_fc(g, o, "dense6_2", 4, 256, 0.2)
    _mt_fc(g, o, "dense6_3", 10, 256, 0.3)
    o["dense6_1.bias"] = torch.tensor([0.0, fb["onet"]])
    # landmarks centred in the box so the 5-point similarity is well conditioned
    o["dense6_3.bias"] = torch.tensor([0.3, 0.7, 0.5, 0.35, 0.65, 0.35, 0.35, 0.55, 0.75, 0.75])
    return p, r, o


def load_state(path):
    """Load a real checkpoint (torch ``.pt``/``.pth`` state dict, ``.safetensors``, or an ArcFace IResNet ``.onnx``
    such as the ``w600k_r50.onnx`` of the reference's buffalo_l pack: onnx_import.py)."""
    if str(path).endswith(".onnx"):
        from .onnx_import import iresnet_state_from_onnx
        st, _ = iresnet_state_from_onnx(str(path))
        return {k: torch.from_numpy(v) for k, v in st.items()}
    if str(path).endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(str(path))
    st = torch.load(str(path), map_location="cpu", weights_only=True)
    return st.get("state_dict", st) if isinstance(st, dict) else st
