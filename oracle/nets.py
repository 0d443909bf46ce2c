"""Oracle: network forwards in plain torch-CPU fp32 (explicit BN, NCHW).

PARITY UNPINNED: these restate the published architectures
because the reference delegates them to third-party insightface
(/root/reference/infrenceServer.py:412-416,528).  State dicts use the public
naming of the respective PyTorch definitions (IResNet: conv1/bn1/prelu/layerN.i.{bn1,
conv1,bn2,prelu,conv2,bn3,downsample.{0,1}}/bn2/fc/features; MTCNN: conv1..4,
prelu1..5, conv4_1/conv4_2, dense4/5/6_*), so real checkpoints drop in.

Test infrastructure only (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _t(state, key):
    v = state[key]
    return v if isinstance(v, torch.Tensor) else torch.as_tensor(v)


def _b(state, key):
    """optional conv bias (present when an exporter folded the following BatchNorm into the conv)"""
    return _t(state, key) if key in state else None


def _bn(x, state, prefix):
    if prefix + ".weight" not in state:        # folded into the conv before it by an exporter (onnx_import.py)
        return x
    return F.batch_norm(x, _t(state, prefix + ".running_mean"), _t(state, prefix + ".running_var"),
                        _t(state, prefix + ".weight"), _t(state, prefix + ".bias"),
                        training=False, eps=BN_EPS)


def _prelu(x, state, key):
    return F.prelu(x, _t(state, key))


IRESNET_LAYERS = {"r18": [2, 2, 2, 2], "r34": [3, 4, 6, 3], "r50": [3, 4, 14, 3],
                  "r100": [3, 13, 30, 3]}


@torch.no_grad()
def iresnet_forward(state, x, layers, taps=None):
    """ArcFace IResNet.  x: float32 [B,3,112,112] RGB in (x-127.5)/127.5.

    Returns the un-normalised 512-d embedding.  ``taps`` (optional dict) receives
    named intermediate activations (NCHW) for layer-by-layer kernel checks.
    """
    x = F.conv2d(x, _t(state, "conv1.weight"), _b(state, "conv1.bias"), stride=1, padding=1)
    x = _prelu(_bn(x, state, "bn1"), state, "prelu.weight")
    if taps is not None:
        taps["stem"] = x
    for li, nblocks in enumerate(layers, start=1):
        for bi in range(nblocks):
            p = f"layer{li}.{bi}"
            stride = 2 if bi == 0 else 1
            out = _bn(x, state, p + ".bn1")
            out = F.conv2d(out, _t(state, p + ".conv1.weight"), _b(state, p + ".conv1.bias"), stride=1, padding=1)
            out = _prelu(_bn(out, state, p + ".bn2"), state, p + ".prelu.weight")
            if taps is not None and bi == 0:
                taps[p + ".mid"] = out
            out = F.conv2d(out, _t(state, p + ".conv2.weight"), _b(state, p + ".conv2.bias"), stride=stride, padding=1)
            out = _bn(out, state, p + ".bn3")
            if bi == 0:
                sc = F.conv2d(x, _t(state, p + ".downsample.0.weight"), _b(state, p + ".downsample.0.bias"), stride=stride)
                sc = _bn(sc, state, p + ".downsample.1")
            else:
                sc = x
            x = out + sc
        if taps is not None:
            taps[f"layer{li}"] = x
    x = _bn(x, state, "bn2")
    x = torch.flatten(x, 1)            # NCHW order: c*49 + h*7 + w (dropout = identity)
    x = F.linear(x, _t(state, "fc.weight"), _t(state, "fc.bias"))
    if "features.weight" in state:
        x = F.batch_norm(x, _t(state, "features.running_mean"), _t(state, "features.running_var"),
                         _t(state, "features.weight"), _t(state, "features.bias"),
                         training=False, eps=BN_EPS)
    return x


@torch.no_grad()
def pnet_forward(state, x):
    """x: [B,3,H,W] (x-127.5)/128.  Returns (prob_face [B,Hc,Wc], reg [B,4,Hc,Wc])."""
    x = _prelu(F.conv2d(x, _t(state, "conv1.weight"), _t(state, "conv1.bias")), state, "prelu1.weight")
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv2.weight"), _t(state, "conv2.bias")), state, "prelu2.weight")
    x = _prelu(F.conv2d(x, _t(state, "conv3.weight"), _t(state, "conv3.bias")), state, "prelu3.weight")
    a = F.conv2d(x, _t(state, "conv4_1.weight"), _t(state, "conv4_1.bias"))
    b = F.conv2d(x, _t(state, "conv4_2.weight"), _t(state, "conv4_2.bias"))
    return torch.softmax(a, dim=1)[:, 1], b


def _flatten_whc(x):
    """MTCNN dense layers take the feature map flattened W-major: (w, h, c)."""
    return x.permute(0, 3, 2, 1).contiguous().view(x.shape[0], -1)


@torch.no_grad()
def rnet_forward(state, x):
    """x: [B,3,24,24].  Returns (prob_face [B], reg [B,4])."""
    x = _prelu(F.conv2d(x, _t(state, "conv1.weight"), _t(state, "conv1.bias")), state, "prelu1.weight")
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv2.weight"), _t(state, "conv2.bias")), state, "prelu2.weight")
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv3.weight"), _t(state, "conv3.bias")), state, "prelu3.weight")
    x = _flatten_whc(x)
    x = _prelu(F.linear(x, _t(state, "dense4.weight"), _t(state, "dense4.bias")), state, "prelu4.weight")
    a = F.linear(x, _t(state, "dense5_1.weight"), _t(state, "dense5_1.bias"))
    b = F.linear(x, _t(state, "dense5_2.weight"), _t(state, "dense5_2.bias"))
    return torch.softmax(a, dim=1)[:, 1], b


@torch.no_grad()
def onet_forward(state, x):
    """x: [B,3,48,48].  Returns (prob_face [B], reg [B,4], landmarks [B,10] = x1..x5,y1..y5)."""
    x = _prelu(F.conv2d(x, _t(state, "conv1.weight"), _t(state, "conv1.bias")), state, "prelu1.weight")
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv2.weight"), _t(state, "conv2.bias")), state, "prelu2.weight")
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv3.weight"), _t(state, "conv3.bias")), state, "prelu3.weight")
    x = F.max_pool2d(x, 2, 2, ceil_mode=True)
    x = _prelu(F.conv2d(x, _t(state, "conv4.weight"), _t(state, "conv4.bias")), state, "prelu4.weight")
    x = _flatten_whc(x)
    x = _prelu(F.linear(x, _t(state, "dense5.weight"), _t(state, "dense5.bias")), state, "prelu5.weight")
    a = F.linear(x, _t(state, "dense6_1.weight"), _t(state, "dense6_1.bias"))
    b = F.linear(x, _t(state, "dense6_2.weight"), _t(state, "dense6_2.bias"))
    c = F.linear(x, _t(state, "dense6_3.weight"), _t(state, "dense6_3.bias"))
    return torch.softmax(a, dim=1)[:, 1], b, c
