import sys, math
sys.path.insert(0, "/root/repo")
import torch
from facerecognition_infrenceengine_amd import _lib, weights
from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
st = weights.synth_mtcnn_states(seed=99)
d = MTCNNHIP(*st, device="cuda:0")
p1, lib = d.p1, d.lib
print("slopes", st[0]["prelu1.weight"])
g = torch.Generator(device="cuda").manual_seed(5)
for (N, H, W) in [(2, 120, 160), (1, 250, 333), (3, 37, 53), (1, 480, 640)]:
    frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device="cuda", dtype=torch.uint8)
    for sc in pyramid_scales(H, W):
        hs, ws = int(math.ceil(H * sc)), int(math.ceil(W * sc))
        if hs < 3 or ws < 3: continue
        h, w = p1.out_hw(hs, ws)
        y = torch.full((N, h, w, 12), float("nan"), device="cuda")
        xs16 = torch.full((N, h, w, 64), 0x7f, dtype=torch.uint8, device="cuda")
        lib.fr_dconv_mfma_f32(0, None, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), _lib.ptr(y), N, hs, ws, None, None, _lib.ptr(frames), H, W, None, 0, None, _lib.stream_ptr())
        lib.fr_pnet_conv1_band(0, _lib.ptr(frames), N, H, W, hs, ws, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), None, _lib.ptr(xs16), None, None, 0, _lib.stream_ptr())
        torch.cuda.synchronize()
        hl = xs16.view(torch.float16).reshape(N, h, w, 2, 16).float()
        bad = torch.isnan(hl).any(-1).any(-1)
        dec = hl[..., 0, :12] + hl[..., 1, :12]
        err = (dec - y).abs().amax(-1)
        err[bad] = 0
        print((N, H, W), (hs, ws), (h, w), "unwritten", bad.nonzero().tolist()[:6], int(bad.sum()), "maxerr %.2e" % float(err.max()), "scale %.2f" % float(y.abs().max()))
