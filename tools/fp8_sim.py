#!/usr/bin/env python3
"""CPU replay of the FOLDED IResNet forward with fake e4m3 quantisation of chosen body convs (dev tool).

Answers, without a GPU, where the fp8 (BASELINE config C5) embedding error comes from and which mixed settings fit
north_star's 1e-3 cosine bound: the convs selected by a policy see e4m3-rounded inputs (per-tensor static scale, as
fr_conv_nhwc_f8 reads them) and e4m3-rounded folded weights (per-output-channel scale); everything else is fp32 (the
f16 roundings of the product path are 1e-6-level and ignored).  Summation order differs from the MFMA's: irrelevant
at the 1e-3 level.

    python tools/fp8_sim.py [--arch r100] [--faces 8] [--table]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from facerecognition_infrenceengine_amd import weights  # noqa: E402
from facerecognition_infrenceengine_amd.iresnet import fold_iresnet, F8_MAX  # noqa: E402


def q8(t):
    return t.clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn).to(torch.float32)


def quant_w(w):
    """per-output-channel e4m3 of a folded [Cout,Cin,KH,KW] weight (f32 in / out)"""
    sw = (w.abs().amax(dim=(1, 2, 3)) / F8_MAX).clamp_min(1e-30)
    return q8(w / sw[:, None, None, None]) * sw[:, None, None, None]


def border_bias(bias9, cout, H, W):
    """bias9 [3,3,cout] (row class, column class) -> [cout,H,W]"""
    b = bias9.reshape(3, 3, cout)
    rc = torch.ones(H, dtype=torch.long); rc[0] = 0; rc[-1] = 2
    cc = torch.ones(W, dtype=torch.long); cc[0] = 0; cc[-1] = 2
    return b[rc][:, cc].permute(2, 0, 1)


def gptq(w, Hm, damp=0.01):
    """GPTQ rounding of a folded weight [Cout,Cin,3,3] to per-row e4m3 given the input second-moment matrix Hm
    [Cin*9, Cin*9] in (ci, kh, kw) order: column by column, each column's rounding error is pushed onto the
    not-yet-rounded columns along the inverse Hessian."""
    Wm = w.reshape(w.shape[0], -1).double().clone()
    sw = (Wm.abs().amax(dim=1) / F8_MAX).clamp_min(1e-30)
    K = Wm.shape[1]
    Hm = Hm + damp * Hm.diag().mean() * torch.eye(K, dtype=torch.float64)
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(Hm)), upper=True)
    Q = torch.empty_like(Wm)
    for k in range(K):
        q = q8((Wm[:, k] / sw).float()).double() * sw
        Q[:, k] = q
        err = (Wm[:, k] - q) / U[k, k]
        if k + 1 < K:
            Wm[:, k + 1:] -= err[:, None] * U[k, k + 1:][None, :]
    return Q.float().reshape(w.shape)


class Sim:
    gptq = False
    def __init__(self, state, arch):
        f = fold_iresnet(state, arch)
        self.f = f
        self.stem_w = f["stem"]["w4"].float()
        self.stem_b = f["stem"]["bias"].float()
        self.stem_s = f["stem"]["slope"].float()
        self.blocks = []
        for b in f["blocks"]:
            d = {}
            for k in ("c1", "c2", "sc"):
                c = b[k]
                if c is None:
                    d[k] = None
                    continue
                d[k] = {"w": c["w"].float(), "bias": c["bias"].float(), "slope": None if c["slope"] is None else c["slope"].float(),
                        "stride": c["stride"], "cin": c["cin"], "cout": c["cout"], "wq": None}
            self.blocks.append(d)
        self.fc_w = f["fc_w"].float()
        self.fc_b = f["fc_bias"].float()
        self.absmax = {}          # (block, conv) -> calibration absmax of that conv's input

    quant_acts = True
    quant_weights = True

    def conv(self, x, c, key, policy, calibrate, centre):
        """x NCHW f32.  policy(key) -> True: fp8 operands."""
        w = c["w"]
        extra = None
        if calibrate:
            self.absmax[key] = max(self.absmax.get(key, 0.0), float(x.abs().max()))
            if centre:
                self.absmax[(key, "mean")] = x.mean(dim=(0, 2, 3))
            if self.gptq and policy(key):
                xc = x - x.mean(dim=(0, 2, 3))[None, :, None, None] if centre else x
                P = F.unfold(xc, 3, padding=1).permute(1, 0, 2).reshape(w.shape[1] * 9, -1).double()     # [Cin*9, n] (ci, kh, kw)
                Hm = (P.float() @ P.float().t()).double() / P.shape[1]
                c["wq"] = gptq(w, Hm)
        elif policy(key):
            if centre:
                mu = self.absmax[(key, "mean")]
                xc = x - mu[None, :, None, None]
                am = float(xc.abs().max())
                sx = 1.5 * am / F8_MAX
                # conv(x) = conv(xc) + conv(mu inside the image): border dependent, exact, folded into the bias in a product
                extra = F.conv2d(torch.ones_like(x[:1]) * mu[None, :, None, None], w, None, stride=c["stride"], padding=w.shape[-1] // 2)
                x = (q8(xc / sx) * sx) if self.quant_acts else xc
            else:
                sx = 1.5 * self.absmax[key] / F8_MAX
                if self.quant_acts:
                    x = q8(x / sx) * sx
            if c["wq"] is None:
                c["wq"] = quant_w(w)
            if self.quant_weights:
                w = c["wq"]
        y = F.conv2d(x, w, None, stride=c["stride"], padding=w.shape[-1] // 2)
        if extra is not None:
            y = y + extra
        return y

    @torch.no_grad()
    def forward(self, x, policy=lambda key: False, calibrate=False, centre=False):
        h = F.conv2d(x, self.stem_w, self.stem_b, padding=1)
        h = F.prelu(h, self.stem_s)
        for bi, b in enumerate(self.blocks):
            c1, c2, sc = b["c1"], b["c2"], b["sc"]
            H, W = h.shape[2:]
            mid = self.conv(h, c1, (bi, 1), policy, calibrate, centre) + border_bias(c1["bias"], c1["cout"], H, W)[None]
            mid = F.prelu(mid, c1["slope"])
            out = self.conv(mid, c2, (bi, 2), policy, calibrate, centre) + c2["bias"][None, :, None, None]
            short = h if sc is None else F.conv2d(h, sc["w"], sc["bias"], stride=sc["stride"])
            h = out + short
        flat = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
        return flat @ self.fc_w.t() + self.fc_b


def eligible(sim):
    """(block, conv) keys fr_conv_nhwc_f8 takes: 3x3/s1, 128-multiple channels, 28x28 / 14x14 maps"""
    keys, hw = [], 112
    for bi, b in enumerate(sim.blocks):
        for k, c in ((1, b["c1"]), (2, b["c2"])):
            if c["stride"] == 1 and c["cin"] % 128 == 0 and c["cout"] % 128 == 0 and hw in (14, 28):
                keys.append((bi, k))
        hw //= b["c2"]["stride"]
    return keys


def flops(sim, keys=None):
    tot, sel, hw = 0.0, 0.0, 112
    for bi, b in enumerate(sim.blocks):
        for k, c in ((1, b["c1"]), (2, b["c2"])):
            ho = hw // c["stride"]
            f = 2.0 * ho * ho * c["cin"] * c["cout"] * 9
            tot += f
            if keys is not None and (bi, k) in keys:
                sel += f
        if b["sc"] is not None:
            tot += 2.0 * (hw // 2) ** 2 * b["sc"]["cin"] * b["sc"]["cout"]
        hw //= b["c2"]["stride"]
    tot += 2 * 112 * 112 * 27 * 64 + 2 * 25088 * 512
    return sel / tot


def one_minus_cos(a, b):
    return 1.0 - F.cosine_similarity(a, b, dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="r100")
    ap.add_argument("--faces", type=int, default=6)
    ap.add_argument("--operands", action="store_true", help="which operand's rounding matters + centred subsets")
    ap.add_argument("--gptq", type=int, default=0, help="calibration faces for a GPTQ-rounded weight experiment")
    ap.add_argument("--per-conv", action="store_true", help="1-cos with ONE conv in fp8 at a time (noise contribution table)")
    args = ap.parse_args()
    torch.set_num_threads(8)
    st = weights.synth_iresnet_state(args.arch)
    sim = Sim(st, args.arch)
    d = np.load(os.path.join(ROOT, "tests", "golden", "r100_kat.npz"))
    g = torch.Generator().manual_seed(77)
    import torch.nn.functional as F_
    lo = torch.rand((args.faces, 3, 14, 14), generator=g)
    xr = (F_.interpolate(lo, (112, 112), mode="bicubic", align_corners=False) * 0.7 + 0.3 * torch.rand((args.faces, 3, 112, 112), generator=g)).clamp(0, 1)
    xr = ((xr * 255).round() - 127.5) / 127.5
    x = torch.cat([torch.from_numpy(d["x"]), xr]) if args.arch == "r100" else xr
    ref = sim.forward(x)
    if args.arch == "r100":
        print("folded fp32 replay vs golden embedding: 1-cos max %.2e" % float(one_minus_cos(ref[:2], torch.from_numpy(d["embedding"])).max()))
    calib = x[2:] if x.shape[0] > 4 else x
    keys = eligible(sim)
    print(f"{len(keys)} eligible convs = {flops(sim, set(keys)) * 100:.1f} % of the FLOPs")

    def run(name, sel, centre=False):
        sel = set(sel)
        sim.absmax.clear()
        sim.forward(calib, calibrate=True, centre=centre)
        e = sim.forward(x, policy=lambda k: k in sel, centre=centre)
        c = one_minus_cos(e, ref)
        print(f"{name:58s} convs {len(sel):3d}  fp8 FLOP share {flops(sim, sel) * 100:5.1f} %   1-cos max {float(c.max()):.2e} mean {float(c.mean()):.2e}", flush=True)
        return float(c.max())

    if args.gptq:
        g2 = torch.Generator().manual_seed(78)
        lo2 = torch.rand((args.gptq, 3, 14, 14), generator=g2)
        xc_ = (F_.interpolate(lo2, (112, 112), mode="bicubic", align_corners=False) * 0.7 + 0.3 * torch.rand((args.gptq, 3, 112, 112), generator=g2)).clamp(0, 1)
        xc_ = ((xc_ * 255).round() - 127.5) / 127.5
        s3 = [k for k in keys if sim.blocks[k[0]]["c1"]["cout"] == 256]
        s2 = [k for k in keys if sim.blocks[k[0]]["c1"]["cout"] == 128]
        allk = set(keys)
        sim.gptq = True                      # GPTQ weights of every eligible conv, from ONE calibration forward
        sim.absmax.clear()
        sim.forward(xc_, policy=lambda k: k in allk, calibrate=True, centre=True)
        sim.gptq = False
        for name, sel in (("14x14 stage", s3), ("14x14 stage + last 4 of 28x28", s2[-4:] + s3), ("14x14 stage + last 8 of 28x28", s2[-8:] + s3),
                          ("14x14 stage + last 12 of 28x28", s2[-12:] + s3), ("all eligible", keys)):
            sel = set(sel)
            e = sim.forward(x, policy=lambda k: k in sel, centre=True)
            c = one_minus_cos(e, ref)
            print(f"centred + GPTQ: {name:40s} convs {len(sel):3d}  fp8 FLOP share {flops(sim, sel) * 100:5.1f} %  1-cos max {float(c.max()):.2e} mean {float(c.mean()):.2e}", flush=True)
        return
    run("all eligible (round-2 setting)", keys)
    if args.operands:
        sim.quant_acts = False
        run("all eligible, WEIGHTS fp8 only", keys)
        sim.quant_acts, sim.quant_weights = True, False
        run("all eligible, ACTIVATIONS fp8 only", keys)
        run("all eligible, ACTIVATIONS fp8 only, centred", keys, centre=True)
        sim.quant_weights = True
        s3 = [k for k in keys if sim.blocks[k[0]]["c1"]["cout"] == 256]
        run("14x14 stage only, centred", s3, centre=True)
        for n in (50, 44, 40, 36, 30):
            run(f"last {n} convs of the 14x14 stage, centred", s3[-n:], centre=True)
            run(f"first {n} convs of the 14x14 stage, centred", s3[:n], centre=True)
        return
    run("all eligible, mean-centred activations", keys, centre=True)
    run("conv1 of each block only", [k for k in keys if k[1] == 1])
    run("conv2 of each block only", [k for k in keys if k[1] == 2])
    s3 = [k for k in keys if sim.blocks[k[0]]["c1"]["cout"] == 256]
    s2 = [k for k in keys if sim.blocks[k[0]]["c1"]["cout"] == 128]
    run("14x14 stage only", s3)
    run("28x28 stage only", s2)
    for frac in (0.75, 0.5, 0.33, 0.25):
        n = int(len(s3) * frac)
        run(f"first {n} convs of the 14x14 stage", s3[:n])
        run(f"last {n} convs of the 14x14 stage", s3[-n:])
    if args.per_conv:
        for k in keys:
            run(f"only {k}", [k])


if __name__ == "__main__":
    main()
