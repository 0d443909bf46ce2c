"""ArcFace IResNet embed network on the HIP conv kernels.

Stands in for the recognition half of ``FaceAnalysis.get`` / ``face.normed_embedding``
(/root/reference/infrenceServer.py:528,532).  Host side = weight folding + launch plan;
all arithmetic runs in libfrhip.so (``fr_conv_nhwc_f16``, ``fr_fc_reduce_l2norm``).

Folding (inference BN):  s = gamma / sqrt(var + eps),  t = beta - mean * s.
  block:  y = bn3(conv2(prelu(bn2(conv1(bn1(x)))))) + shortcut(x)
  conv1':  w = s2[co] * w1 * s1[ci];  the bn1 shift t1 cannot be a plain bias because the
           zero padding is applied AFTER bn1, so it becomes a border-class bias
           bias9[rc][cc][co] = t2[co] + s2[co] * sum_{valid taps} sum_ci w1 * t1[ci]
           (exact; 9 classes = top/mid/bottom x left/mid/right), epilogue PReLU.
  conv2':  w = s3[co] * w2, bias t3, epilogue += shortcut (f16 residual stream).
  shortcut (first block of a stage): 1x1/s2 conv with sd folded, bias td.
  tail:    bn2 -> flatten(NCHW) -> fc -> features(BN1d) folds into one 25088->512 GEMM
           (columns permuted to NHWC order), run split-K on the same MFMA kernel.
"""
import ctypes
import logging
import threading

import torch

from . import _lib
from .weights import IRESNET_LAYERS, IRESNET_WIDTHS

BN_EPS = 1e-5
FC_SPLITK = 28
# up to this many faces the 3x3 convs run split along K (see IResNetHIP._small_batch_splitk)
SMALL_BATCH = 48      # measured crossover: 32 faces 2.7 vs 3.3 ms, 64 faces 4.6 vs 3.7 ms
# up to this many faces the 3x3 / stride-1 convs with >= 128 input channels run split along K INSIDE a workgroup, one launch
# per conv (fr_conv_inblock_f16) instead of the partials launch + fr_conv_splitk_epilogue (see IResNetHIP._inblock)
INBLOCK_BATCH = 8      # measured (tools/bench_inblock.py, forward ms): 1 face 0.89 vs 1.30, 2: 0.95 / 1.41, 4: 1.23 / 1.65, 5: 1.25 / 1.78,
                       # 6: 1.82 / 1.88, 8: 1.85 / 2.19, 12: 2.94 / 2.11 (one, two or four pixel tiles per workgroup by the workgroup count)
# up to this many faces (single frames) every K slice is at most 3 K steps long: a slice's steps are dependent HBM
# round trips (the weights are cold: 130 MB per forward), so a launch takes ~1.2 us per step + ~3 us
LOW_BATCH = 8
# single-frame plans (4 x 12.8 MB of activations + ~40 MB of split-K scratch per HIP stream) are kept for this many streams
MAX_PLAN_STREAMS = 8
# from this many faces up the stride-1 blocks of the 14x14 stage run as ONE launch with the image resident in LDS
# (fr_conv_stage14_f16: one workgroup per image, one image per CU); below, the per-layer path fills the CUs better
# (measured r100 forward ALONE, stage / layer by layer: 128 faces 4.62 / 4.44 ms, 160: 5.16 / 5.74, 192: 5.49 / 6.17, 256: 6.8 / 7.6).
# Taken from 128 faces all the same: in a pipeline the CUs a 128-workgroup launch leaves free are the detector's - config C3 (8 x 4K
# frames, 128 faces per step; tools/ab_c3_stage_min.py, same box): 19 640 - 19 820 faces/s with the threshold at 144, 20 430 - 20 550 at 128
STAGE14_MIN_BATCH = 128
# fr_conv_walk64_f16: one workgroup per (face, 64-cout group), a face's walk cut into 2 / 4 / 8 pieces while that fills <= 256
# CUs.  Measured (tools/bench_walk64_crossover.py, r100 forward): it beats the per-tile kernel at 64 / 96 / 128 faces (cut walks) and
# from 160 up; between 129 and 159 faces an uncut walk leaves 40 % of the CUs idle and loses by 1 %.  Not in the small-batch
# modes (<= small_batch faces: the prepared single-frame sequence runs fr_conv_nhwc_f16, and a mode's kernels are one set).
WALK64_SKIP = range(129, 160)
STAGE28_MIN_BATCH = 144     # fr_conv_stage28_f16: one workgroup per face, as the 14x14 stage kernel


def _bn_fold(st, prefix, n, conv=None):
    """(scale, shift) of an inference BatchNorm; the identity when the state dict has no such BN (an exporter folded
    it into the conv before it: onnx_import.py).  ``conv``: name of the conv feeding it - its bias, if any, joins
    the shift."""
    if prefix + ".weight" in st:
        s = st[prefix + ".weight"].double() / torch.sqrt(st[prefix + ".running_var"].double() + BN_EPS)
        t = st[prefix + ".bias"].double() - st[prefix + ".running_mean"].double() * s
    else:
        s, t = torch.ones(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    if conv is not None and conv + ".bias" in st:
        t = t + s * st[conv + ".bias"].double()
    return s, t


def _pack_w(w):
    """[Cout,Cin,KH,KW] -> [Cout, KH*KW*Cin] f16 (K contiguous, tap-major)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()


F8_MAX = 448.0                 # largest finite OCP e4m3 value
F8_MARGIN = 1.5                # activation scale head-room over the calibration batch's absmax


def f8_eligible(c, H):
    """Layers fr_conv_nhwc_f8 takes: 3x3 / stride 1 at 28x28 and 14x14 with 128-multiple channels (78 % of r100)."""
    return c.k == 3 and c.stride == 1 and H in (14, 28) and c.cin % 128 == 0 and c.cout % 128 == 0


def round_e4m3(t):
    """Round a float tensor to the nearest OCP e4m3 VALUE (ties to even, saturating at +-448; subnormal step 2^-9) in
    plain float arithmetic - device- and dtype-independent (the GPTQ loop runs in f64 on the GPU)."""
    a = t.abs().clamp(max=F8_MAX)
    _, e = torch.frexp(a.clamp_min(2.0 ** -9))                    # a = m * 2^e, m in [0.5, 1)
    step = torch.ldexp(torch.ones_like(a), (e - 1).clamp_min(-6) - 3)
    return torch.copysign(torch.round(a / step) * step, t)


def gptq_factor(hm, damp=0.01):
    """Upper Cholesky factor U of the inverse of the (damped) second-moment matrix hm [K,K], f64."""
    K = hm.shape[0]
    hm = hm.double() + damp * hm.diagonal().mean().double() * torch.eye(K, dtype=torch.float64, device=hm.device)
    return torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(hm)), upper=True).contiguous()


def gptq_e4m3_torch(w, hm, sw, block=128, damp=0.01):
    """GPTQ rounding (Frantar et al. 2022) of folded weights ``w`` [Cout, K] to the per-row e4m3 grid ``sw[co] * e4m3``
    given the second-moment matrix ``hm`` [K, K] of the conv's (centred) input patches, K order as ``w``'s columns:
    columns are rounded one after the other and each column's rounding error is pushed onto the columns still to
    come along the inverse Hessian, so that the OUTPUT error ||(W - Wq) X|| - not the weight error - is what is
    minimised.  Returns the rounded weights divided by sw (values on the e4m3 grid), f32.  Plain-torch form (blocked):
    the cross-check of fr_gptq_round_e4m3, which the product uses."""
    W = w.double().clone()
    K = W.shape[1]
    U = gptq_factor(hm, damp)
    sw = sw.double()
    Q = torch.empty_like(W)
    for i0 in range(0, K, block):
        i1 = min(i0 + block, K)
        W1, U1 = W[:, i0:i1].clone(), U[i0:i1, i0:i1]
        E1 = torch.empty_like(W1)
        for j in range(i1 - i0):
            q = round_e4m3(W1[:, j] / sw)
            Q[:, i0 + j] = q
            err = (W1[:, j] - q * sw) / U1[j, j]
            W1[:, j:] -= err[:, None] * U1[j, j:][None, :]
            E1[:, j] = err
        if i1 < K:
            W[:, i1:] -= E1 @ U[i0:i1, i1:]
    return Q.to(torch.float32)


def quantise_weights_f8(w):
    """[Cout, K] (f64/f32, folded) -> (uint8 e4m3 bytes [Cout, K], per-output-channel scale sw f32 [Cout]):
    w8 = fp8(w / sw), sw = max|w[co]| / 448."""
    w = w.to(torch.float32)
    sw = (w.abs().amax(dim=1) / F8_MAX).clamp_min(1e-30)
    q = (w / sw[:, None]).clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), sw


def fold_iresnet(state, arch="r100"):
    """Inference-BN folding of an IResNet state dict in f64 on the host (module docstring).  Pure: no device, no
    library - ``IResNetHIP`` packs the result for the kernels, ``tools/fp8_sim.py`` replays it on the CPU.
    Returns {"stem": conv, "blocks": [{"c1", "c2", "sc"}], "fc_w" [512, 49*512] (NHWC K order), "fc_bias"}; a conv is
    {"w" f64 [Cout,Cin,KH,KW] (stem: packed [64,128]), "bias" (c1: bias9 flattened [9*Cout]), "slope", "cin", "cout",
    "stride"}."""
    layers = IRESNET_LAYERS[arch]
    st = {k: torch.as_tensor(v).detach().to("cpu") for k, v in state.items()}
    # stem: conv1 + bn1 + prelu, packed K = 16 taps x 8 channels
    s, t = _bn_fold(st, "bn1", 64, conv="conv1")
    w = st["conv1.weight"].double() * s[:, None, None, None]
    wp = torch.zeros(64, 16, 8, dtype=torch.float64)
    wp[:, :9, :3] = w.permute(0, 2, 3, 1).reshape(64, 9, 3)
    out = {"stem": {"w": wp.reshape(64, 128), "w4": w, "bias": t, "slope": st["prelu.weight"], "cin": 8, "cout": 64,
                    "stride": 1}, "blocks": []}
    cin = 64
    for li, (n, cout) in enumerate(zip(layers, IRESNET_WIDTHS), start=1):
        for bi in range(n):
            p = f"layer{li}.{bi}"
            stride = 2 if bi == 0 else 1
            s1, t1 = _bn_fold(st, p + ".bn1", cin)
            s2, t2 = _bn_fold(st, p + ".bn2", cout, conv=p + ".conv1")
            s3, t3 = _bn_fold(st, p + ".bn3", cout, conv=p + ".conv2")
            w1 = st[p + ".conv1.weight"].double()
            w1f = w1 * s2[:, None, None, None] * s1[None, :, None, None]
            tap = (w1 * t1[None, :, None, None]).sum(1) * s2[:, None, None]      # [co,kh,kw]
            valid = {0: [1, 2], 1: [0, 1, 2], 2: [0, 1]}
            bias9 = torch.empty(3, 3, cout, dtype=torch.float64)
            for rc in range(3):
                for cc in range(3):
                    bias9[rc, cc] = t2 + tap[:, valid[rc]][:, :, valid[cc]].sum((1, 2))
            c1 = {"w": w1f, "bias": bias9.reshape(9 * cout), "slope": st[p + ".prelu.weight"], "cin": cin,
                  "cout": cout, "stride": 1}
            w2f = st[p + ".conv2.weight"].double() * s3[:, None, None, None]
            c2 = {"w": w2f, "bias": t3, "slope": None, "cin": cout, "cout": cout, "stride": stride}
            sc = None
            if bi == 0:
                sd, td = _bn_fold(st, p + ".downsample.1", cout, conv=p + ".downsample.0")
                wd = st[p + ".downsample.0.weight"].double() * sd[:, None, None, None]
                sc = {"w": wd, "bias": td, "slope": None, "cin": cin, "cout": cout, "stride": stride}
            out["blocks"].append({"c1": c1, "c2": c2, "sc": sc})
            cin = cout
    # tail
    sb, tb = _bn_fold(st, "bn2", 512)
    sf, tf = _bn_fold(st, "features", 512)
    W = st["fc.weight"].double().reshape(512, 512, 49)                   # [o, c, hw]
    out["fc_bias"] = sf * (st["fc.bias"].double() + (W * tb[None, :, None]).sum((1, 2))) + tf
    out["fc_w"] = (W * sb[None, :, None] * sf[:, None, None]).permute(0, 2, 1).reshape(512, 49 * 512)   # NHWC K order
    return out


class _Conv:
    __slots__ = ("w", "bias", "slope", "cin", "cout", "k", "stride", "pad", "bias_mode", "w32", "w8", "sw", "sx",
                 "oscale", "mu", "bias8", "c2", "w64")

    def __init__(self, w, bias, slope, cin, cout, k, stride, pad, bias_mode, device):
        self.w = w.to(device)
        self.w32 = self.w8 = self.sw = self.sx = self.oscale = self.mu = self.bias8 = None      # fp8 form, filled by IResNetHIP.enable_fp8
        self.bias = None if bias is None else bias.to(torch.float32).contiguous().to(device)
        self.slope = None if slope is None else slope.to(torch.float32).contiguous().to(device)
        self.cin, self.cout, self.k, self.stride, self.pad, self.bias_mode = cin, cout, k, stride, pad, bias_mode
        self.w64 = None              # fr_conv_walk64_pack'ed weights (3x3 / s1, 64 input channels), IResNetHIP._pack_walk64
        self.c2 = 0                  # channels of a second input that enters through a 1x1 tap (fr_conv_args.x2): fused shortcut


class IResNetHIP:
    """Folded IResNet resident on one GPU.  ``forward`` takes the packed stem input
    f16 [B,112,112,8] (RGB in channels 0..2, (x-127.5)/127.5, rest zero) produced by
    ``fr_warp_affine_5pt`` and returns (embedding, normed_embedding) f32 [B,512] on device."""

    def __init__(self, state, arch="r100", device="cuda:0", max_chunk=256, small_batch=SMALL_BATCH, low_batch=LOW_BATCH,
                 inblock_batch=INBLOCK_BATCH):
        """``small_batch`` / ``low_batch``: batch-size modes of the split-K single-frame path (module constants above;
        arguments, not environment variables: the product reads no environment)."""
        _lib.require_gpu()
        self.small_batch, self.low_batch = int(small_batch), int(low_batch)
        self.inblock_batch = int(inblock_batch)
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.arch = arch
        self.max_chunk = max_chunk
        dev = self.device
        f = fold_iresnet(state, arch)
        self.stem = _Conv(f["stem"]["w"].to(torch.float16).contiguous(), f["stem"]["bias"], f["stem"]["slope"],
                          8, 64, 3, 1, 1, 0, dev)
        self.blocks = []
        for b in f["blocks"]:
            d1, d2, ds = b["c1"], b["c2"], b["sc"]
            c1 = _Conv(_pack_w(d1["w"]), d1["bias"], d1["slope"], d1["cin"], d1["cout"], 3, 1, 1, 1, dev)
            c2 = _Conv(_pack_w(d2["w"]), d2["bias"], None, d2["cin"], d2["cout"], 3, d2["stride"], 1, 0, dev)
            if d1["cin"] % 128 == 0 and d1["cout"] % 128 == 0:        # folded f32 weights kept on the host for enable_fp8()
                c1.w32 = d1["w"].permute(0, 2, 3, 1).reshape(d1["cout"], -1).to(torch.float32)
            if d2["stride"] == 1 and d2["cout"] % 128 == 0:
                c2.w32 = d2["w"].permute(0, 2, 3, 1).reshape(d2["cout"], -1).to(torch.float32)
            sc = None
            if ds is not None:
                sc = _Conv(_pack_w(ds["w"]), ds["bias"], None, ds["cin"], ds["cout"], 1, ds["stride"], 0, 0, dev)
            self.blocks.append((c1, c2, sc))
        # A stage-entry block's 1x1 / stride-2 shortcut conv joins its stride-2 3x3 conv as extra K rows of ONE implicit GEMM
        # (f32 accumulation, biases summed): no shortcut launch, no f16 shortcut map written and read back.
        self.fused_sc = {}
        self.fuse_shortcut = True    # False: shortcut conv as its own launch (A/B, tests)
        for i, (c1, c2, sc) in enumerate(self.blocks):
            if sc is not None and sc.k == 1 and sc.stride == c2.stride and sc.cin % 64 == 0 and c2.cin % 64 == 0:
                fz = _Conv(torch.cat([c2.w.reshape(c2.cout, -1), sc.w.reshape(sc.cout, -1)], 1).contiguous(),
                           c2.bias + sc.bias, None, c2.cin, c2.cout, 3, c2.stride, 1, 0, dev)
                fz.c2 = sc.cin
                self.fused_sc[i] = fz
        self._pack_stage14()
        self._pack_stage28()
        self._pack_walk64()
        self.fc_w = f["fc_w"].to(torch.float16).contiguous().to(dev)
        self.fc_bias = f["fc_bias"].to(torch.float32).contiguous().to(dev)
        self.flops_per_face = self._count_flops()
        self._plans = {}             # (B, stream) -> prepared fr_conv_sequence of the single-frame forward
        self._plan_bufs = {}         # stream -> (4 activation buffers, split-K scratch) shared by that stream's plans
        self._plan_lock = threading.Lock()       # engines cloned with clone_with() share this network across threads
        self._plan_limit_logged = False
        self.profile = None          # bench.py: list collecting (kernel variant, flops, ev0, ev1) per conv launch
        self.fp8 = False             # enable_fp8(): eligible body convs run on the fp8 matrix cores
        self.use_stage28 = True      # False: the 28x28 stage runs layer by layer whatever the batch (A/B, tests)
        self.use_stage14 = True      # False: the 14x14 stage runs layer by layer whatever the batch (A/B, tests)
        self.stage14_f8 = None       # enable_fp8(): the run's fp8 form (fr_conv_stage14_f8)
        self._calib = None

    # ---- the 14x14 stage as one launch (fr_conv_stage14_f16)
    def _pack_stage14(self):
        """The longest run of consecutive stride-1 256 -> 256 blocks (r100: blocks 17..45, the 29 blocks behind stage
        3's entry block): their weights as ONE pre-swizzled stream in kernel order + [10][256] f32 parameters per conv
        (nine border-class biases - a plain bias nine times - and the PReLU slope, 1.0 = none)."""
        self.stage14 = None
        run, best = [], []
        for i, (c1, c2, sc) in enumerate(self.blocks):
            ok = sc is None and c1.cin == 256 and c1.cout == 256 and c2.stride == 1 and c2.cout == 256
            run = run + [i] if ok else []
            if len(run) > len(best):
                best = run
        if len(best) < 2:
            return
        nconv = 2 * len(best)
        per = self.lib.fr_conv_stage14_weight_bytes(1) // 2
        stream = torch.empty(nconv * per, dtype=torch.float16, device=self.device)
        prm = torch.empty((nconv, 10, 256), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for k, i in enumerate(best):
                c1, c2, _ = self.blocks[i]
                for j, c in enumerate((c1, c2)):
                    self.lib.fr_conv_stage14_pack(_lib.ptr(c.w), _lib.ptr(stream[(2 * k + j) * per:]), _lib.stream_ptr())
                    prm[2 * k + j, :9] = c.bias.reshape(9, 256) if c.bias_mode == 1 else c.bias[None, :]
                    prm[2 * k + j, 9] = c.slope if c.slope is not None else 1.0
            torch.cuda.synchronize(self.device)
        self.stage14 = {"first": best[0], "n": len(best), "w": stream, "prm": prm.contiguous()}

    def _pack_walk64(self):
        """Weights of the 3x3 / s1 convs with 64 input channels (112x112, 56x56) in fr_conv_walk64_f16's stream order."""
        self.use_walk64 = True       # False: these convs on the per-tile halo kernel whatever the batch (A/B, tests)
        with torch.cuda.device(self.device):
            for c1, c2, _ in self.blocks:
                for c in (c1, c2):
                    if c.k == 3 and c.stride == 1 and c.cin == 64 and c.cout % 64 == 0:
                        c.w64 = torch.empty(self.lib.fr_conv_walk64_weight_bytes(c.cout) // 2, dtype=torch.float16, device=self.device)
                        self.lib.fr_conv_walk64_pack(_lib.ptr(c.w), _lib.ptr(c.w64), c.cout, _lib.stream_ptr())
            torch.cuda.synchronize(self.device)

    # ---- the 28x28 stage's stride-1 blocks as one launch (fr_conv_stage28_f16)
    def _pack_stage28(self):
        """The longest run of consecutive stride-1 128 -> 128 blocks (r100: blocks 4..15, the 12 blocks behind stage 2's
        entry block): weights as one pre-swizzled stream in kernel order + [10][128] f32 parameters per conv."""
        self.stage28 = None
        run, best = [], []
        for i, (c1, c2, sc) in enumerate(self.blocks):
            ok = sc is None and c1.cin == 128 and c1.cout == 128 and c2.stride == 1 and c2.cout == 128
            run = run + [i] if ok else []
            if len(run) > len(best):
                best = run
        if len(best) < 2:
            return
        nconv = 2 * len(best)
        per = self.lib.fr_conv_stage28_weight_bytes(1) // 2
        stream = torch.empty(nconv * per, dtype=torch.float16, device=self.device)
        prm = torch.empty((nconv, 10, 128), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for k, i in enumerate(best):
                c1, c2, _ = self.blocks[i]
                for j, c in enumerate((c1, c2)):
                    self.lib.fr_conv_stage28_pack(_lib.ptr(c.w), _lib.ptr(stream[(2 * k + j) * per:]), _lib.stream_ptr())
                    prm[2 * k + j, :9] = c.bias.reshape(9, 128) if c.bias_mode == 1 else c.bias[None, :]
                    prm[2 * k + j, 9] = c.slope if c.slope is not None else 1.0
            torch.cuda.synchronize(self.device)
        self.stage28 = {"first": best[0], "n": len(best), "w": stream, "prm": prm.contiguous()}

    def _run_stage28(self, h, B, nblocks):
        """The run's first ``nblocks`` blocks, in place: ``h`` (the run's input, nobody else's) comes back as their output."""
        st = self.stage28
        mid = torch.empty_like(h)
        args = (_lib.ptr(h), _lib.ptr(mid), _lib.ptr(st["w"]), _lib.ptr(st["prm"]), B, nblocks, _lib.stream_ptr())
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.lib.fr_conv_stage28_f16(*args)
            e1.record()
            self.profile.append(("conv_stage28_kernel", 2.0 * B * 784 * 128 * 1152 * 2 * nblocks, e0, e1))
        else:
            self.lib.fr_conv_stage28_f16(*args)
        return h

    def _run_stage14(self, h, B):
        st = self.stage14
        y = torch.empty_like(h)
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.lib.fr_conv_stage14_f16(_lib.ptr(h), _lib.ptr(y), _lib.ptr(st["w"]), _lib.ptr(st["prm"]), B, st["n"], _lib.stream_ptr())
            e1.record()
            self.profile.append(("conv_stage14_kernel<0, 0>", 2.0 * B * 196 * 256 * 2304 * 2 * st["n"], e0, e1))
        else:
            self.lib.fr_conv_stage14_f16(_lib.ptr(h), _lib.ptr(y), _lib.ptr(st["w"]), _lib.ptr(st["prm"]), B, st["n"], _lib.stream_ptr())
        return y

    # ---- fp8 path (BASELINE config C5)
    def fp8_candidates(self):
        """[(conv, H)] of the layers fr_conv_nhwc_f8 takes, in network order."""
        out, hw = [], 112
        for c1, c2, _ in self.blocks:
            for c in (c1, c2):
                if c.w32 is not None and f8_eligible(c, hw):
                    out.append((c, hw))
            hw //= c2.stride
        return out

    def enable_fp8(self, calib_crops, select="accurate", centre=True, gptq=True):
        """Switch body convs to fr_conv_nhwc_f8 (3x3/s1 at 28x28 and 14x14: up to 82 % of the r100 FLOPs).

        e4m3 carries 3 mantissa bits: every fp8 conv adds rounding noise of ~3 % of its output's random part, from the
        weights and from the activations in equal shares, and the noise of the convs adds up in the residual stream
        (measured, tools/fp8_sim.py: 1 - cos of the embedding grows by ~4e-5 per fp8 conv, 3.4e-3 for all 84).  What
        this method does about it, from ONE f16 calibration forward of ``calib_crops`` (f16 [B,112,112,8]):
          * ``centre``: a conv's input is rounded as (x - mu[c]) / sx with mu the per-channel calibration mean; the
            exact term W.mu - which depends on the taps that fall inside the image - joins the 9-class border bias.
            |x - mu| < |x| on average and e4m3's error is relative: halves both noise shares.
          * ``gptq``: the weights are rounded column by column with error feedback along the inverse second-moment
            matrix of the (centred) input patches (``gptq_e4m3``), which removes most of the weight share.
          * ``select``: "accurate" (default) = every eligible 14x14 conv + the last four eligible 28x28 convs (61 % of
            the r100 FLOPs; 1 - cos < 1e-3, north_star's bound); "all" = every eligible conv (82 %; ~1.1e-3).
        Activation scales stay static per tensor: sx = 1.5 * absmax|x - mu| / 448 (e4m3 is a floating format: finer
        scale granularity buys nothing).  The residual stream, stem, stride-2 convs, 1x1 shortcuts, the 7x7 stage and
        the FC stay f16."""
        assert calib_crops.dtype == torch.float16 and calib_crops.shape[1:] == (112, 112, 8)
        cands = self.fp8_candidates()
        if select == "all":
            chosen = [c for c, _ in cands]
        elif select == "accurate":
            chosen = [c for c, hw in cands if hw == 14] + [c for c, hw in cands if hw == 28][-4:]
        else:
            chosen = [c for i, (c, _) in enumerate(cands) if select(i, len(cands))]
        self.fp8 = False
        for c, _ in cands:
            c.oscale = None
        self._calib = {"want": {id(c) for c in chosen}, "centre": centre, "gptq": gptq, "stats": {}}
        self.forward(calib_crops.contiguous())
        stats, self._calib = self._calib["stats"], None
        n = 0
        for c in chosen:
            st = stats.get(id(c))
            if st is None:
                continue
            c.sx = max(st["absmax"] * F8_MARGIN / F8_MAX, 1e-12)
            c.mu = st["mu"]
            w8, sw = quantise_weights_f8(c.w32)
            if st.get("wq") is not None:                  # GPTQ-rounded values on the same per-row grid
                w8 = st["wq"].cpu().to(torch.float8_e4m3fn).view(torch.uint8).contiguous()
            c.w8, c.sw = w8.to(self.device), sw.to(self.device)
            c.oscale = (c.sw * c.sx).contiguous()
            # exact W.mu through the zero padding: 9 border classes (c1's bias is one already; c2's plain bias widens)
            base = c.bias.double().cpu()
            base = base.reshape(3, 3, c.cout) if c.bias_mode == 1 else base[None, None, :].expand(3, 3, c.cout)
            tap = (c.w32.double().reshape(c.cout, 9, c.cin) * (c.mu.double().cpu()[None, None, :] if c.mu is not None else 0.0)).sum(2)
            tap = tap.reshape(c.cout, 3, 3)
            valid = {0: [1, 2], 1: [0, 1, 2], 2: [0, 1]}
            b9 = torch.empty(3, 3, c.cout, dtype=torch.float64)
            for rc in range(3):
                for cc in range(3):
                    b9[rc, cc] = base[rc, cc] + tap[:, valid[rc]][:, :, valid[cc]].sum((1, 2))
            c.bias8 = b9.reshape(9 * c.cout).to(torch.float32).contiguous().to(self.device)
            n += 1
        self.fp8 = n > 0
        self._pack_stage14_f8()
        return n

    def _pack_stage14_f8(self):
        """fp8 form of the 14x14 run (fr_conv_stage14_f8) when every conv of the run was switched to fp8: the e4m3 weights
        as one pre-swizzled stream + f32 [14][256] parameters per conv (oscale, 1 / oscale, nine border-class biases,
        PReLU slope, the NEXT conv's input centre mu and 1 / sx: a conv's epilogue writes its consumer's codes)."""
        self.stage14_f8 = None
        st = self.stage14
        if st is None or not self.fp8:
            return
        run = [self.blocks[st["first"] + k] for k in range(st["n"])]
        convs = [c for c1, c2, _ in run for c in (c1, c2)]
        if any(c.oscale is None or c.mu is None for c in convs):
            return
        after = self.blocks[st["first"] + st["n"]][0] if st["first"] + st["n"] < len(self.blocks) else None
        if after is not None and (after.oscale is None or after.cin != 256):
            after = None                                  # the run's consumer is not an fp8 conv: the last codes are unused
        per = self.lib.fr_conv_stage14_f8_weight_bytes(1)
        rows = self.lib.fr_conv_stage14_f8_param_floats() // 256
        stream = torch.empty(len(convs) * per, dtype=torch.uint8, device=self.device)
        prm = torch.zeros((len(convs), rows, 256), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for k, c in enumerate(convs):
                self.lib.fr_conv_stage14_f8_pack(_lib.ptr(c.w8), _lib.ptr(stream[k * per:]), _lib.stream_ptr())
                nxt = convs[k + 1] if k + 1 < len(convs) else after
                prm[k, 0] = c.oscale
                prm[k, 1] = (1.0 / c.oscale.double()).to(torch.float32)
                prm[k, 2:11] = c.bias8.reshape(9, 256)
                prm[k, 11] = c.slope if c.slope is not None else 1.0
                if nxt is not None:
                    prm[k, 12] = nxt.mu
                    prm[k, 13, 0] = 1.0 / nxt.sx
                else:
                    prm[k, 13, 0] = 1.0
            torch.cuda.synchronize(self.device)
        self.stage14_f8 = {"w": stream, "prm": prm.contiguous()}

    def _run_stage14_f8(self, h, h8, B):
        st, f8 = self.stage14, self.stage14_f8
        y = torch.empty_like(h)
        args = (_lib.ptr(h8), _lib.ptr(h), _lib.ptr(y), _lib.ptr(f8["w"]), _lib.ptr(f8["prm"]), B, st["n"], _lib.stream_ptr())
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.lib.fr_conv_stage14_f8(*args)
            e1.record()
            self.profile.append(("conv_stage14_f8_kernel<0>", 2.0 * B * 196 * 256 * 2304 * 2 * st["n"], e0, e1))
        else:
            self.lib.fr_conv_stage14_f8(*args)
        return y

    def _calib_observe(self, c, x, H):
        """calibration forward: statistics of the tensor ``x`` (f16 NHWC) an fp8 candidate reads"""
        cal = self._calib
        if cal is None or id(c) not in cal["want"] or not f8_eligible(c, H) or c.w32 is None:
            return
        xf = x.float()
        mu = xf.mean(dim=(0, 1, 2)) if cal["centre"] else None
        xc = xf - mu if mu is not None else xf
        st = {"absmax": float(xc.abs().max()), "mu": mu.contiguous() if mu is not None else None, "wq": None}
        if cal["gptq"]:
            import torch.nn.functional as F
            P = F.unfold(xc.permute(0, 3, 1, 2), 3, padding=1)                  # [B, Cin*9 (ci, kh, kw), L]
            P = P.permute(1, 0, 2).reshape(P.shape[1], -1)
            hm = (P @ P.t()) / P.shape[1]
            del P
            perm = (torch.arange(c.cin, device=x.device)[None, :] * 9 + torch.arange(9, device=x.device)[:, None]).reshape(-1)
            hm = hm[perm][:, perm]                                               # (tap, ci): the column order of w32
            w = c.w32.to(x.device)
            sw = (w.abs().amax(dim=1) / F8_MAX).clamp_min(1e-30).contiguous()
            U = gptq_factor(hm)
            wd = w.double().contiguous()
            st["wq"] = torch.empty(w.shape, dtype=torch.float32, device=x.device)
            self.lib.fr_gptq_round_e4m3(_lib.ptr(wd), _lib.ptr(U), _lib.ptr(sw), _lib.ptr(st["wq"]), w.shape[0], w.shape[1],
                                        _lib.stream_ptr())
        cal["stats"][id(c)] = st

    def _conv_f8(self, x8, c, B, H, W, residual=None, want16=True, nxt=None):
        """``nxt``: the fp8 conv that reads this one's output (its fp8 copy is written centred and scaled for it)"""
        y16 = torch.empty((B, H, W, c.cout), dtype=torch.float16, device=self.device) if want16 else None
        y8 = torch.empty((B, H, W, c.cout), dtype=torch.uint8, device=self.device) if nxt is not None else None
        a = _lib.ConvF8Args(_lib.ptr(x8), _lib.ptr(c.w8), _lib.ptr(y16), _lib.ptr(y8), _lib.ptr(c.oscale),
                            _lib.ptr(c.bias8), _lib.ptr(c.slope), _lib.ptr(residual), B, H, W, c.cin, c.cout,
                            1, float(1.0 / nxt.sx) if nxt is not None else 0.0,
                            _lib.ptr(nxt.mu) if nxt is not None else None)
        if self.profile is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.lib.fr_conv_nhwc_f8(ctypes.byref(a), _lib.stream_ptr())
            e1.record()
            self.profile.append(("conv_halo_kernel<2, 13, %d, 1, 4, false, true, 4, true, 8, 0>" % (256 if H == 14 else 320),
                                 2.0 * B * H * W * c.cout * 9 * c.cin, e0, e1))
        else:
            self.lib.fr_conv_nhwc_f8(ctypes.byref(a), _lib.stream_ptr())
        return y16, y8

    def _quantise(self, x16, c):
        """fp8 input of conv ``c`` from an f16 tensor: (x - mu[channel]) / sx, saturating"""
        out = torch.empty(x16.shape, dtype=torch.uint8, device=self.device)
        if c.mu is not None:
            self.lib.fr_quantize_f16_f8_centred(_lib.ptr(x16), _lib.ptr(out), x16.numel(), c.cin, _lib.ptr(c.mu),
                                                float(1.0 / c.sx), _lib.stream_ptr())
        else:
            self.lib.fr_quantize_f16_f8(_lib.ptr(x16), _lib.ptr(out), x16.numel(), float(1.0 / c.sx), _lib.stream_ptr())
        return out

    def _count_flops(self):
        f, hw = 2 * 112 * 112 * 27 * 64, 112
        for c1, c2, sc in self.blocks:
            f += 2 * hw * hw * c1.cin * c1.cout * 9
            ho = hw // c2.stride
            f += 2 * ho * ho * c2.cin * c2.cout * 9
            if sc is not None:
                f += 2 * ho * ho * sc.cin * sc.cout
            hw = ho
        return f + 2 * 25088 * 512

    # ---- launches
    def _small_batch_splitk(self, c, B):
        """Small batches (single frames: a handful of faces) leave most CUs without an output tile and every block
        runs its whole K loop alone (measured: 16 faces 3.06 ms, 29 us per conv launch).  Their 3x3 convs are cut
        along K into slices that run side by side, followed by ``fr_conv_splitk_epilogue``.  The slice count
        depends on the layer and on the MODE only (B <= LOW_BATCH: slices of 3 K steps; B <= SMALL_BATCH: of 9), so
        results do not depend on the batch size inside a mode (between modes they differ by f32 summation order)."""
        if B > self.small_batch or c.k != 3 or c.cin % 64:
            return 1
        nk = 9 * c.cin // 64
        if nk < 18:                  # Cin = 64 (the 112x112 / 56x56 layers): one pass
            return 1
        if B <= self.low_batch:
            return -(-nk // 3)
        return min(8, nk // 9)

    def _inblock(self, c, B):
        """Up to eight faces (single frames): a 3x3 / stride-1 conv with >= 128 input channels is ONE launch that splits K
        among the sixteen waves of a workgroup (csrc/conv_inblock.hip) - the split-K form is two launches at their
        latency floor, 89 times per forward.  A mode of its own: inside it a face's embedding does not depend on its
        batch mates, against the other modes it differs by f32 summation order."""
        return (B <= self.inblock_batch and c.k == 3 and c.stride == 1 and c.pad == 1 and c.cin % 32 == 0
                and 128 <= c.cin <= 512 and c.cout % 32 == 0 and isinstance(c, _Conv))

    def _conv(self, x, c, B, H, W, residual=None, partial=None, splitk=1, y=None, x2=None):
        Ho = (H + 2 * c.pad - c.k) // c.stride + 1
        Wo = (W + 2 * c.pad - c.k) // c.stride + 1
        if partial is None and x2 is None and self.profile is None and self._inblock(c, B):
            if y is None:
                y = torch.empty((B, Ho, Wo, c.cout), dtype=torch.float16, device=self.device)
            a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(c.w), _lib.ptr(y), _lib.ptr(c.bias), _lib.ptr(c.slope), _lib.ptr(residual), None,
                              B, H, W, c.cin, c.cout, c.k, c.k, c.stride, c.pad, Ho, Wo, c.bias_mode, 1, None, 0)
            self.lib.fr_conv_inblock_f16(ctypes.byref(a), _lib.stream_ptr())
            return y, Ho, Wo
        sk = self._small_batch_splitk(c, B) if partial is None and self.profile is None else 1
        if sk > 1:
            if y is None:
                y = torch.empty((B, Ho, Wo, c.cout), dtype=torch.float16, device=self.device)
            M = B * Ho * Wo
            part = torch.empty((sk, M, c.cout), dtype=torch.float32, device=self.device)
            a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(c.w), None, None, None, None, _lib.ptr(part),
                              B, H, W, c.cin, c.cout, c.k, c.k, c.stride, c.pad, Ho, Wo, 0, sk, _lib.ptr(x2), getattr(c, "c2", 0))
            self.lib.fr_conv_nhwc_f16(ctypes.byref(a), _lib.stream_ptr())
            self.lib.fr_conv_splitk_epilogue(_lib.ptr(part), sk, M, c.cout, Ho, Wo, _lib.ptr(c.bias), c.bias_mode,
                                             _lib.ptr(c.slope), _lib.ptr(residual), _lib.ptr(y), _lib.stream_ptr())
            return y, Ho, Wo
        if partial is None and y is None:
            y = torch.empty((B, Ho, Wo, c.cout), dtype=torch.float16, device=self.device)
        if (self.use_walk64 and partial is None and x2 is None and H == W and H % 28 == 0 and B > self.small_batch and B not in WALK64_SKIP
                and getattr(c, "w64", None) is not None):
            args = (_lib.ptr(x), _lib.ptr(c.w64), _lib.ptr(y), _lib.ptr(c.bias), c.bias_mode, _lib.ptr(c.slope),
                    _lib.ptr(residual), B, H, c.cout, _lib.stream_ptr())
            if self.profile is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self.lib.fr_conv_walk64_f16(*args)
                e1.record()
                self.profile.append(("conv_walk64_kernel %dx%d -> %d" % (H, W, c.cout), 2.0 * B * H * W * c.cout * 576, e0, e1))
            else:
                self.lib.fr_conv_walk64_f16(*args)
            return y, Ho, Wo
        a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(c.w) if isinstance(c, _Conv) else None, _lib.ptr(y),
                          _lib.ptr(c.bias), _lib.ptr(c.slope), _lib.ptr(residual), _lib.ptr(partial),
                          B, H, W, c.cin, c.cout, c.k, c.k, c.stride, c.pad, Ho, Wo, c.bias_mode, splitk,
                          _lib.ptr(x2), getattr(c, "c2", 0) if x2 is not None else 0)
        if self.profile is not None:
            # HIP events on the stream the kernel is launched on (torch's current stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.lib.fr_conv_nhwc_f16(ctypes.byref(a), _lib.stream_ptr())
            e1.record()
            # mirror of the C dispatch (fr_conv_nhwc_f16 -> fr_conv_halo_try): halo kernel for 3x3/s1 body convs
            halo = None
            if c.k == 3 and c.stride == 1 and H == W and partial is None and c.cin % 64 == 0:
                if H in (7, 14, 28) and c.cout % 128 == 0:       # lean variant: BN = 128, two blocks per CU
                    halo = "conv_halo_kernel<2, 13, %d, 1, 4, false, true, 4, false, 8, 0>" % {7: 384, 14: 256, 28: 320}[H]
                elif H == 56 and c.cin == 64:
                    halo = "conv_halo_kernel<1, 14, 384, 1, 4, false, false, 4, false, 8, 0>"
                elif H == 112 and c.cin == 64 and c.cout == 64:
                    halo = "conv_halo_kernel<1, 14, 512, 1, 4, false, false, 4, false, 8, 0>"
            variant = halo or ("conv_stem_kernel<112>" if c.cin == 8 else
                               "conv_mfma_kernel<%d, false, true>" % (2 if c.cout % 128 == 0 else 1))
            kreal = 27 if c.cin == 8 else c.k * c.k * c.cin + (getattr(c, "c2", 0) if x2 is not None else 0)   # algorithmic K (stem: 3 real channels)
            self.profile.append((variant, 2.0 * B * Ho * Wo * c.cout * kreal, e0, e1))
        else:
            self.lib.fr_conv_nhwc_f16(ctypes.byref(a), _lib.stream_ptr())
        return y, Ho, Wo

    def forward(self, x, taps=None):
        assert x.dtype == torch.float16 and x.shape[1:] == (112, 112, 8) and x.is_contiguous()
        B = x.shape[0]
        emb = torch.empty((B, 512), dtype=torch.float32, device=self.device)
        normed = torch.empty_like(emb)
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_chunk):
                b1 = min(B, b0 + self.max_chunk)
                self._forward_chunk(x[b0:b1], emb[b0:b1], normed[b0:b1], taps)
        return emb, normed

    # ---- single frames: the whole conv stack as ONE C call over persistent buffers
    def _plan_partial_floats(self, B):
        """split-K scratch of the largest conv of a B-face forward (floats)"""
        n, hw = 1, 112
        for c1, c2, sc in self.blocks:
            n = max(n, self._small_batch_splitk(c1, B) * B * hw * hw * c1.cout)
            ho = hw // c2.stride
            n = max(n, self._small_batch_splitk(c2, B) * B * ho * ho * c2.cout)
            hw = ho
        return n

    def release_plans(self):
        """Drop every prepared single-frame plan and its per-stream buffers (they are otherwise kept for the life of
        the network: a captured HIP graph may replay them).  Call only when no captured graph of this network is
        alive - ``FaceAnalysis.enable_graphs(False)`` does, after dropping its graphs."""
        with self._plan_lock:
            self._plans.clear()
            self._plan_bufs.clear()
            self._plan_limit_logged = False

    def _plan(self, B):
        """Up to LOW_BATCH faces the forward is ~200 launches of a few microseconds each and the Python / ctypes work
        per launch (argument structs, allocations, stream look-ups) is what the GPU waits for.  The same launch
        sequence as ``_forward_chunk`` is laid out once per (B, stream) - four rotating activation buffers, one
        split-K scratch - and replayed by ``fr_conv_sequence``.  Per stream: launches on one stream run in order, so
        they can share the buffers; another stream gets its own."""
        sid = torch.cuda.current_stream(self.device).cuda_stream
        key = (B, sid)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        dev = self.device
        # buffers are per STREAM and sized for LOW_BATCH faces: every batch size of the mode lays its steps over them
        # (never freed: a captured HIP graph may hold them); past 8 streams, launch by launch
        shared = self._plan_bufs.get(sid)
        if shared is None:
            if len(self._plan_bufs) >= MAX_PLAN_STREAMS:
                if not self._plan_limit_logged:
                    self._plan_limit_logged = True
                    logging.getLogger(__name__).warning(
                        "IResNetHIP: single-frame plans exist for %d streams already; further streams run the conv stack "
                        "launch by launch (slower single-frame latency).  release_plans() frees them.", MAX_PLAN_STREAMS)
                return None
            shared = self._plan_bufs[sid] = (
                [torch.empty(self.low_batch * 112 * 112 * 64, dtype=torch.float16, device=dev) for _ in range(4)],
                torch.empty(self._plan_partial_floats(self.low_batch), dtype=torch.float32, device=dev))
        bufs, partial = shared
        steps, part_floats = [], 0

        def view(buf, Ho, Wo, c):
            return buf[:B * Ho * Wo * c].view(B, Ho, Wo, c)

        def add(c, x, y, H, W, residual=None, x2=None):
            nonlocal part_floats
            Ho = (H + 2 * c.pad - c.k) // c.stride + 1
            Wo = (W + 2 * c.pad - c.k) // c.stride + 1
            if x2 is None and self._inblock(c, B):
                steps.append((2, c, x, y, residual, H, W, Ho, Wo, 1, x2))
                return Ho, Wo
            sk = self._small_batch_splitk(c, B)
            if sk > 1:
                part_floats = max(part_floats, sk * B * Ho * Wo * c.cout)
            steps.append((1 if sk > 1 else 0, c, x, y, residual, H, W, Ho, Wo, sk, x2))
            return Ho, Wo

        free = list(bufs)
        h = free.pop()
        H, W = add(self.stem, None, h, 112, 112)                      # x (the crops) is patched in per call
        hc = 64
        for bi_, (c1, c2, sc) in enumerate(self.blocks):
            mid = free.pop()
            add(c1, view(h, H, W, hc), mid, H, W)
            short = view(h, H, W, hc)
            s_buf = None
            fz = self.fused_sc.get(bi_) if self.fuse_shortcut else None
            if sc is not None and fz is None:
                s_buf = free.pop()
                Ho, Wo = add(sc, view(h, H, W, hc), s_buf, H, W)
                short = view(s_buf, Ho, Wo, sc.cout)
            out = free.pop()
            if fz is not None:
                Ho, Wo = add(fz, view(mid, H, W, c1.cout), out, H, W, x2=short)
            else:
                Ho, Wo = add(c2, view(mid, H, W, c1.cout), out, H, W, residual=short)
            free += [b for b in (h, mid, s_buf) if b is not None]
            h, H, W, hc = out, Ho, Wo, c2.cout
        assert part_floats <= partial.numel()
        arr = (_lib.ConvStep * len(steps))()
        for st, (kind, c, x, y, residual, Hi, Wi, Ho, Wo, sk, x2) in zip(arr, steps):
            st.kind = kind
            st.args = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(c.w), _lib.ptr(y), _lib.ptr(c.bias), _lib.ptr(c.slope),
                                    _lib.ptr(residual), _lib.ptr(partial) if kind == 1 else None, B, Hi, Wi, c.cin, c.cout,
                                    c.k, c.k, c.stride, c.pad, Ho, Wo, c.bias_mode, sk, _lib.ptr(x2), c.c2 if x2 is not None else 0)
        plan = self._plans[key] = (arr, len(steps), view(h, H, W, hc), bufs, partial)
        return plan

    def _forward_chunk(self, x, emb, normed, taps):
        B = x.shape[0]
        plan = None
        if B <= self.low_batch and taps is None and self.profile is None and self._calib is None and not self.fp8:
            with self._plan_lock:
                plan = self._plan(B)
        if plan is not None:
            arr, n, h, _, _ = plan
            with self._plan_lock:                # the input pointer is patched into the shared step array
                arr[0].args.x = x.data_ptr()
                self.lib.fr_conv_sequence(arr, n, _lib.stream_ptr())
                self._fc(h, B, emb, normed)
            return
        h, H, W = self._conv(x, self.stem, B, 112, 112)
        if taps is not None:
            taps["stem"] = h
        li = 0
        h8 = None                                  # fp8 copy of h, scaled for the conv that will read it (or None)
        nb = len(self.blocks)
        use_stage = (self.stage14 is not None and self.use_stage14 and B >= STAGE14_MIN_BATCH and taps is None
                     and self._calib is None and (not self.fp8 or self.stage14_f8 is not None))
        # the 28x28 run: its leading blocks that hold no fp8 conv (enable_fp8 "accurate": all but the last two)
        s28_first = s28_n = 0
        if (self.stage28 is not None and self.use_stage28 and B >= STAGE28_MIN_BATCH and taps is None and self._calib is None):
            s28_first = self.stage28["first"]
            for c1, c2, _ in self.blocks[s28_first:s28_first + self.stage28["n"]]:
                if self.fp8 and (c1.oscale is not None or c2.oscale is not None):
                    break
                s28_n += 1
            if s28_n < 2:
                s28_n = 0
        for bi_, (c1, c2, sc) in enumerate(self.blocks):
            if s28_first <= bi_ < s28_first + s28_n:
                if bi_ == s28_first:
                    h = self._run_stage28(h, B, s28_n)
                continue
            if use_stage and self.stage14["first"] <= bi_ < self.stage14["first"] + self.stage14["n"]:
                if bi_ == self.stage14["first"]:       # all n blocks in one launch; the loop skips the rest of the run
                    if self.fp8:
                        h = self._run_stage14_f8(h, h8 if h8 is not None else self._quantise(h, c1), B)
                        h8 = None                      # the run's consumer quantises the f16 output itself
                    else:
                        h = self._run_stage14(h, B)
                continue
            self._calib_observe(c1, h, H)              # enable_fp8(): statistics of the tensors the candidates read
            f1 = self.fp8 and c1.oscale is not None
            f2 = self.fp8 and c2.oscale is not None
            mid8 = None
            if f1:
                if h8 is None:
                    h8 = self._quantise(h, c1)
                mid, mid8 = self._conv_f8(h8, c1, B, H, W, want16=not f2 or taps is not None, nxt=c2 if f2 else None)
            else:
                mid, _, _ = self._conv(h, c1, B, H, W)
            fz = self.fused_sc.get(bi_) if self.fuse_shortcut and not f2 else None
            if sc is not None:
                li += 1
                if taps is not None:
                    taps[f"layer{li}.0.mid"] = mid
                short = None if fz is not None else self._conv(h, sc, B, H, W)[0]
            else:
                short = h
            self._calib_observe(c2, mid, H)
            if fz is not None:                         # stride-2 conv + 1x1 shortcut of the block input as one implicit GEMM
                h, H, W = self._conv(mid, fz, B, H, W, x2=h)
                h8 = None
            elif f2:
                if mid8 is None:
                    mid8 = self._quantise(mid, c2)
                nxt = self.blocks[bi_ + 1][0] if bi_ + 1 < nb else None
                nf1 = nxt is not None and nxt.oscale is not None
                h, h8 = self._conv_f8(mid8, c2, B, H, W, residual=short, want16=True, nxt=nxt if nf1 else None)
            else:
                h, H, W = self._conv(mid, c2, B, H, W, residual=short)
                h8 = None
            if taps is not None:
                taps[f"_block{len(taps)}"] = None
        if taps is not None:
            taps["final"] = h
        self._fc(h, B, emb, normed)

    def _fc(self, h, B, emb, normed):
        # FC as a 1x1 conv over a 1x1 image with Cin = 25088, split-K -> f32 partials
        partial = torch.empty((FC_SPLITK, B, 512), dtype=torch.float32, device=self.device)
        fc = _FC(self.fc_w)
        self._conv(h, fc, B, 1, 1, partial=partial, splitk=FC_SPLITK)
        self.lib.fr_fc_reduce_l2norm(_lib.ptr(partial), FC_SPLITK, B, 512, _lib.ptr(self.fc_bias),
                                     _lib.ptr(emb), _lib.ptr(normed), _lib.stream_ptr())


class _FC(_Conv):
    def __init__(self, w):
        self.w, self.bias, self.slope = w, None, None
        self.cin, self.cout, self.k, self.stride, self.pad, self.bias_mode = 25088, 512, 1, 1, 0, 0
