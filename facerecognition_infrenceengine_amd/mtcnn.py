"""MTCNN cascade on the HIP detector kernels.

Detector half of ``FaceAnalysis.get`` (/root/reference/infrenceServer.py:528).  The host side is
a launch plan only: every stage works on fixed-capacity per-frame slot lists with device-side
counts, so a whole batch of frames runs without a host synchronisation.  Conventions (resize,
ordering, thresholds, capacities) are those written down in ``oracle/detect.py``.
"""
import contextlib
import ctypes
import logging
import math
import threading

import torch

from . import _lib


def pyramid_scales(h, w, minsize=20, factor=0.709):
    m = 12.0 / minsize
    minl = min(h, w) * m
    scales, k = [], 0
    while minl >= 12:
        scales.append(m * factor ** k)
        minl *= factor
        k += 1
    return scales


def _pool_out(n, k, s):
    o = -(-(n - k) // s) + 1
    if (o - 1) * s >= n:
        o -= 1
    return o


class _DConv:
    """Direct-conv layer: weights [KH][KW][Cin][CoutP] f32, bias/slope [CoutP]."""

    def __init__(self, w, b, slope, device, pool2=False, head=None):
        cout, cin, kh, kw = w.shape
        gran = 32 if cout % 32 == 0 else 16
        coutp = -(-cout // gran) * gran
        wp = torch.zeros((kh, kw, cin, coutp), dtype=torch.float32)
        wp[..., :cout] = w.permute(2, 3, 1, 0)
        bp = torch.zeros(coutp); bp[:cout] = b
        self.w = wp.contiguous().to(device)
        self.b = bp.to(device)
        self.slope = None
        if slope is not None:
            sp = torch.zeros(coutp); sp[:cout] = slope
            self.slope = sp.to(device)
        self.cin, self.cout, self.coutp, self.kh, self.kw, self.pool2 = cin, cout, coutp, kh, kw, pool2
        self.head_w = self.head_b = None
        self.nhead = 0
        if head is not None:                       # (w [nh, cout], b [nh])
            self.head_w = head[0].t().contiguous().to(torch.float32).to(device)      # [cout][nh]
            self.head_b = head[1].to(torch.float32).contiguous().to(device)
            self.nhead = head[0].shape[0]

    def out_hw(self, h, w):
        hc, wc = h - self.kh + 1, w - self.kw + 1
        return ((hc + 1) // 2, (wc + 1) // 2) if self.pool2 else (hc, wc)


# layer id -> (cin, cout, kh, kw, ntb): must mirror the table in csrc/dconv_mfma.hip
MFMA_LAYERS = {0: (3, 12, 3, 3, 1), 1: (12, 16, 3, 3, 1), 2: (16, 32, 3, 3, 2),
               10: (4, 28, 3, 3, 2), 11: (28, 48, 3, 3, 3), 12: (48, 64, 2, 2, 4), 13: (64, 128, 3, 3, 4),
               14: (128, 6, 1, 1, 1),
               20: (4, 32, 3, 3, 1), 21: (32, 64, 3, 3, 2), 22: (64, 64, 3, 3, 4), 23: (64, 128, 2, 2, 4),
               24: (128, 256, 3, 3, 4), 25: (256, 16, 1, 1, 1)}


class _MConv:
    """Detector layer packed for fr_dconv_mfma_f32: w [cout_group][tap][CinP][CP] f32 where
    CinP = Cin rounded up to 4 and CP = NTB*16 (+16 when NTB is even: LDS bank spread)."""

    def __init__(self, layer, w, b, slope, device, head=None):
        cin, cout, kh, kw, ntb = MFMA_LAYERS[layer]      # packed sizes: real channels are zero-padded up to them
        rcout, rcin = w.shape[0], w.shape[1]
        assert tuple(w.shape[2:]) == (kh, kw) and rcout <= cout and rcin <= cin, (layer, tuple(w.shape))
        wfull = torch.zeros((cout, cin, kh, kw), dtype=torch.float32)
        wfull[:rcout, :rcin] = w
        w = wfull
        bfull = torch.zeros(cout); bfull[:rcout] = b; b = bfull
        if slope is not None:
            sfull = torch.zeros(cout); sfull[:rcout] = slope; slope = sfull
        self.layer, self.cin, self.cout, self.kh, self.kw = layer, cin, cout, kh, kw
        cinp = -(-cin // 4) * 4
        cp = ntb * 16 + (16 if ntb % 2 == 0 else 0)
        gsz = ntb * 16
        ngroups = -(-cout // gsz)
        wp = torch.zeros((ngroups, kh * kw, cinp, cp), dtype=torch.float32)
        wt = w.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)              # [tap][ci][co]
        for g in range(ngroups):
            n = min(gsz, cout - g * gsz)
            wp[g, :, :cin, :n] = wt[:, :, g * gsz:g * gsz + n]
        self.w = wp.contiguous().to(device)
        bp = torch.zeros(ngroups * gsz); bp[:cout] = b
        self.b = bp.to(device)
        self.slope = None
        if slope is not None:
            sp = torch.zeros(ngroups * gsz); sp[:cout] = slope
            self.slope = sp.to(device)
        self.head_w = self.head_b = None
        self.nhead = 0
        if head is not None:
            self.head_w = head[0].t().contiguous().to(torch.float32).to(device)      # [cout][nh]
            self.head_b = head[1].to(torch.float32).contiguous().to(device)
            self.nhead = head[0].shape[0]
        self.pool = {0: 2, 10: 3, 11: 3, 20: 3, 21: 3, 22: 2}.get(layer, 0)     # fused ceil-mode max pool, stride 2

    def out_hw(self, h, w):
        hc, wc = h - self.kh + 1, w - self.kw + 1
        return (_pool_out(hc, self.pool, 2), _pool_out(wc, self.pool, 2)) if self.pool else (hc, wc)


def _dense_as_conv(w, k, c):
    """MTCNN dense layer over a k x k x c map flattened (w, h, c) -> conv weight [o, c, kh, kw]."""
    o = w.shape[0]
    return w.reshape(o, k, k, c).permute(0, 3, 2, 1).contiguous()      # [o, w, h, c] -> [o, c, h, w]


class MTCNNHIP:
    # level streams of a single-frame call while it is captured into a HIP graph.  Round 3, final kernels (tools/bench_latency_streams.py,
    # get() + match under replay, 24 / 4 hardware queues): 0 streams 1.81 - 1.84 / 1.80 - 1.84 ms, 2: 1.67 - 1.69 / 1.66 - 1.74, 4: 1.59 - 1.65 /
    # 1.58 - 1.64, 11 (one per level, the earlier default): 1.70 - 1.74 / 1.57 - 1.66
    SINGLE_FRAME_LEVEL_STREAMS = 4
    def __init__(self, pstate, rstate, ostate, device="cuda:0", minsize=20, factor=0.709,
                 thresholds=(0.6, 0.7, 0.7), cap_scale=2048, keep_scale=256, cap_p=512, cap_r=64, cap_o=16,
                 fused_pnet=True, batch_min_pixels=None):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.minsize, self.factor, self.thresholds = minsize, factor, tuple(float(t) for t in thresholds)
        self.cap_scale, self.keep_scale, self.cap_p, self.cap_r, self.cap_o = cap_scale, keep_scale, cap_p, cap_r, cap_o
        assert cap_scale <= 4096 and cap_p <= 1024 and cap_r <= 1024 and cap_o <= 1024
        self._sides = {}
        self._tls = threading.local()      # per-thread launch stream: detect_batch is re-entrant across threads
        self.one_stream = False            # True (profiling): every pyramid level on the caller's stream, per-kernel times add up
        self.level_streams = 2             # side streams the pyramid levels 1.. are dealt over (detect_batch's default; round 4: 2 - see detect_batch)
        self.single_frame_level_streams = self.SINGLE_FRAME_LEVEL_STREAMS      # the same for calls of <= solo_max_frames frames (0: as level_streams)
        self.phase_marks = None            # tools: a list -> (name, event on the caller's stream) at the cascade's phase ends
        self.merged_level_nms = True       # the per-level NMS of ALL levels as one launch behind the pyramid (False: one per level)
        # From this many PIXELS per call (frames x H x W) the BATCH arithmetic runs: the band-only exact P-Net pass, P-Net conv1 /
        # R-Net / O-Net on the f16 matrix cores with split-precision operands + exact f32 passes at the thresholds.  Below it the
        # all-f32 detector (and, under ``solo_max_frames`` + 1 frames, its recorded call list).  The batch path's extra launches (tile
        # lists, exact chains) cost a fixed ~0.4 ms: measured crossover on MI355X between 8 and 12 x 1080p frames (8: 1.71 against
        # 1.51 ms, 12: 1.80 / 1.90, 16: 1.93 / 2.18, 32: 2.88 / 3.56; profiles/r05_detector_crossover.txt, tools/crossover_det.py),
        # i.e. ~ 22 Mpixel - 8 x 4K frames (66 Mpx) are far on the batch side, 8 x 1080p (16.6 Mpx) are not.
        self.batch_min_pixels = 22_000_000 if batch_min_pixels is None else int(batch_min_pixels)
        self.solo_max_frames = 7            # up to this many frames a call that is not a batch runs on ONE stream, eagerly, and is recorded
        d = self.device
        p, r, o = ({k: v.detach().float().cpu() for k, v in s.items()} for s in (pstate, rstate, ostate))
        self.p1 = _MConv(0, p["conv1.weight"], p["conv1.bias"], p["prelu1.weight"], d)
        self.p2 = _MConv(1, p["conv2.weight"], p["conv2.bias"], p["prelu2.weight"], d)
        hw = torch.cat([p["conv4_1.weight"].reshape(2, 32), p["conv4_2.weight"].reshape(4, 32)])
        hb = torch.cat([p["conv4_1.bias"], p["conv4_2.bias"]])
        self.p3 = _MConv(2, p["conv3.weight"], p["conv3.bias"], p["prelu3.weight"], d, head=(hw, hb))
        # fused conv2 -> conv3 -> heads on the f16 matrix cores (split precision) + exact f32 re-evaluation of every
        # cell that can pass the threshold (csrc/pnet_fused.hip): weights as (cout, tap, channel), 10 taps x 16 channels
        self.fused_pnet = bool(fused_pnet)
        w2p = torch.zeros((16, 10, 16)); w2p[:, :9, :10] = p["conv2.weight"].permute(0, 2, 3, 1).reshape(16, 9, 10)
        w3p = torch.zeros((32, 10, 16)); w3p[:, :9, :16] = p["conv3.weight"].permute(0, 2, 3, 1).reshape(32, 9, 16)
        self._p23 = tuple(t.to(torch.float32).contiguous().to(d) for t in (
            w2p, p["conv2.bias"], p["prelu2.weight"], w3p, p["conv3.bias"], p["prelu3.weight"], hw.t().contiguous(), hb))
        self.refine_margin = 2e-3           # in logit units, ~200x the split-precision error
        # The batch path (batch_min_pixels): only the cells within ``refine_margin`` of the face threshold are re-evaluated exactly (every
        # keep / reject decision is that of f32 arithmetic); kept cells above the band carry the split-precision heads (~2e-6 from
        # the f32 ones) - as the R-/O-Net crops do (``split_ro``).  False: every cell that can be kept carries the f32 path's bits.
        self.pnet_band = True
        self.split_pconv1 = True            # with pnet_band: conv1 on the f16 matrix cores too; the exact pass gets an exact f32 map
                                            # under its cells' windows from the f32 conv1 kernel run over just those tiles
        self.split_pconv1_min_px = 100000   # ... on levels whose conv1 map has at least this many pixels (the level pays four more
                                            # small launches for it.  64 x 1080p, detector alone, by the number of levels that take it:
                                            # 0: 4.74 ms, 1: 4.65, 2: 4.68, 3: 4.71, 4: 4.75, 6: 4.91, all 12: 5.26 - level 0 of a 1080p
                                            # pyramid (186 k pixels), levels 0 - 2 of a 4K one; profiles/r05_detector_crossover.txt)
        self.refined_cells = None           # optional device int32[1]: cells re-evaluated exactly (diagnostics)
        self.p23_all_heads = False          # True: the fused kernel also writes the approximate heads of the cells it rules out
        self.use_sequence = True            # eager single-frame calls of a known frame shape replay a recorded C call list (fr_detect_sequence)
        # first R-/O-Net layer fused with the crop (csrc/ro_conv1.hip): weights as [k = (kh, kw, channel)][cout]
        self.fused_crop = True
        self._rc1 = tuple(t.to(torch.float32).contiguous().to(d) for t in (
            r["conv1.weight"].permute(2, 3, 1, 0).reshape(27, 28), r["conv1.bias"], r["prelu1.weight"]))
        self._oc1 = tuple(t.to(torch.float32).contiguous().to(d) for t in (
            o["conv1.weight"].permute(2, 3, 1, 0).reshape(27, 32), o["conv1.bias"], o["prelu1.weight"]))
        # second R-/O-Net layer on the f16 matrix cores with split-precision operands (csrc/ro_conv2.hip; the batch path):
        # weights as [cout][tap][32 channels] f32; the crops whose logit lies within ``ro_margin`` of the stage threshold are
        # re-evaluated by the all-f32 layers, so every keep / reject decision is that of f32 arithmetic
        self.split_ro = True
        self.split_conv1 = True             # with split_ro: the first layer's conv on the f16 matrix cores too (csrc/ro_conv1.hip, F16)
        self.split_tail = True              # with split_ro: conv3 / dense4 (R-Net), conv4 / dense5 (O-Net) as split-precision GEMMs too
        self.ro_margin = 1e-3               # in logit units; the split path's measured head error is ~1e-6
        self.ro_list_cap = (1024, 256)      # slots of the exact pass's work list (R-Net, O-Net); entries past it keep the split values
        def c2w(w):
            o, c = w.shape[0], w.shape[1]
            wp = torch.zeros((o, 9, 32))
            wp[:, :, :c] = w.permute(0, 2, 3, 1).reshape(o, 9, c)
            return wp
        self._rc2 = tuple(t.to(torch.float32).contiguous().to(d) for t in (c2w(r["conv2.weight"]), r["conv2.bias"], r["prelu2.weight"]))
        self._oc2 = tuple(t.to(torch.float32).contiguous().to(d) for t in (c2w(o["conv2.weight"]), o["conv2.bias"], o["prelu2.weight"]))
        self.r1 = _MConv(10, r["conv1.weight"], r["conv1.bias"], r["prelu1.weight"], d)
        self.r2 = _MConv(11, r["conv2.weight"], r["conv2.bias"], r["prelu2.weight"], d)
        self.r3 = _MConv(12, r["conv3.weight"], r["conv3.bias"], r["prelu3.weight"], d)
        self.r4 = _MConv(13, _dense_as_conv(r["dense4.weight"], 3, 64), r["dense4.bias"], r["prelu4.weight"], d)
        self.r5 = _MConv(14, torch.cat([r["dense5_1.weight"], r["dense5_2.weight"]]).reshape(6, 128, 1, 1),
                         torch.cat([r["dense5_1.bias"], r["dense5_2.bias"]]), None, d)
        self.o1 = _MConv(20, o["conv1.weight"], o["conv1.bias"], o["prelu1.weight"], d)
        self.o2 = _MConv(21, o["conv2.weight"], o["conv2.bias"], o["prelu2.weight"], d)
        self.o3 = _MConv(22, o["conv3.weight"], o["conv3.bias"], o["prelu3.weight"], d)
        self.o4 = _MConv(23, o["conv4.weight"], o["conv4.bias"], o["prelu4.weight"], d)
        self.o5 = _MConv(24, _dense_as_conv(o["dense5.weight"], 3, 128), o["dense5.bias"], o["prelu5.weight"], d)
        # the small tail layers as split-precision GEMMs on the f16 matrix cores (csrc/ro_gemm.hip; batch path): f32 weights
        # [cout][K], K = (kh, kw, channel) ascending = the input map's memory order, packed into MFMA fragment order
        self._gemm = {}
        for lid, wconv, b, sl in ((12, r["conv3.weight"], r["conv3.bias"], r["prelu3.weight"]),
                                  (13, _dense_as_conv(r["dense4.weight"], 3, 64), r["dense4.bias"], r["prelu4.weight"]),
                                  (22, o["conv3.weight"], o["conv3.bias"], o["prelu3.weight"]),
                                  (23, o["conv4.weight"], o["conv4.bias"], o["prelu4.weight"]),
                                  (24, _dense_as_conv(o["dense5.weight"], 3, 128), o["dense5.bias"], o["prelu5.weight"])):
            wk = wconv.permute(0, 2, 3, 1).reshape(wconv.shape[0], -1).to(torch.float32).contiguous().to(d)
            packed = torch.empty(self.lib.fr_ro_gemm_weight_bytes(lid), dtype=torch.uint8, device=d)
            with torch.cuda.device(d):
                self.lib.fr_ro_gemm_pack(lid, _lib.ptr(wk), _lib.ptr(packed), _lib.stream_ptr())
                torch.cuda.synchronize(d)
            self._gemm[lid] = (packed, b.to(torch.float32).contiguous().to(d), sl.to(torch.float32).contiguous().to(d))
        self.o6 = _MConv(25, torch.cat([o["dense6_1.weight"], o["dense6_2.weight"], o["dense6_3.weight"]]).reshape(16, 256, 1, 1),
                         torch.cat([o["dense6_1.bias"], o["dense6_2.bias"], o["dense6_3.bias"]]), None, d)

    def set_exact(self, on=True):
        """All-exact switch of the batch path: with ``on`` every value a kept candidate carries - P-Net cell scores / regressions,
        R-/O-Net heads - is the all-f32 kernels', bit for bit (pnet_band and split_ro off: the exact P-Net pass re-evaluates every
        cell that can be kept, R-/O-Net run their f32 layers), so that NMS ORDER, box truncation and crop coordinates are those of
        f32 arithmetic too, not only the threshold decisions.  Costs the batch path its round-4 gain (64 x 1080p: 4.9 -> 6.5 ms).
        Default off: threshold decisions exact, kept values within ~5e-6 (DESIGN.md 4.3a says what that can and cannot change)."""
        self.pnet_band = self.split_ro = not on
        return self

    @property
    def _dl(self):
        return getattr(self._tls, "dl", (None, 0.0))

    @_dl.setter
    def _dl(self, v):
        self._tls.dl = v

    @property
    def _s(self):
        return self._tls.s

    @_s.setter
    def _s(self, v):
        self._tls.s = v

    # ---- thin launch helpers (all on the current stream)
    def _new(self, shape, dtype):
        """A work tensor.  Inside an eager single-frame call (``detect_batch`` sets ``_tls.cache``) the k-th allocation of a
        call returns the tensor the k-th allocation of the previous call with the same frame shape, stream and thread
        made: ~90 ``torch.empty`` per call were 0.15 ms of an interpreter-bound 0.73 ms."""
        c = getattr(self._tls, "cache", None)
        if c is None:
            return torch.empty(shape, dtype=dtype, device=self.device)
        lst, i = c
        c[1] = i + 1
        if i < len(lst) and lst[i].shape == tuple(shape) and lst[i].dtype == dtype:
            return lst[i]
        t = torch.empty(shape, dtype=dtype, device=self.device)
        self._tls.cache_grew = True             # a recorded call list of this frame shape would hold stale pointers
        if i < len(lst):
            lst[i] = t
        else:
            lst.append(t)
        return t

    def _f32(self, *shape):
        return self._new(shape, torch.float32)

    def _i32(self, *shape):
        return self._new(shape, torch.int32)

    def _fptr(self, frames):
        """The frame batch's pointer, tagged with its role for the call recorder (_lib.RolePtr): a replayed call list
        patches exactly the slots recorded through here."""
        return _lib.ptr(frames, role="frame")

    def _dconv(self, x, c, B, H, W, frames=None, counts=None, cap=0, y_split=None):
        ho, wo = c.out_hw(H, W)
        y = self._f32(B, ho, wo, c.nhead if c.nhead else c.cout)
        if isinstance(c, _MConv):
            fh, fw = (frames.shape[1], frames.shape[2]) if frames is not None else (0, 0)
            self.lib.fr_dconv_mfma_f32(c.layer, _lib.ptr(x), _lib.ptr(c.w), _lib.ptr(c.b), _lib.ptr(c.slope), _lib.ptr(y),
                                       B, H, W, _lib.ptr(c.head_w), _lib.ptr(c.head_b), self._fptr(frames), fh, fw,
                                       _lib.ptr(counts), cap, _lib.ptr(y_split), self._s)
        else:
            self.lib.fr_dconv_f32(_lib.ptr(x), _lib.ptr(c.w), _lib.ptr(c.b), _lib.ptr(c.slope), _lib.ptr(y), B, H, W,
                                  c.cin, c.cout, c.coutp, c.kh, c.kw, 1 if c.pool2 else 0, _lib.ptr(c.head_w),
                                  _lib.ptr(c.head_b), c.nhead, self._s)
        return y, ho, wo

    def _gemm_split(self, lid, x, B, shape, counts, cap):
        w, b, sl = self._gemm[lid]
        y = self._f32(B, *shape)
        self.lib.fr_ro_gemm_split(lid, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(sl), _lib.ptr(y), B, _lib.ptr(counts), cap, self._s)
        return y

    def _pool(self, x, B, H, W, C, k, s):
        ho, wo = _pool_out(H, k, s), _pool_out(W, k, s)
        y = self._f32(B, ho, wo, C)
        self.lib.fr_maxpool_f32(_lib.ptr(x), _lib.ptr(y), B, H, W, C, k, s, self._s)
        return y, ho, wo

    def _nms(self, boxes, scores, aux, naux, counts, L, nseg, seg_cap, seg_major, thr, mode, keep, out=None, results=False):
        """results: the four outputs are what detect_batch returns - their pointers carry the roles "out0".."out3" for the
        call recorder."""
        if out is None:
            bo, so, co = self._f32(L, keep, 4), self._f32(L, keep), self._i32(L)
            ao = self._f32(L, keep, max(naux, 1))
        else:
            bo, so, ao, co = out
        po = [_lib.ptr(t, role="out%d" % i if results else None) for i, t in enumerate((bo, so, ao, co))]
        self.lib.fr_sort_nms(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(aux), naux, _lib.ptr(counts), L, nseg, seg_cap,
                             seg_major, thr, mode, keep, *po, keep, self._s)
        return bo, so, ao, co

    # ---- nets
    def pnet_level(self, frames, scale, trace=None, cand=None):
        """frames u8 [N,H,W,3] -> head f32 [N,hc,wc,6] of one pyramid level.
        cand (detect_batch, batches): (scale, thr, cap, boxes, scores, regs, counts) of the level's candidate list - with conv1 on the
        f16 matrix cores (``split_pconv1``) the level extracts its candidates itself (``_tls.level_done``)."""
        N, H, W, _ = frames.shape
        hs, ws = int(math.ceil(H * scale)), int(math.ceil(W * scale))
        # the pyramid level is resized inside P-Net conv1's tile load (no f32 level image in HBM)
        h, w = self.p1.out_hw(hs, ws)
        self._tls.level_done = False
        path = getattr(self._tls, "path", None)                    # what this call ran, for tests / diagnostics (detect_batch resets it)
        if self.fused_pnet and N * h * w * 64 < 2 ** 31:          # the split map is addressed with 32-bit buffer offsets
            t0 = self.thresholds[0]
            lt = math.log(t0 / (1.0 - t0))
            band = self.pnet_band and N * H * W >= self.batch_min_pixels and trace is None
            if path is not None:
                path["fused_levels"] += 1
                path["band_levels"] += int(band)
            xs = self._new((N, h, w, 64), torch.uint8)     # split-f16 copy of conv1's map
            if band and self.split_pconv1 and cand is not None and h * w >= self.split_pconv1_min_px:
                # conv1 on the f16 matrix cores -> fused conv2/3/heads -> the conv1 tiles under the band cells' windows, EXACTLY -> the
                # exact pass + the candidates (csrc/pnet_conv1.hip F16 / LIST, fr_pnet_band_tiles, fr_pnet_finish_levels)
                p1 = self.p1
                # the f32 map: SPARSE - written only in the 16 x 64-pixel tiles fr_pnet_band_tiles lists (where an exact 5x5 window
                # is needed), uninitialised elsewhere; read by the exact pass (fr_pnet_finish_levels) inside those windows only
                x = self._f32(N, h, w, 12)
                if path is not None:
                    path["pconv1_mfma_levels"].append((h, w))
                self.lib.fr_pnet_conv1_band(0, self._fptr(frames), N, H, W, hs, ws, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope),
                                            None, _lib.ptr(xs), None, None, 0, self._s)
                head = self._f32(N, h - 4, w - 4, 6)
                wsp = self._new((self.lib.fr_pnet23_workspace_bytes(N, h, w) // 4,), torch.float32)
                self.lib.fr_pnet23_split_f16(_lib.ptr(x), _lib.ptr(xs), N, h, w, *[_lib.ptr(t) for t in self._p23], _lib.ptr(head),
                                             (1 if self.p23_all_heads else 0) | 2, lt - self.refine_margin, lt + self.refine_margin,
                                             _lib.ptr(self.refined_cells), _lib.ptr(wsp), wsp.numel() * 4, self._s)
                nt = self.lib.fr_pnet_band_tiles_count(N, h, w)
                tbuf, tiles = self._i32(1 + (nt + 31) // 32), self._i32(nt)
                self.lib.fr_pnet_band_tiles(_lib.ptr(wsp), N, h, w, _lib.ptr(tbuf), _lib.ptr(tiles), self._s)
                self.lib.fr_pnet_conv1_band(1, self._fptr(frames), N, H, W, hs, ws, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope),
                                            _lib.ptr(x), None, _lib.ptr(tiles), _lib.ptr(tbuf), nt, self._s)
                sc, thr, cap, lb, ls, lr, lc = cand
                bc = self._i32(N * (-(-(h - 4) * (w - 4) // 256)))
                lv = (_lib.PnetLevel * 1)(_lib.PnetLevel(x.data_ptr(), head.data_ptr(), wsp.data_ptr(), h, w, float(sc), lb.data_ptr(),
                                                          ls.data_ptr(), lr.data_ptr(), lc.data_ptr(), bc.data_ptr()))
                self.lib.fr_pnet_finish_levels(lv, 1, N, *[_lib.ptr(t) for t in self._p23], thr, cap, lt - self.refine_margin,
                                               _lib.ptr(self.refined_cells), self._s)
                self._dl = (wsp, lt - self.refine_margin)
                self._tls.level_done = True
                self._tls.keep = (x, xs, head, wsp, tbuf, tiles, bc)      # alive until the stream has been joined (detect_batch)
                return head, h - 4, w - 4
            x, h, w = self._dconv(None, self.p1, N, hs, ws, frames=frames, y_split=xs)
            head = self._f32(N, h - 4, w - 4, 6)
            ws = self._new((self.lib.fr_pnet23_workspace_bytes(N, h, w) // 4,), torch.float32)
            self.lib.fr_pnet23_split_f16(_lib.ptr(x), _lib.ptr(xs), N, h, w, *[_lib.ptr(t) for t in self._p23], _lib.ptr(head),
                                         1 if (self.p23_all_heads or trace is not None) else 0,
                                         lt - self.refine_margin, lt + self.refine_margin if band else float("-inf"),
                                         _lib.ptr(self.refined_cells), _lib.ptr(ws), ws.numel() * 4, self._s)
            self._dl = (ws, math.log(t0 / (1.0 - t0)) - self.refine_margin)     # pre-filter for fr_pnet_candidates
            return head, h - 4, w - 4
        # A level too large for 32-bit offsets into the split map (detect_batch cuts batches so that this does not happen; what is
        # left is a single frame beyond ~ 8K x 16K): the three P-Net layers as generic f32 launches.  Same results, slower - said aloud.
        if self.fused_pnet and not getattr(self, "_unfused_logged", False):
            self._unfused_logged = True
            logging.getLogger(__name__).warning("MTCNNHIP: a %d x %d pyramid level of %d frame(s) exceeds the fused P-Net's 32-bit map "
                                                "offsets; it runs layer by layer (f32)", hs, ws, N)
        if path is not None:
            path["unfused_levels"] += 1
        self._dl = (None, 0.0)
        x, h, w = self._dconv(None, self.p1, N, hs, ws, frames=frames)
        x, h, w = self._dconv(x, self.p2, N, h, w)
        head, h, w = self._dconv(x, self.p3, N, h, w)
        return head, h, w

    def crop_conv1(self, net, frames, boxes, counts, cap):
        """frames u8 [N,H,W,3], boxes f32 [N,cap,4], counts i32 [N] -> the net's pooled conv1 map of every valid slot."""
        N, H, W, _ = frames.shape
        w, b, s = self._rc1 if net == 0 else self._oc1
        p, c = (11, 28) if net == 0 else (23, 32)
        y = self._f32(N * cap, p, p, c)
        self.lib.fr_crop_conv1_f32(net, self._fptr(frames), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, _lib.ptr(w),
                                   _lib.ptr(b), _lib.ptr(s), _lib.ptr(y), self._s)
        return y

    def crop_conv12_split(self, net, frames, boxes, counts, cap):
        """crop -> conv1 -> pool (map written as split f16) -> conv2 -> pool on the f16 matrix cores: the net's pooled conv2
        map of every valid slot, f32 [N*cap, 4, 4, 48] / [N*cap, 10, 10, 64]."""
        N, H, W, _ = frames.shape
        w1, b1, s1 = self._rc1 if net == 0 else self._oc1
        w2, b2, s2 = self._rc2 if net == 0 else self._oc2
        p1, p2, c2 = (11, 4, 48) if net == 0 else (23, 10, 64)
        xs = self._new((N * cap, p1 * p1, 128), torch.uint8)
        self.lib.fr_crop_conv1_split(net, self._fptr(frames), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, _lib.ptr(w1),
                                     _lib.ptr(b1), _lib.ptr(s1), _lib.ptr(xs), 1 if self.split_conv1 else 0, self._s)
        y = self._f32(N * cap, p2, p2, c2)
        lc = self._i32(1)                   # the exact pass's list counter: cleared by the conv2 kernel, filled by fr_ro_margin_list
        self.lib.fr_ro_conv2_split(net, _lib.ptr(xs), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(s2), _lib.ptr(y), N * cap,
                                   _lib.ptr(counts), cap, _lib.ptr(lc), self._s)
        return y, lc

    def exact_pass(self, net, frames, boxes, counts, cap, head, thr, lc):
        """The split-precision heads ``head`` [N*cap, 6 | 16] of the crops whose logit difference lies within ``ro_margin`` of
        the stage threshold are replaced by those of the all-f32 layers (compact work list, device-side count: no sync)."""
        N, H, W, _ = frames.shape
        w1, b1, s1 = self._rc1 if net == 0 else self._oc1
        lcap = self.ro_list_cap[net]
        nh = head.shape[1]
        lst = self._i32(lcap)
        self.lib.fr_ro_margin_list(_lib.ptr(head), nh, _lib.ptr(counts), N, cap, math.log(thr / (1.0 - thr)), self.ro_margin,
                                   _lib.ptr(lst), _lib.ptr(lc), lcap, self._s)
        p, c = (11, 28) if net == 0 else (23, 32)
        x1 = self._f32(lcap, p, p, c)
        self.lib.fr_crop_conv1_list_f32(net, self._fptr(frames), N, H, W, _lib.ptr(boxes), cap, _lib.ptr(lst), _lib.ptr(lc), lcap,
                                        _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(s1), _lib.ptr(x1), self._s)
        exact = (self.rnet if net == 0 else self.onet)(None, lcap, lc, lcap, x1=x1)
        self.lib.fr_ro_scatter_rows(_lib.ptr(exact), _lib.ptr(lst), _lib.ptr(lc), lcap, nh, _lib.ptr(head), self._s)
        self._ro_lists = getattr(self, "_ro_lists", {})
        self._ro_lists[net] = lc            # diagnostics / tests: how many crops the exact pass took

    def rnet(self, x, B, counts=None, cap=0, x1=None, x2=None):
        """counts / cap: only the first counts[frame] of a frame's cap crop slots are computed (device-side).
        x1: conv1's pooled map when it was computed straight from the frames (crop_conv1); x2: conv2's (crop_conv12_split)."""
        k = dict(counts=counts, cap=cap)
        if x2 is not None:
            x, h, w = x2, 4, 4
        elif x1 is not None:
            x, h, w = x1, 11, 11
        else:
            x, h, w = self._dconv(x, self.r1, B, 24, 24, **k)   # + fused 3x3/s2 pool -> 11x11
        if x2 is None:
            x, h, w = self._dconv(x, self.r2, B, h, w, **k)     # + fused 3x3/s2 pool -> 4x4
        if x2 is not None and self.split_tail:               # the batch path: conv3 / dense4 as split-precision GEMMs
            x = self._gemm_split(12, x, B, (3, 3, 64), counts, cap)
            x = self._gemm_split(13, x, B, (1, 1, 128), counts, cap)
        else:
            x, h, w = self._dconv(x, self.r3, B, h, w, **k)
            x, h, w = self._dconv(x, self.r4, B, h, w, **k)
        x, h, w = self._dconv(x, self.r5, B, 1, 1, **k)
        return x.reshape(B, 6)

    def onet(self, x, B, counts=None, cap=0, x1=None, x2=None):
        k = dict(counts=counts, cap=cap)
        if x2 is not None:
            x, h, w = x2, 10, 10
        elif x1 is not None:
            x, h, w = x1, 23, 23
        else:
            x, h, w = self._dconv(x, self.o1, B, 48, 48, **k)   # + fused 3x3/s2 pool -> 23x23
        if x2 is None:
            x, h, w = self._dconv(x, self.o2, B, h, w, **k)     # + fused 3x3/s2 pool -> 10x10
        if x2 is not None and self.split_tail:
            x = self._gemm_split(22, x, B, (4, 4, 64), counts, cap)     # conv3 + fused 2x2/s2 pool -> 4x4
            x = self._gemm_split(23, x, B, (3, 3, 128), counts, cap)
            x = self._gemm_split(24, x, B, (1, 1, 256), counts, cap)
        else:
            x, h, w = self._dconv(x, self.o3, B, h, w, **k)     # + fused 2x2/s2 pool -> 4x4
            x, h, w = self._dconv(x, self.o4, B, h, w, **k)
            x, h, w = self._dconv(x, self.o5, B, h, w, **k)
        x, h, w = self._dconv(x, self.o6, B, 1, 1, **k)
        return x.reshape(B, 16)

    def _mark(self, name):
        if self.phase_marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.phase_marks.append((name, e))

    # ---- cascade
    def detect_batch(self, frames, trace=None, level_streams=None, _out=None, _chunk=False):
        """frames: uint8 [N,H,W,3] BGR device tensor (contiguous).

        level_streams: side HIP streams (1 or 2) the pyramid levels 1.. are dealt over; level 0 stays on the caller's
        stream.  Default ``self.level_streams`` = 2 since round 4 (split-precision R-/O-Net, band-only exact P-Net pass: a 64 x 1080p
        batch alone 5.30 ms with one side stream, 4.9 - 5.1 with two; inside the bench C2 24 150 -> 25 400 faces/s, C5 25 470 -> 27 070, C3
        20 780 -> 20 880: tools/ab_bench_knobs.sh).  Round 3 had measured: a 64 x 1080p batch ALONE (tools/bench_det_phases.py):
        one side stream + an NMS launch per level 6.80 ms, two side streams 6.47, one launch for the NMS of all levels
        (``merged_level_nms``, the default: 768 one-workgroup sorts at once instead of twelve launches of 64) 6.54 / 6.37 ms.
        Inside the bench, where the detector shares the GPU with the embedder: 64 x 1080p the four combinations are within
        the run-to-run noise (12.3 - 12.7 ms/step); 8 x 4K frames (config C3) 6.10 ms/step with one side stream + merged NMS,
        6.30 with the per-level launches, 6.6 - 6.9 with two side streams - hence the defaults.

        Returns device tensors: boxes f32 [N,cap_o,4], scores f32 [N,cap_o], kps f32 [N,cap_o,5,2],
        counts i32 [N] (faces per frame, in descending-score order).

        ``_tls.path`` (this thread's last call): which arithmetic ran - {"frames", "batch", "chunks", "fused_levels", "band_levels",
        "pconv1_mfma_levels" [(h, w) of the conv1 maps computed on the f16 matrix cores], "unfused_levels", "split_ro"} - so that a
        test can assert the path it means to test was the one taken."""
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3 and frames.is_contiguous()
        N, H, W, _ = frames.shape
        lib, t0, t1, t2 = self.lib, *self.thresholds
        batch = N * H * W >= self.batch_min_pixels
        few = N <= self.solo_max_frames and not batch
        if not _chunk:
            self._tls.path = {"frames": N, "batch": batch, "chunks": 1, "fused_levels": 0, "band_levels": 0, "pconv1_mfma_levels": [],
                              "unfused_levels": 0, "split_ro": False}
        # The fused P-Net addresses a level's split conv1 map [N, h, w, 64 B] with 32-bit offsets.  A batch whose largest level
        # exceeds them (64 x 4K frames: 3.05e9 B) is cut into the fewest equal groups of frames that fit - frames are independent, every
        # group's final NMS writes its rows of the result tensors - rather than dropping that level to the layer-by-layer f32 path.
        if self.fused_pnet and trace is None and not _chunk and N > 1:
            sc0 = pyramid_scales(H, W, self.minsize, self.factor)
            if sc0:
                h0, w0 = self.p1.out_hw(int(math.ceil(H * sc0[0])), int(math.ceil(W * sc0[0])))
                fit = (2 ** 31 - 1) // max(h0 * w0 * 64, 1)
                if 1 <= fit < N:
                    ng = -(-N // fit)
                    per = -(-N // ng)
                    with torch.cuda.device(self.device):
                        out = (torch.empty((N, self.cap_o, 4), dtype=torch.float32, device=self.device),
                               torch.empty((N, self.cap_o), dtype=torch.float32, device=self.device),
                               torch.empty((N, self.cap_o, 14), dtype=torch.float32, device=self.device),
                               torch.empty((N,), dtype=torch.int32, device=self.device))
                    self._tls.path["chunks"] = ng
                    for n0 in range(0, N, per):
                        n1 = min(N, n0 + per)
                        self.detect_batch(frames[n0:n1], None, level_streams, _out=tuple(t[n0:n1] for t in out), _chunk=True)
                    return out[0], out[1], out[2][..., 4:14].unflatten(-1, (5, 2)), out[3]
        with torch.cuda.device(self.device):
            self._s = _lib.stream_ptr()
            self._tls.cache = None              # (a call that raised may have left these set)
            self._tls.cache_grew = False
            self.lib.stop_recording()
            self._mark("start")
            scales = pyramid_scales(H, W, self.minsize, self.factor)
            nlev = len(scales)
            if nlev == 0:                       # frame smaller than one 12-px cell at the coarsest usable scale
                z = torch.zeros((N, self.cap_o, 14), dtype=torch.float32, device=self.device)
                return (z[..., :4].contiguous(), z[..., 0].contiguous(), z[..., 4:14].unflatten(-1, (5, 2)),
                        torch.zeros(N, dtype=torch.int32, device=self.device))
            assert nlev * self.keep_scale <= 4096, "too many pyramid levels for the merged NMS list"
            cs = self.cap_scale
            ksz = self.keep_scale
            # Level 0 holds half of the pyramid's pixels; the remaining levels are small launches that cannot fill
            # 256 CUs on their own, so they are dealt round-robin over side HIP streams beside level 0 (joined before
            # the NMS).
            main = torch.cuda.current_stream()
            sides = self._sides.get(main.cuda_stream)         # side streams per caller stream: independent
            if sides is None:                                 # pipelines (bench --pipes) do not couple through them
                sides = self._sides[main.cuda_stream] = [torch.cuda.Stream(device=self.device) for _ in range(2)]
            nside = max(1, min(2, level_streams if level_streams is not None else self.level_streams))
            if few and level_streams is None and self.single_frame_level_streams > 0 and torch.cuda.is_current_stream_capturing():
                # A single frame is a chain of launch latencies.  While a HIP graph is being captured, every level goes to a
                # stream of its own: the graph then holds the levels' five-kernel chains side by side (640x480 get() + match
                # under replay 2.21 -> 2.01 ms).  Not in eager calls: the host issues the launches one by one anyway and the
                # extra fork / join events cost it 0.1 ms.
                nside = min(self.single_frame_level_streams, max(1, nlev - 1))
                while len(sides) < nside:
                    sides.append(torch.cuda.Stream(device=self.device))
            sides = sides[:nside]
            # An EAGER single-frame call is bound by the interpreter (host issue 0.73 ms against 0.85 ms until the GPU is done,
            # tools/host_time_c1.py): side streams would only add their fork / join events and a stream switch per level
            solo = trace is not None or self.one_stream or (few and level_streams is None and not torch.cuda.is_current_stream_capturing())
            record = False
            if solo and trace is None and few and _out is None and not torch.cuda.is_current_stream_capturing():
                caches = self._tls.__dict__.setdefault("caches", {})
                seqs = self._tls.__dict__.setdefault("seqs", {})
                key = (N, H, W, main.cuda_stream)
                if key not in caches and len(caches) >= 4:
                    old_key = next(iter(caches))                    # oldest frame shape of this thread
                    caches.pop(old_key); seqs.pop(old_key, None)
                known = key in caches
                self._tls.cache = [caches.setdefault(key, []), 0]
                # From the second call of a frame shape on every work tensor is the cached one of the call before, so the call
                # is the same list of C calls with the same arguments - but for the frame and the four result tensors.  It is
                # recorded once (on the second call: the first filled the cache) and replayed by ONE C call from the third on
                # (fr_detect_sequence): the interpreter's ~50 ctypes calls were what an eager single-frame call waited for.
                cfg = (self.fused_pnet, self.fused_crop, self.merged_level_nms, self.thresholds, self.p23_all_heads, self.refine_margin,
                       self.cap_scale, self.keep_scale, self.cap_p, self.cap_r, self.cap_o, self.minsize, self.factor)
                # Only the default configuration is recorded: fr_detect_sequence replays the entry points of _lib.SEQ_FN, and
                # the stand-alone crop / the f32 P-Net layers on generic shapes go through others (the recorder refuses
                # such a list as well: Lib.stop_recording); ``one_stream`` asks for every level on the caller's stream,
                # which a recorded list (levels on side streams) would not honour.
                seq_ok = (self.use_sequence and self.phase_marks is None and self.refined_cells is None and self.fused_pnet
                          and self.fused_crop and not self.one_stream)
                seq = seqs.get(key)
                if seq is not None and (not seq_ok or seq["cfg"] != cfg):
                    seqs.pop(key)                  # recorded under another configuration: never replayed again
                    seq = None
                if seq_ok:
                    if seq is not None:
                        self._tls.cache = None
                        return self._replay(seq, frames)
                    record = known
                    if record:
                        lib.start_recording()
                        # The recorded call deals the pyramid levels 1.. over side streams, forked from and joined to the
                        # caller's stream by events of its own (the call list carries their record / wait): replayed by one
                        # C call the host is no longer what the GPU waits for, the serial chain of ten levels is
                        nrec = min(4, max(1, nlev - 1))
                        while len(self._sides[main.cuda_stream]) < nrec:
                            self._sides[main.cuda_stream].append(torch.cuda.Stream(device=self.device))
                        rec_sides = self._sides[main.cuda_stream][:nrec]
                        rec_events = [torch.cuda.Event() for _ in range(nrec + 1)]
                        rec_events[0].record(main)
                        lib.note(8, rec_events[0].cuda_event, main.cuda_stream)
                        for side in rec_sides:
                            side.wait_event(rec_events[0])
                            lib.note(9, side.cuda_stream, rec_events[0].cuda_event)
            lb, ls, lr, lc = self._f32(nlev, N, cs, 4), self._f32(nlev, N, cs), self._f32(nlev, N, cs, 4), self._i32(nlev, N)
            kb, ks, ka, kc = self._f32(nlev, N, ksz, 4), self._f32(nlev, N, ksz), self._f32(nlev, N, ksz, 4), self._i32(nlev, N)
            if not solo:
                for side in sides:
                    side.wait_stream(main)
            keep = []
            for li, s in enumerate(scales):
                side = sides[(li - 1) % len(sides)] if li else sides[0]
                with (contextlib.nullcontext() if solo else torch.cuda.stream(main if li == 0 else side)):
                    if not solo:
                        self._s = _lib.stream_ptr()
                    elif record:
                        self._s = ctypes.c_void_p((main if li == 0 else rec_sides[(li - 1) % len(rec_sides)]).cuda_stream)
                    head, hc, wc = self.pnet_level(frames, s, trace, cand=None if (trace is not None or not batch) else
                                                   (float(s), t0, cs, lb[li], ls[li], lr[li], lc[li]))
                    nblk = -(-hc * wc // 256)
                    bc = self._i32(N * nblk)
                    prob = self._f32(N, hc, wc) if trace is not None else None
                    dl, dl_min = self._dl
                    if self._tls.level_done:                    # the level extracted its candidates itself
                        keep.append(self._tls.keep)
                        self._tls.keep = None
                    else:
                        lib.fr_pnet_candidates(_lib.ptr(head), N, hc, wc, float(s), t0, cs, _lib.ptr(lb[li]), _lib.ptr(ls[li]),
                                               _lib.ptr(lr[li]), _lib.ptr(lc[li]), _lib.ptr(bc), _lib.ptr(prob), _lib.ptr(dl),
                                               dl_min, self._s)
                    # per-level NMS 0.5 -> keep_scale survivors: one launch for all levels behind the loop (default), or
                    # right behind the level's own kernels on the level's stream (merged_level_nms False, batches only)
                    if not few and not self.merged_level_nms:
                        self._nms(lb[li], ls[li], lr[li], 4, lc[li], N, 1, cs, 0, 0.5, 0, ksz, out=(kb[li], ks[li], ka[li], kc[li]))
                    if trace is not None:
                        trace.setdefault("pnet_head", []).append(head)
                        trace.setdefault("pnet_prob", []).append(prob)
            if not solo:
                for side in sides:
                    main.wait_stream(side)
                self._s = _lib.stream_ptr()
            elif record:
                for side, ev in zip(rec_sides, rec_events[1:]):
                    ev.record(side)
                    lib.note(8, ev.cuda_event, side.cuda_stream)
                    main.wait_event(ev)
                    lib.note(9, main.cuda_stream, ev.cuda_event)
                self._s = ctypes.c_void_p(main.cuda_stream)
            for grp in keep:                                    # allocated on a level stream, must outlive the kernels queued there
                for t in grp:
                    t.record_stream(main)
            keep = None                                         # (hundreds of MB at level 0 of 64 x 1080p: back to the allocator now)
            self._mark("pnet")
            if few or self.merged_level_nms:
                self._nms(lb, ls, lr, 4, lc, nlev * N, 1, cs, 0, 0.5, 0, ksz, out=(kb, ks, ka, kc))
            # cross-level NMS 0.7 -> cap_p
            b1, s1, a1, c1 = self._nms(kb, ks, ka, 4, kc, N, nlev, self.keep_scale, 1, 0.7, 0, self.cap_p)
            lib.fr_box_refine(_lib.ptr(b1), _lib.ptr(a1), 4, _lib.ptr(c1), N, self.cap_p, 0, self._s)
            self._mark("stage1_nms")
            if trace is not None:
                trace.update(stage1_boxes=b1, stage1_scores=s1, stage1_counts=c1)
            # ---- stage 2
            B2 = N * self.cap_p
            crops = None
            split = self.split_ro and self.fused_crop and trace is None and batch
            self._tls.path["split_ro"] = bool(split)
            if split:                                   # conv2 on the f16 matrix cores (split precision) + exact pass at the threshold
                y2, lc2 = self.crop_conv12_split(0, frames, b1, c1, self.cap_p)
                head2 = self.rnet(None, B2, c1, self.cap_p, x2=y2)
                self.exact_pass(0, frames, b1, c1, self.cap_p, head2, t1, lc2)
            elif self.fused_crop and trace is None:     # crop + conv1 + pool in one kernel: no crop tensor in HBM
                head2 = self.rnet(None, B2, c1, self.cap_p, x1=self.crop_conv1(0, frames, b1, c1, self.cap_p))
            else:
                crops = self._f32(B2, 24, 24, 4)
                lib.fr_crop_resize_norm(self._fptr(frames), N, H, W, _lib.ptr(b1), _lib.ptr(c1), self.cap_p, 24,
                                        _lib.ptr(crops), self._s)
                head2 = self.rnet(crops, B2, c1, self.cap_p)
            sb, ss, sa, sc = self._f32(N, self.cap_p, 4), self._f32(N, self.cap_p), self._f32(N, self.cap_p, 4), self._i32(N)
            prob2 = self._f32(N, self.cap_p) if trace is not None else None
            lib.fr_stage_select(_lib.ptr(b1), _lib.ptr(head2), 6, _lib.ptr(c1), N, self.cap_p, t1, _lib.ptr(sb),
                                _lib.ptr(ss), _lib.ptr(sa), 4, _lib.ptr(sc), _lib.ptr(prob2), self._s)
            b2, s2, a2, c2 = self._nms(sb, ss, sa, 4, sc, N, 1, self.cap_p, 0, 0.7, 0, self.cap_r)
            lib.fr_box_refine(_lib.ptr(b2), _lib.ptr(a2), 4, _lib.ptr(c2), N, self.cap_r, 1, self._s)
            self._mark("stage2")
            if trace is not None:
                trace.update(rnet_crops=crops, rnet_head=head2, rnet_prob=prob2, stage2_boxes=b2, stage2_scores=s2,
                             stage2_counts=c2)
            # ---- stage 3
            B3 = N * self.cap_r
            if split:
                y3, lc3 = self.crop_conv12_split(1, frames, b2, c2, self.cap_r)
                head3 = self.onet(None, B3, c2, self.cap_r, x2=y3)
                self.exact_pass(1, frames, b2, c2, self.cap_r, head3, t2, lc3)
            elif self.fused_crop and trace is None:
                head3 = self.onet(None, B3, c2, self.cap_r, x1=self.crop_conv1(1, frames, b2, c2, self.cap_r))
            else:
                crops3 = self._f32(B3, 48, 48, 4)
                lib.fr_crop_resize_norm(self._fptr(frames), N, H, W, _lib.ptr(b2), _lib.ptr(c2), self.cap_r, 48,
                                        _lib.ptr(crops3), self._s)
                head3 = self.onet(crops3, B3, c2, self.cap_r)
            tb, ts, ta, tc = self._f32(N, self.cap_r, 4), self._f32(N, self.cap_r), self._f32(N, self.cap_r, 14), self._i32(N)
            prob3 = self._f32(N, self.cap_r) if trace is not None else None
            lib.fr_stage_select(_lib.ptr(b2), _lib.ptr(head3), 16, _lib.ptr(c2), N, self.cap_r, t2, _lib.ptr(tb),
                                _lib.ptr(ts), _lib.ptr(ta), 14, _lib.ptr(tc), _lib.ptr(prob3), self._s)
            lib.fr_box_refine(_lib.ptr(tb), _lib.ptr(ta), 14, _lib.ptr(tc), N, self.cap_r, 2, self._s)
            self._tls.cache = None                      # what is returned to the caller is never a cached work tensor
            b3, s3, a3, c3 = self._nms(tb, ts, ta, 14, tc, N, 1, self.cap_r, 0, 0.7, 1, self.cap_o, out=_out, results=True)
            self._mark("stage3")
            if trace is not None:
                trace.update(onet_head=head3, onet_prob=prob3)
            if record:
                calls = lib.stop_recording()         # None: a launch the list cannot express happened (Lib.recording_invalid)
                if calls and not self._tls.cache_grew:
                    seq = self._make_sequence(calls, frames, (b3, s3, a3, c3), cfg)
                    if seq is not None:
                        seq["events"] = rec_events                   # the call list holds their handles
                        seqs[key] = seq
            if self._tls.cache_grew and getattr(self._tls, "seqs", None):
                # a work tensor was replaced in this call (recording or not): a list recorded earlier for this frame shape
                # holds the freed tensor's pointer
                self._tls.seqs.pop((N, H, W, main.cuda_stream), None)
            # aux = (reg4, (x1,y1)..(x5,y5)): kps is a strided view, no copy
            kps = a3[..., 4:14].unflatten(-1, (5, 2))
        return b3, s3, kps, c3

    def _make_sequence(self, calls, frames, outs, cfg):
        """The recorded C calls of one eager single-frame detect_batch as an fr_call array.  The slots a replay patches -
        the frame and the four result tensors - are those the call sites passed WITH that role (_lib.RolePtr: ``_fptr``,
        ``_nms(results=True)``), noted by the recorder beside the call; no slot is found by comparing pointer values.  The
        values only serve as a check: a slot that holds the frame's or a result's address without the role would be a
        call site that forgot it, and the list is refused: None is returned, nothing is cached for the frame shape, the call
        that was being recorded has already produced its (valid, eager) results and later calls stay eager."""
        log = logging.getLogger(__name__)
        arr = (_lib.Call * len(calls))()
        want = {"frame": frames.data_ptr(), **{"out%d" % i: t.data_ptr() for i, t in enumerate(outs)}}
        fpos, opos = [], [[] for _ in outs]
        for k, (fid, slots, roles) in enumerate(calls):
            arr[k].fn, arr[k].nargs = fid, len(slots)
            for i, v in enumerate(slots):
                arr[k].a[i] = v
            tagged = dict(roles)
            for i, role in roles:
                if slots[i] != want[role]:
                    log.warning("MTCNNHIP: recorded call %d: slot %d carries the role '%s' but holds another tensor; call list refused", k, i, role)
                    return None
                (fpos if role == "frame" else opos[int(role[3:])]).append((k, i))
            if fid < 8:                              # (event / stream handles of the pseudo calls are not tensors)
                for i, v in enumerate(slots):
                    if i not in tagged and v in want.values():
                        log.warning("MTCNNHIP: recorded call %d: slot %d holds a patched tensor without its role; call list refused", k, i)
                        return None
        if not fpos or not all(opos):
            log.warning("MTCNNHIP: the recorded call list does not mention the frame and every result tensor; refused")
            return None
        return {"arr": arr, "n": len(calls), "fpos": fpos, "opos": opos, "cfg": cfg,
                "meta": [(tuple(t.shape), t.dtype) for t in outs]}

    def _replay(self, seq, frames):
        arr = seq["arr"]
        outs = [torch.empty(shape, dtype=dtype, device=self.device) for shape, dtype in seq["meta"]]
        fptr = frames.data_ptr()
        for k, i in seq["fpos"]:
            arr[k].a[i] = fptr
        for t, pos in zip(outs, seq["opos"]):
            p = t.data_ptr()
            for k, i in pos:
                arr[k].a[i] = p
        self.lib.fr_detect_sequence(arr, seq["n"])
        b3, s3, a3, c3 = outs
        return b3, s3, a3[..., 4:14].unflatten(-1, (5, 2)), c3
