"""Device-resident gallery + matcher (rows a-6/a-7/a-8/a-10 of SURVEY.md section 8).

Replaces the per-face Python loop over ``dict[id -> float32[512]]`` at
/root/reference/infrenceServer.py:535-552 (and peopleCount.py:866-887) with one HIP scan
of a row-indexed ``[N,512]`` matrix.  Row order = insertion order of the reference's
dict (employees in cursor order, then visitors: infrenceServer.py:264,288); the winner is
the maximum score, lowest row on exact ties (strict ``>`` in the reference loop).
"""
import numpy as np
import torch

from . import _lib

DIM = 512


class GalleryMatcher:
    """``G[N,512]`` float32 unit rows on the device + the id table."""

    def __init__(self, device="cuda:0", f16_scan=False):
        """``f16_scan``: keep an f16 copy of the rows and scan it on the f16 matrix cores, then re-score the
        top-4 rows per query exactly in f32 (large galleries / many queries); the f32 rows are kept."""
        _lib.require_gpu()
        self.f16_scan = f16_scan
        self.G16 = None
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.ids = []
        self.G = torch.empty((0, DIM), dtype=torch.float32, device=self.device)
        self._ws = None

    def __len__(self):
        return self.G.shape[0]

    def set_rows(self, ids, rows, normalise=True):
        """rows: float32 [N,512] (host or device).  ``normalise`` applies v/||v|| on the
        device exactly as the gallery ingest does (infrenceServer.py:271,324)."""
        rows = torch.as_tensor(np.asarray(rows, np.float32) if not torch.is_tensor(rows) else rows)
        rows = rows.to(self.device, torch.float32).contiguous().reshape(-1, DIM)
        if len(ids) != rows.shape[0]:
            raise ValueError("ids and rows disagree")
        if normalise and rows.shape[0]:
            out = torch.empty_like(rows)
            with torch.cuda.device(self.device):
                self.lib.fr_l2norm_rows_f32(_lib.ptr(rows), _lib.ptr(out), rows.shape[0], DIM, _lib.stream_ptr())
            rows = out
        self.ids = list(ids)
        self.G = rows
        self.G16 = None
        if self.f16_scan and rows.shape[0]:
            self.G16 = torch.empty(rows.shape, dtype=torch.float16, device=self.device)
            with torch.cuda.device(self.device):
                self.lib.fr_f32_to_f16(_lib.ptr(rows), _lib.ptr(self.G16), rows.numel(), _lib.stream_ptr())

    def _workspace(self, F):
        need = self.lib.fr_gallery_match_workspace(F, self.G.shape[0])
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def match_device(self, Q, renormalise=True, row_offset=0):
        """Q: float32 [F,512] device tensor of ``normed_embedding`` rows.
        Returns device tensors (idx int64[F] (-1: empty gallery), score float32[F])."""
        Q = Q.to(self.device, torch.float32).contiguous().reshape(-1, DIM)
        F = Q.shape[0]
        idx = torch.empty(F, dtype=torch.int64, device=self.device)
        score = torch.empty(F, dtype=torch.float32, device=self.device)
        if F == 0:
            return idx, score
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr()
            if renormalise:                      # infrenceServer.py:532
                Qn = torch.empty_like(Q)
                self.lib.fr_l2norm_rows_f32(_lib.ptr(Q), _lib.ptr(Qn), F, DIM, s)
                Q = Qn
            if self.G16 is not None:
                need = self.lib.fr_gallery_match_f16_workspace(F, self.G.shape[0])
                if self._ws is None or self._ws.numel() < need:
                    self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
                self.lib.fr_gallery_match_f16(_lib.ptr(Q), _lib.ptr(self.G16), _lib.ptr(self.G), F, self.G.shape[0], DIM,
                                              row_offset, _lib.ptr(idx), _lib.ptr(score), _lib.ptr(self._ws),
                                              self._ws.numel(), s)
                return idx, score
            ws = self._workspace(F)
            self.lib.fr_gallery_match_f32(_lib.ptr(Q), _lib.ptr(self.G), F, self.G.shape[0], DIM, row_offset,
                                          _lib.ptr(idx), _lib.ptr(score), _lib.ptr(ws), ws.numel(), s)
        return idx, score

    def decide_device(self, idx, score, thr, unknown_thr=None):
        """1 recognised / 0 unknown / 2 dropped (peopleCount.py:876-887 band)."""
        d = torch.empty(idx.shape[0], dtype=torch.int32, device=self.device)
        if idx.shape[0]:
            with torch.cuda.device(self.device):
                self.lib.fr_match_decide(_lib.ptr(idx), _lib.ptr(score), idx.shape[0], float(thr),
                                         float(thr if unknown_thr is None else unknown_thr), _lib.ptr(d),
                                         _lib.stream_ptr())
        return d

    def match(self, Q, thr=0.4, unknown_thr=None):
        """Host-facing: returns (ids list (None = unknown), scores float32[F], idx int64[F])."""
        Q = torch.as_tensor(np.asarray(Q, np.float32)) if not torch.is_tensor(Q) else Q
        idx, score = self.match_device(Q)
        dec = self.decide_device(idx, score, thr, unknown_thr)
        idx_h, score_h, dec_h = idx.cpu().numpy(), score.cpu().numpy(), dec.cpu().numpy()
        ids = [self.ids[i] if (d == 1 and i >= 0) else None for i, d in zip(idx_h, dec_h)]
        return ids, score_h, idx_h
