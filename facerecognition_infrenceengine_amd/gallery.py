"""Device-resident gallery + matcher (rows a-6/a-7/a-8/a-10 of SURVEY.md section 8).

Replaces the per-face Python loop over ``dict[id -> float32[512]]`` at
/root/reference/infrenceServer.py:535-552 (and peopleCount.py:866-887) with one HIP scan
of a row-indexed ``[N,512]`` matrix.  Row order = insertion order of the reference's
dict (employees in cursor order, then visitors: infrenceServer.py:264,288); the winner is
the maximum score, lowest row on exact ties (strict ``>`` in the reference loop).
"""
import threading

import numpy as np
import torch

from . import _lib

DIM = 512
_PIN = threading.local()       # pinned host buffers of GalleryMatcher.match, per thread and query count


class StaleViewError(_lib.FrError):
    """A GalleryView was used after its gallery's membership changed: fetch a fresh view and retry."""


class GalleryMatcher:
    """``G[N,512]`` float32 unit rows on the device + the id table."""

    def __init__(self, device="cuda:0", f16_scan=False, scan=None):
        """``scan``: "f32" (default: exact f32 scan on the f32 matrix cores), "f16" or "f8": keep a 16-bit / 8-bit
        copy of the rows, scan THAT in one pass on the f16 / fp8 matrix cores for the whole query batch, then
        re-score the top-4 / top-8 rows per query exactly in f32 (large galleries / many queries: BASELINE
        configs C4 / C5); the f32 rows are kept.  ``f16_scan=True`` is the older spelling of scan="f16"."""
        _lib.require_gpu()
        self.scan = scan or ("f16" if f16_scan else "f32")
        if self.scan not in ("f32", "f16", "f8"):
            raise ValueError("scan must be 'f32', 'f16' or 'f8'")
        self.f16_scan = self.scan == "f16"
        self.G16 = None                       # the coarse copy (f16 or fp8 e4m3 x 256)
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.ids = []
        self.G = torch.empty((0, DIM), dtype=torch.float32, device=self.device)

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
        if self.scan != "f32" and rows.shape[0]:
            with torch.cuda.device(self.device):
                if self.scan == "f16":
                    self.G16 = torch.empty(rows.shape, dtype=torch.float16, device=self.device)
                    self.lib.fr_f32_to_f16(_lib.ptr(rows), _lib.ptr(self.G16), rows.numel(), _lib.stream_ptr())
                else:
                    self.G16 = torch.empty(rows.shape, dtype=torch.uint8, device=self.device)
                    self.lib.fr_f32_to_f8(_lib.ptr(rows), _lib.ptr(self.G16), rows.numel(), _lib.stream_ptr())

    def _workspace(self, need):
        """Scratch for ONE scan, taken from torch's caching allocator under the current stream: the block is
        owned by that stream, so concurrent matches on different streams never share (or free) each other's
        partial results.  (A per-object buffer did: two pipes driving one matcher overwrote ws_score/ws_idx.)"""
        return torch.empty(max(int(need), 16), dtype=torch.uint8, device=self.device)

    def match_device(self, Q, renormalise=True, row_offset=0, counts=None, seg_len=0):
        """Q: float32 [F,512] device tensor of ``normed_embedding`` rows.
        Returns device tensors (idx int64[F] (-1: empty gallery), score float32[F]).
        ``counts`` (device int32 [F / seg_len]) marks padding slots of a gathered batch (sharded match): slot f
        is real iff f % seg_len < counts[f // seg_len]; padding costs no scan work and reports (-1, -1)."""
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
            cp = _lib.ptr(counts) if counts is not None else None
            if counts is not None:
                assert counts.dtype == torch.int32 and counts.is_contiguous() and seg_len > 0 and F == counts.numel() * seg_len
            if self.G16 is not None:
                wsz, fn = ((self.lib.fr_gallery_match_f16_workspace, self.lib.fr_gallery_match_f16) if self.scan == "f16"
                           else (self.lib.fr_gallery_match_f8_workspace, self.lib.fr_gallery_match_f8))
                ws = self._workspace(wsz(F, self.G.shape[0]))
                fn(_lib.ptr(Q), _lib.ptr(self.G16), _lib.ptr(self.G), F, self.G.shape[0], DIM, row_offset,
                   _lib.ptr(idx), _lib.ptr(score), _lib.ptr(ws), ws.numel(), cp, seg_len, s)
                return idx, score
            ws = self._workspace(self.lib.fr_gallery_match_workspace(F, self.G.shape[0]))
            self.lib.fr_gallery_match_f32(_lib.ptr(Q), _lib.ptr(self.G), F, self.G.shape[0], DIM, row_offset,
                                          _lib.ptr(idx), _lib.ptr(score), _lib.ptr(ws), ws.numel(), cp, seg_len, s)
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
        # three results, ONE synchronisation: asynchronous copies into pinned host memory, then a stream sync
        F = idx.shape[0]
        pin = _PIN.__dict__.setdefault("bufs", {})          # per thread: match() may be called by several on one matcher
        h = pin.get(F)
        if h is None:
            if len(pin) >= 16:
                pin.clear()
            h = pin[F] = (torch.empty(F, dtype=torch.int64).pin_memory(), torch.empty(F, dtype=torch.float32).pin_memory(),
                          torch.empty(F, dtype=torch.int32).pin_memory())
        with torch.cuda.device(self.device):
            for dst, src in zip(h, (idx, score, dec)):
                dst.copy_(src, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        idx_h, score_h, dec_h = h[0].numpy().copy(), h[1].numpy().copy(), h[2].numpy().copy()
        ids = [self.ids[i] if (d == 1 and i >= 0) else None for i, d in zip(idx_h, dec_h)]
        return ids, score_h, idx_h


class DeviceGallery:
    """One device-resident slab of gallery row slots with in-place updates + per-company views
    (SURVEY.md section 8f row 2).

    The reference keeps ``dict[id -> float32[512]]`` (infrenceServer.py:46), inserts/overwrites entries on
    every incremental sync (:271-273, :324-326), deletes inactive people (:250-252) and, PER FRAME, filters
    the dict by the company's member ids (:343-380).  Here the rows live once in HBM: ``upsert`` writes
    changed rows into their slots in place (new ids take a free slot, capacity doubles when full),
    ``remove`` frees slots, and a company view is only an int64 slot list in the reference's dict order -
    ``GalleryView.match_device`` scans the slab through it (no per-company copy of the rows).
    """

    def __init__(self, device="cuda:0", capacity=1024):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.G = torch.zeros((max(int(capacity), 1), DIM), dtype=torch.float32, device=self.device)
        self.slot_of = {}                # id -> slot
        self._free = []                  # freed slots, reused LIFO
        self._next = 0                   # first never-used slot
        self.generation = 0              # bumps on every membership change (views check it)

    def __len__(self):
        return len(self.slot_of)

    @property
    def capacity(self):
        return self.G.shape[0]

    def _grow(self, need):
        cap = self.capacity
        while cap < need:
            cap *= 2
        if cap != self.capacity:
            G = torch.zeros((cap, DIM), dtype=torch.float32, device=self.device)
            G[: self.capacity].copy_(self.G)
            self.G = G

    def upsert(self, ids, rows, normalise=False):
        """rows: float32 [n,512] (host or device).  Existing ids are overwritten in place (same slot), new
        ids take a slot.  ``normalise`` applies v/||v|| on the device (the manager normalises on the host with
        NumPy, bit-exactly as infrenceServer.py:271, and passes False)."""
        ids = list(ids)
        rows = torch.as_tensor(np.asarray(rows, np.float32) if not torch.is_tensor(rows) else rows)
        rows = rows.to(self.device, torch.float32).contiguous().reshape(-1, DIM)
        if len(ids) != rows.shape[0]:
            raise ValueError("ids and rows disagree")
        if not ids:
            return
        last = {i: k for k, i in enumerate(ids)}          # an id given twice: the last row wins, as dict assignment
        if len(last) != len(ids):
            keep = sorted(last.values())
            ids = [ids[k] for k in keep]
            rows = rows[torch.tensor(keep, device=self.device)]
        new = [i for i in ids if i not in self.slot_of]
        self._grow(self._next + max(len(new) - len(self._free), 0))
        for i in new:
            if self._free:
                self.slot_of[i] = self._free.pop()
            else:
                self.slot_of[i] = self._next
                self._next += 1
        if new:
            self.generation += 1
        slots = torch.tensor([self.slot_of[i] for i in ids], dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_gallery_update_rows_f32(_lib.ptr(self.G), _lib.ptr(slots), _lib.ptr(rows), len(ids), DIM,
                                                1 if normalise else 0, _lib.stream_ptr())

    def remove(self, ids):
        n = 0
        for i in ids:
            s = self.slot_of.pop(i, None)
            if s is not None:
                self._free.append(s)
                n += 1
        if n:
            self.generation += 1
        return n

    def view(self, ids):
        """View over the given ids, in the given order (ids not in the gallery are skipped)."""
        return GalleryView(self, [i for i in ids if i in self.slot_of])


class GalleryView:
    """Ordered subset of a ``DeviceGallery``: the rows one company's frames are matched against."""

    def __init__(self, gallery, ids):
        self.gallery, self.ids = gallery, list(ids)
        self.generation = gallery.generation
        self.lib, self.device = gallery.lib, gallery.device
        self.slots = torch.tensor([gallery.slot_of[i] for i in self.ids], dtype=torch.int64, device=self.device)

    def __len__(self):
        return len(self.ids)

    def rows(self):
        """The view's rows as a [N,512] device tensor (a copy; for tests / export)."""
        return self.gallery.G[self.slots]

    def match_device(self, Q, renormalise=True):
        """As ``GalleryMatcher.match_device``; idx is the position in ``self.ids`` (-1: empty view)."""
        if self.generation != self.gallery.generation:
            raise StaleViewError("GalleryView is stale: the gallery's membership changed after the view was made")
        Q = Q.to(self.device, torch.float32).contiguous().reshape(-1, DIM)
        F = Q.shape[0]
        idx = torch.empty(F, dtype=torch.int64, device=self.device)
        score = torch.empty(F, dtype=torch.float32, device=self.device)
        if F == 0:
            return idx, score
        with torch.cuda.device(self.device):
            s = _lib.stream_ptr()
            if renormalise:
                Qn = torch.empty_like(Q)
                self.lib.fr_l2norm_rows_f32(_lib.ptr(Q), _lib.ptr(Qn), F, DIM, s)
                Q = Qn
            ws = GalleryMatcher._workspace(self, self.lib.fr_gallery_match_workspace(F, len(self.ids)))
            self.lib.fr_gallery_match_view_f32(_lib.ptr(Q), _lib.ptr(self.gallery.G), _lib.ptr(self.slots), F,
                                               len(self.ids), DIM, _lib.ptr(idx), _lib.ptr(score),
                                               _lib.ptr(ws), ws.numel(), s)
        return idx, score

    decide_device = GalleryMatcher.decide_device
    match = GalleryMatcher.match
