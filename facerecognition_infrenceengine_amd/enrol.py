"""Enrolment arithmetic and unknown-person clustering on the engine.

Mirrors /root/reference/trainingServer.py:170-247,328,355-358,393 (largest face -> pose consistency ->
mean -> duplicate check -> pickled float32[512] row) and /root/reference/peopleCount.py:52-91,432-449
(10-deep running mean, first cluster with dot >= 0.65).  The O(N) per-row GridFS read + cosine loop of
the reference's duplicate check becomes one `fr_gallery_first_above_f32` scan of the device gallery.
"""
import pickle
from collections import deque

import numpy as np
import torch

from . import _lib

DIM = 512


def largest_face_index(faces):
    """trainingServer.py:234-239: first index of the max bbox area."""
    areas = [(f.bbox[2] - f.bbox[0]) * (f.bbox[3] - f.bbox[1]) for f in faces]
    return areas.index(max(areas))


class Enroller:
    def __init__(self, face_analysis, similarity_threshold=0.4, duplicate_threshold=0.4):
        self.app, self.similarity_threshold, self.duplicate_threshold = face_analysis, similarity_threshold, duplicate_threshold
        self.lib = _lib.load()
        self.device = face_analysis.device

    def process_image(self, image):
        """trainingServer.py:216-247: normed embedding of the largest face, or None.  ``image``: a BGR uint8 array, or
        the encoded bytes the reference reads from GridFS (``:219-221``: decoded here, None when they do not decode)."""
        if isinstance(image, (bytes, bytearray, memoryview)):
            from .ingest import decode_image
            image = decode_image(image)
            if image is None:
                return None
        faces = self.app.get(image)
        if not faces:
            return None
        return faces[largest_face_index(faces) if len(faces) > 1 else 0].normed_embedding

    def check_image_similarity(self, embeddings):
        """trainingServer.py:202-214: first (i, j), i < j, with cosine < threshold."""
        k = len(embeddings)
        if k < 2:
            return True, None
        x = torch.from_numpy(np.asarray(embeddings, np.float32)).to(self.device).contiguous()
        out = torch.empty((k, k), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_cosine_matrix_f32(_lib.ptr(x), _lib.ptr(x), k, k, DIM, _lib.ptr(out), _lib.stream_ptr())
        s = out.cpu().numpy()
        for i in range(k):
            for j in range(i + 1, k):
                if s[i, j] < self.similarity_threshold:
                    return False, (i, j)
        return True, None

    def mean_embedding(self, embeddings):
        """trainingServer.py:355: np.mean(face_embeddings, axis=0) (float32, NOT renormalised)."""
        x = torch.from_numpy(np.asarray(embeddings, np.float32)).to(self.device).contiguous()
        out = torch.empty(DIM, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_mean_rows_f32(_lib.ptr(x), x.shape[0], DIM, _lib.ptr(out), _lib.stream_ptr())
        return out.cpu().numpy()

    def check_duplicate(self, new_embedding, matcher):
        """trainingServer.py:170-200 against a GalleryMatcher of unit rows: (is_dup, id of the first row with
        cosine > threshold)."""
        idx, score = first_above(self.lib, matcher, new_embedding, self.duplicate_threshold, inclusive=False)
        return (idx >= 0), (matcher.ids[idx] if idx >= 0 else None)

    def enrol(self, pose_images, matcher):
        """Whole job arithmetic (trainingServer.py:312-398).  Returns dict(status=..., ...)."""
        embs = [e for e in (self.process_image(im) for im in pose_images) if e is not None]
        if not embs:
            return {"status": "no_face"}
        ok, pair = self.check_image_similarity(embs)
        if not ok:
            return {"status": "different_people", "pair": pair}
        avg = self.mean_embedding(embs)
        dup, dup_id = self.check_duplicate(avg, matcher)
        if dup:
            return {"status": "duplicate", "duplicate_id": dup_id, "embedding": avg}
        return {"status": "done", "embedding": avg, "blob": pickle.dumps(avg)}       # :393 gallery row format


def first_above(lib, matcher, embedding, thr, inclusive):
    """Lowest gallery row whose dot with the L2-normalised query passes the threshold."""
    q = torch.from_numpy(np.asarray(embedding, np.float32).reshape(1, DIM)).to(matcher.device)
    qn = torch.empty_like(q)
    idx = torch.empty(1, dtype=torch.int64, device=matcher.device)
    score = torch.empty(1, dtype=torch.float32, device=matcher.device)
    ws = torch.empty(8, dtype=torch.uint8, device=matcher.device)
    with torch.cuda.device(matcher.device):
        s = _lib.stream_ptr()
        lib.fr_l2norm_rows_f32(_lib.ptr(q), _lib.ptr(qn), 1, DIM, s)
        lib.fr_gallery_first_above_f32(_lib.ptr(qn), _lib.ptr(matcher.G), 1, matcher.G.shape[0], DIM, float(thr),
                                       1 if inclusive else 0, 0, _lib.ptr(idx), _lib.ptr(score), _lib.ptr(ws), 8, s)
    return int(idx.item()), float(score.item())


class UnknownClusters:
    """peopleCount.py:52-91 + :432-449 on the device: cluster means live in one [C,512] matrix (NOT unit rows,
    exactly as the reference keeps them); assignment = first cluster with dot(avg, e) >= threshold."""

    def __init__(self, device="cuda:0", threshold=0.65, depth=10, capacity=1024):
        _lib.require_gpu()
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.threshold, self.depth = threshold, depth
        self.avg = torch.zeros((capacity, DIM), dtype=torch.float32, device=self.device)
        self.hist = []                  # per cluster: deque of device rows (maxlen = depth)
        self.counts = []

    def assign(self, embedding):
        e = torch.from_numpy(np.asarray(embedding, np.float32).reshape(1, DIM)).to(self.device)
        n = len(self.hist)
        hit = -1
        if n:
            idx = torch.empty(1, dtype=torch.int64, device=self.device)
            score = torch.empty(1, dtype=torch.float32, device=self.device)
            ws = torch.empty(8, dtype=torch.uint8, device=self.device)
            with torch.cuda.device(self.device):
                self.lib.fr_gallery_first_above_f32(_lib.ptr(e), _lib.ptr(self.avg), 1, n, DIM, float(self.threshold), 1,
                                                    0, _lib.ptr(idx), _lib.ptr(score), _lib.ptr(ws), 8, _lib.stream_ptr())
            hit = int(idx.item())
        if hit < 0:
            if n == self.avg.shape[0]:
                raise RuntimeError("UnknownClusters capacity exceeded")
            self.hist.append(deque([e], maxlen=self.depth))
            self.counts.append(1)
            self.avg[n] = e[0]              # first embedding is the mean as is (peopleCount.py:66)
            return n
        self.hist[hit].append(e)
        self.counts[hit] += 1
        rows = torch.cat(list(self.hist[hit])).contiguous()
        with torch.cuda.device(self.device):
            self.lib.fr_mean_rows_f32(_lib.ptr(rows), rows.shape[0], DIM, _lib.ptr(self.avg[hit]), _lib.stream_ptr())
        return hit
