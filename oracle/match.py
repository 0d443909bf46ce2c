"""Oracle: gallery match arithmetic (numpy float32), literal restatement.

Follows /root/reference/infrenceServer.py:530-552 (live path),
/root/reference/peopleCount.py:860-887 (counting path) and the gallery-row
ingest at /root/reference/infrenceServer.py:269-273,324 /
/root/reference/trainingServer.py:355,393.

Test infrastructure only (see oracle/__init__.py).
"""
import pickle

import numpy as np


def renormalise(normed_embedding):
    """infrenceServer.py:532 -- second L2 normalise of an already-normalised row."""
    return normed_embedding / np.linalg.norm(normed_embedding)


def linear_scan(face_embedding, embeddings):
    """infrenceServer.py:535-542 -- strict '>' first-maximum over an ORDERED mapping.

    ``embeddings`` is an ordered mapping id -> float32[512] (dict keeps insertion
    order).  Returns (best_match_id, best_score) exactly as the reference's loop
    leaves them (best_score stays the int -1 if nothing beats it).
    """
    best_match_id = None
    best_score = -1
    for person_id, registered_embedding in embeddings.items():
        similarity = np.dot(face_embedding, registered_embedding)
        if similarity > best_score:
            best_score = similarity
            best_match_id = person_id
    return best_match_id, best_score


def decide_live(best_match_id, best_score, recognition_threshold=0.4):
    """infrenceServer.py:545-552 -- (matched id | None, recognition_score)."""
    if best_match_id and best_score >= recognition_threshold:
        return best_match_id, best_score
    return None, 0


def decide_counting(best_match_id, best_score, recognition_threshold=0.45,
                    unknown_threshold=0.35):
    """peopleCount.py:876-887 -- 'recognized' | 'unknown' | 'dropped'."""
    if best_match_id and best_score >= recognition_threshold:
        return "recognized"
    elif best_score < unknown_threshold:
        return "unknown"
    return "dropped"


def match_rows(Q, G):
    """Batched form of renormalise+linear_scan for row-indexed galleries.

    Q: float32[F,D] (normed embeddings), G: float32[N,D] unit rows, ids = row
    index.  Row-by-row np.dot (BLAS sdot, like the reference), first maximum.
    Returns (idx int64[F] (-1 if N == 0), score float32[F]).
    """
    Q = np.asarray(Q, np.float32)
    G = np.asarray(G, np.float32)
    idx = np.full(Q.shape[0], -1, np.int64)
    score = np.full(Q.shape[0], -1, np.float32)
    for f in range(Q.shape[0]):
        q = renormalise(Q[f])
        best, best_i = -1, -1
        for n in range(G.shape[0]):
            s = np.dot(q, G[n])
            if s > best:
                best, best_i = s, n
        idx[f], score[f] = best_i, best
    return idx, score


def match_rows_fast(Q, G):
    """Vectorised equivalent of match_rows for large N (matrix product instead of
    per-row sdot: scores agree to ~1e-6, argmax ties resolved to lowest row)."""
    Q = np.asarray(Q, np.float32)
    G = np.asarray(G, np.float32)
    Qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    S = Qn @ G.T
    idx = np.argmax(S, axis=1)  # first maximum
    return idx.astype(np.int64), S[np.arange(len(idx)), idx].astype(np.float32)


def gallery_row_blob(pose_embeddings):
    """trainingServer.py:355,393 -- mean of the pose rows, pickled."""
    avg_embedding = np.mean(pose_embeddings, axis=0)
    return pickle.dumps(avg_embedding)


def gallery_row_load(blob):
    """infrenceServer.py:270-271 -- unpickle and divide by the L2 norm."""
    embedding = pickle.loads(blob)
    return embedding / np.linalg.norm(embedding)
