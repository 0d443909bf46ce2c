"""Oracle: enrolment arithmetic + unknown-person clustering (numpy), restated.

Follows /root/reference/trainingServer.py:188-192 (duplicate cosine),
:202-214 (pose consistency), :234-243 (largest face) and
/root/reference/peopleCount.py:52-91 (UnknownPerson running mean),
:432-500 (first-hit cluster assignment at 0.65).

Test infrastructure only (see oracle/__init__.py).
"""
from collections import deque

import numpy as np


def cosine(a, b):
    """trainingServer.py:188-190 / :207-209."""
    return np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b))


def check_image_similarity(embeddings, similarity_threshold=0.4):
    """trainingServer.py:202-214 -- first (i, j) pair whose cosine < threshold."""
    if len(embeddings) < 2:
        return True, None
    for i in range(len(embeddings)):
        for j in range(i + 1, len(embeddings)):
            if cosine(embeddings[i], embeddings[j]) < similarity_threshold:
                return False, (i, j)
    return True, None


def check_duplicate(new_embedding, stored_rows, duplicate_threshold=0.4):
    """trainingServer.py:181-194 -- index of the first stored row with cosine > thr."""
    for k, existing in enumerate(stored_rows):
        if existing is not None and cosine(new_embedding, existing) > duplicate_threshold:
            return True, k
    return False, None


def largest_face_index(bboxes):
    """trainingServer.py:234-239 -- first index of the max (x2-x1)*(y2-y1)."""
    areas = [(b[2] - b[0]) * (b[3] - b[1]) for b in bboxes]
    return areas.index(max(areas))


class UnknownCluster:
    """peopleCount.py:52-91 -- 10-deep running mean, NOT renormalised."""

    def __init__(self, first_embedding):
        self.embeddings = deque(maxlen=10)
        self.embeddings.append(first_embedding)
        self.avg_embedding = first_embedding
        self.detection_count = 1

    def update(self, embedding):
        self.detection_count += 1
        self.embeddings.append(embedding)
        self.avg_embedding = np.mean(list(self.embeddings), axis=0)

    def compute_similarity(self, embedding):
        return np.dot(self.avg_embedding, embedding)


def assign_unknown(clusters, embedding, threshold=0.65):
    """peopleCount.py:441-449 -- first cluster (insertion order) with sim >= thr;
    otherwise a new cluster is appended.  Returns the cluster index."""
    for k, c in enumerate(clusters):
        if c.compute_similarity(embedding) >= threshold:
            c.update(embedding)
            return k
    clusters.append(UnknownCluster(embedding))
    return len(clusters) - 1
