"""GPU parity for the 'next' rows: enrolment arithmetic and unknown-person clustering vs vectors produced by
the reference's own methods (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _FakeApp:
    device = torch.device("cuda:0")


def test_pose_consistency_and_duplicate_vs_reference(golden):
    from facerecognition_infrenceengine_amd.enrol import Enroller
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    d = golden("enrol_kat.npz")
    en = Enroller(_FakeApp())
    for k in ("same", "one_off", "single", "edge_lo", "edge_hi"):
        ok, pair = en.check_image_similarity(list(d[f"sim_{k}_in"]))
        assert int(ok) == d[f"sim_{k}_ok"][0]
        assert tuple(pair if pair else (-1, -1)) == tuple(d[f"sim_{k}_pair"])
    m = GalleryMatcher("cuda:0")
    m.set_rows(list(range(len(d["stored"]))), d["stored"], normalise=True)      # stored rows are un-normalised means
    for k in ("dup", "nodup"):
        is_dup, idx = en.check_duplicate(d[f"{k}_new"], m)
        assert int(is_dup) == d[f"{k}_is"][0] and (idx if idx is not None else -1) == d[f"{k}_idx"][0]


def test_mean_and_row_blob_vs_reference(golden):
    import pickle
    from facerecognition_infrenceengine_amd.enrol import Enroller
    d = golden("gallery_row_kat.npz")
    avg = Enroller(_FakeApp()).mean_embedding(list(d["poses"]))
    assert avg.dtype == np.float32
    np.testing.assert_allclose(avg, d["avg"], rtol=0, atol=1e-7)                # np.mean pairwise vs sequential sum
    assert len(pickle.dumps(avg)) == len(d["blob"]) == 2200


def test_first_above_lowest_row_and_inclusive():
    from facerecognition_infrenceengine_amd import _lib
    from facerecognition_infrenceengine_amd.enrol import first_above
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    rng = np.random.default_rng(0)
    G = rng.standard_normal((5000, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    q = G[4000].copy()
    G[123] = q; G[4500] = q
    m = GalleryMatcher("cuda:0"); m.set_rows(list(range(5000)), G, normalise=False)
    idx, score = first_above(_lib.load(), m, q, 0.9, inclusive=False)
    assert idx == 123 and abs(score - 1) < 1e-5
    assert first_above(_lib.load(), m, -q, 0.9, inclusive=False)[0] == -1
    m0 = GalleryMatcher("cuda:0")
    assert first_above(_lib.load(), m0, q, 0.1, inclusive=True)[0] == -1          # empty gallery


def test_unknown_clustering_vs_reference(golden):
    from facerecognition_infrenceengine_amd.enrol import UnknownClusters
    d = golden("unknown_kat.npz")
    uc = UnknownClusters("cuda:0")
    assign = [uc.assign(e) for e in d["seq"]]
    assert assign == list(d["assign"])
    assert uc.counts == list(d["counts"])
    np.testing.assert_allclose(uc.avg[:len(uc.hist)].cpu().numpy(), d["final_avg"], atol=1e-7)


def test_enrol_end_to_end():
    import os, sys, warnings
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import FaceAnalysis, GalleryMatcher
    from facerecognition_infrenceengine_amd.enrol import Enroller, largest_face_index
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        app = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)
    frame = synth_frame(240, 320, 4)
    faces = app.get(frame)
    en = Enroller(app)
    e = en.process_image(frame)
    assert np.array_equal(e, faces[largest_face_index(faces)].normed_embedding)
    rng = np.random.default_rng(1)
    G = rng.standard_normal((50, 512)).astype(np.float32)
    m = GalleryMatcher("cuda:0"); m.set_rows([f"p{i}" for i in range(50)], G)
    r = en.enrol([frame, frame], m)
    assert r["status"] == "done" and len(r["blob"]) == 2200
    G[7] = e
    m.set_rows([f"p{i}" for i in range(50)], G)
    r = en.enrol([frame], m)
    assert r["status"] == "duplicate" and r["duplicate_id"] == "p7"
    assert en.enrol([np.zeros((8, 8, 3), np.uint8)], m)["status"] == "no_face"
