"""device-resident gallery slab with in-place row updates and per-company views,
against the reference's dict semantics (infrenceServer.py:260-380) replayed on an ordered dict + the literal
scan loop of the oracle."""
from datetime import datetime, timedelta

import numpy as np
import pytest
import torch

from oracle import match as omatch

pytestmark = pytest.mark.gpu


def unit(rng, n):
    v = rng.standard_normal((n, 512)).astype(np.float32)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def check(view, mirror, Q):
    """ids identical to the literal loop over the mirror dict; scores to f32 summation-order tolerance."""
    assert view.ids == list(mirror)
    idx, score = view.match_device(torch.from_numpy(Q).cuda(), renormalise=True)
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    for f in range(Q.shape[0]):
        bid, bs = omatch.linear_scan(omatch.renormalise(Q[f]), mirror)
        if bid is None:
            assert idx[f] == -1 and score[f] == -1.0
        else:
            assert view.ids[idx[f]] == bid, (f, view.ids[idx[f]], bid)
            assert abs(score[f] - bs) < 2e-6


def test_slab_upsert_remove_views_match_dict_semantics():
    from facerecognition_infrenceengine_amd.gallery import DeviceGallery
    from facerecognition_infrenceengine_amd._lib import FrError
    rng = np.random.default_rng(7)
    g = DeviceGallery("cuda:0", capacity=4)               # forces several capacity doublings
    mirror = {}
    Q = unit(rng, 9)

    def upsert(ids, rows):
        g.upsert(ids, rows)
        for i, r in zip(ids, rows):
            mirror[i] = r                                  # dict assignment: existing key keeps its position

    rows = unit(rng, 300)
    upsert([f"p{i}" for i in range(300)], rows)
    # plant near-copies of the queries so winners are meaningful, plus an exact duplicate pair (tie -> first in order)
    upsert(["p17", "p250"], np.stack([Q[0], Q[1]]))
    upsert(["p40", "p41"], np.stack([Q[2], Q[2]]))
    assert g.capacity >= 300 and len(g) == 300
    v = g.view(list(mirror))
    check(v, mirror, Q)
    # overwrite in place: the SAME view object sees the new rows (no rebuild, no generation change)
    upsert(["p17"], unit(rng, 1))
    upsert(["p5"], Q[0:1])
    check(v, mirror, Q)
    # removal frees slots; stale views are refused; re-inserted ids go to the END of the order
    for i in ("p40", "p5", "p299"):
        del mirror[i]
    assert g.remove(["p40", "p5", "p299", "nobody"]) == 3
    with pytest.raises(FrError):
        v.match_device(torch.from_numpy(Q).cuda())
    check(g.view(list(mirror)), mirror, Q)                 # tie winner is now p41
    nxt = g._next
    upsert(["p40", "new1"], np.stack([Q[2], Q[3]]))        # p40 again: after p41 in the order now
    assert g._next == nxt and len(g._free) == 1            # freed slots were reused
    check(g.view(list(mirror)), mirror, Q)
    # per-company style views: arbitrary ordered subsets, an id twice in one upsert (last wins), empty view
    sub = [i for k, i in enumerate(mirror) if k % 3 == 1][::-1]
    check(g.view(sub), {i: mirror[i] for i in sub}, Q)
    upsert(["dup", "dup"], unit(rng, 2))
    check(g.view(["dup", "p41", "ghost"]), {"dup": mirror["dup"], "p41": mirror["p41"]}, Q)
    check(g.view([]), {}, Q)


def test_update_rows_normalise_matches_ingest():
    """fr_gallery_update_rows_f32(normalise=1) == the row the ingest stores (v / ||v||, float32)."""
    from facerecognition_infrenceengine_amd.gallery import DeviceGallery
    rng = np.random.default_rng(8)
    raw = (rng.standard_normal((64, 512)) * rng.uniform(0.1, 30, (64, 1))).astype(np.float32)
    g = DeviceGallery("cuda:0", capacity=8)
    g.upsert(list(range(64)), raw, normalise=True)
    got = g.view(list(range(64))).rows().cpu().numpy()
    want = np.stack([r / np.linalg.norm(r) for r in raw])
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-6


def test_embedding_manager_incremental_sync_updates_rows_in_place():
    from facerecognition_infrenceengine_amd.processor import EmbeddingManager, InMemoryStore
    rng = np.random.default_rng(9)
    store = InMemoryStore()
    raw = {}
    for i in range(40):
        raw[f"e{i}"] = rng.standard_normal(512).astype(np.float32) * 3
        store.add_employee(f"e{i}", "acme" if i % 2 else "globex", raw[f"e{i}"], name=f"E{i}")
    for i in range(10):
        raw[f"v{i}"] = rng.standard_normal(512).astype(np.float32)
        store.add_visitor(f"v{i}", "acme", raw[f"v{i}"], name=f"V{i}")
    mgr = EmbeddingManager(store=store)
    Q = unit(rng, 6)
    Q[0] = raw["e7"] / np.linalg.norm(raw["e7"]); Q[1] = raw["v3"] / np.linalg.norm(raw["v3"])

    def check_company(c):
        view, meta = mgr.get_matcher_for_company(c)
        emb, meta_ref = mgr.get_embeddings_for_company(c) if c is not None else mgr.get_all()
        assert list(meta) == list(emb)
        check(view, emb, Q)
        return view

    v_acme = check_company("acme"); check_company("globex"); check_company(None)
    assert mgr.get_matcher_for_company("acme")[0] is v_acme          # cached between syncs
    uploads = []
    orig = mgr._gallery.upsert
    mgr._gallery.upsert = lambda ids, rows, normalise=False: (uploads.append(list(ids)), orig(ids, rows, normalise))[1]
    # documents change: one employee re-enrolled, one new visitor, one employee blacklisted
    later = datetime.utcnow() + timedelta(seconds=5)
    new7 = rng.standard_normal(512).astype(np.float32)
    import pickle
    store.employee_blobs["e7"] = pickle.dumps(new7)
    next(d for d in store.employees if d["_id"] == "e7")["lastUpdated"] = later
    store.add_visitor("v_new", "acme", Q[2] * 2, name="New")["lastUpdated"] = later
    next(d for d in store.employees if d["_id"] == "e9")["blacklisted"] = True
    slot7 = mgr._gallery.slot_of["e7"]
    mgr.force_sync()
    check_company("acme"); check_company("globex"); check_company(None)
    assert uploads == [["e7", "v_new"]]                               # only the changed rows travelled
    assert mgr._gallery.slot_of["e7"] == slot7 and "e9" not in mgr._gallery.slot_of
    view, meta = mgr.get_matcher_for_company("acme")
    ids, score, idx = view.match(Q, thr=0.4)
    assert ids[2] == "v_new" and ids[0] is None                      # e7's old row is gone, the new visitor matches
    mgr.last_sync_time = later + timedelta(seconds=1)                 # (the test's documents are dated in the future)
    mgr.force_sync()                                                  # nothing changed: nothing uploaded
    check_company("acme")
    assert uploads == [["e7", "v_new"]]
