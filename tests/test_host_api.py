"""Host logic without a GPU: EmbeddingManager ingest/sync/eviction/views, HTTP surface."""
import pickle
import threading
import time

import numpy as np
import pytest

from facerecognition_infrenceengine_amd.processor import EmbeddingManager, InMemoryStore
from facerecognition_infrenceengine_amd.server import create_app
from oracle import match as omatch


def _store():
    rng = np.random.default_rng(0)
    s = InMemoryStore()
    for i in range(5):
        s.add_employee(f"e{i}", "c1" if i < 3 else "c2", rng.standard_normal(512) * 3, name=f"E{i}")
    s.add_employee("e_black", "c1", rng.standard_normal(512), blacklisted=True)
    s.add_visitor("v0", "c1", rng.standard_normal(512), name="V0")
    s.add_visitor("v_pending", "c1", rng.standard_normal(512), status="queued")
    return s


def test_ingest_normalises_and_orders_rows():
    s = _store()
    m = EmbeddingManager(store=s)
    assert list(m.embeddings) == ["e0", "e1", "e2", "e3", "e4", "v0"]        # employees then visitors
    for _id, row in m.embeddings.items():
        blob = (s.employee_blobs if _id.startswith("e") else s.visitor_blobs)[_id]
        assert np.array_equal(row, omatch.gallery_row_load(blob))               # infrenceServer.py:270-271
    emb, meta = m.get_embeddings_for_company("c1")
    assert list(emb) == ["e0", "e1", "e2", "v0"] and meta["v0"]["type"] == "visitor"
    st = m.get_stats()
    assert st == {"total_embeddings": 6, "employees": 5, "visitors": 1, "last_sync": st["last_sync"],
                  "initial_load_complete": True}


def test_sync_picks_up_updates_and_evicts():
    s = _store()
    m = EmbeddingManager(store=s)
    time.sleep(0.01)
    s.add_employee("e_new", "c1", np.ones(512), name="New")
    s.employees[1]["status"] = "inactive"                                     # e1 leaves
    m.force_sync()
    assert "e_new" in m.embeddings and "e1" not in m.embeddings
    assert list(m.get_embeddings_for_company("c1")[0]) == ["e0", "e2", "e_new", "v0"]


def test_device_slab_bookkeeping_uploads_only_changed_rows(monkeypatch):
    """SURVEY.md 8f row 2, host half (no GPU): which rows travel / which slots are freed on each sync."""
    from facerecognition_infrenceengine_amd import gallery

    class FakeSlab:
        def __init__(self, device, capacity=0):
            self.log, self.slot_of, self.generation = [], {}, 0
        def upsert(self, ids, rows, normalise=False):
            self.log.append(("upsert", list(ids)))
            assert not normalise and np.allclose(np.linalg.norm(rows, axis=1), 1, atol=1e-6)
            for i in ids:
                if i not in self.slot_of:
                    self.slot_of[i] = len(self.slot_of); self.generation += 1
        def remove(self, ids):
            self.log.append(("remove", sorted(ids)))
            for i in ids:
                self.slot_of.pop(i, None)
            self.generation += 1
        def view(self, ids):
            v = type("V", (), {})(); v.ids, v.generation = list(ids), self.generation
            return v

    monkeypatch.setattr(gallery, "DeviceGallery", FakeSlab)
    s = _store()
    m = EmbeddingManager(store=s)
    v, meta = m.get_matcher_for_company("c1")
    assert v.ids == ["e0", "e1", "e2", "v0"] and list(meta) == v.ids
    assert m._gallery.log == [("upsert", ["e0", "e1", "e2", "e3", "e4", "v0"])]
    assert m.get_matcher_for_company("c1")[0] is v                           # cached until the next sync
    time.sleep(0.01)
    s.add_employee("e_new", "c1", np.ones(512), name="New")
    s.employees[1]["status"] = "inactive"
    m.force_sync()
    v2, _ = m.get_matcher_for_company("c1")
    assert v2.ids == ["e0", "e2", "e_new", "v0"]
    assert m._gallery.log[1:] == [("remove", ["e1"]), ("upsert", ["e_new"])]  # nothing else travelled
    m.force_sync()
    m.get_matcher_for_company("c1")
    assert len(m._gallery.log) == 3                                          # idle sync: no device traffic


def test_requires_injected_store():
    with pytest.raises(ValueError):
        EmbeddingManager("mongodb://example", "db")


def test_http_surface_matches_reference():
    s = _store()
    m = EmbeddingManager(store=s)
    started, stopped = [], []

    class Cam:
        def start_cameras(self, sources, company_id): started.append((sources, company_id))
        def stop_cameras(self): stopped.append(1)
    c = create_app(m, Cam()).test_client()
    r = c.get("/api/embeddings/stats")
    assert r.status_code == 200 and set(r.json) == {"total_embeddings", "employees", "visitors", "last_sync",
                                                    "initial_load_complete"}
    assert r.headers["Access-Control-Allow-Origin"] == "*"
    r = c.post("/api/embeddings/sync")
    assert r.status_code == 200 and r.json == {"status": "success", "message": "Sync completed"}
    r = c.post("/api/camera/start", json={"sources": [0]})
    assert r.status_code == 400 and r.json == {"status": "error", "message": "Company ID required"}
    r = c.post("/api/camera/start", json={"company_id": "c1"})
    assert r.status_code == 200 and r.json == {"status": "success", "message": "Camera started"}
    for _ in range(100):
        if started:
            break
        time.sleep(0.01)
    assert started == [([0], "c1")]
    r = c.post("/api/camera/stop")
    assert r.json == {"status": "success", "message": "Camera stopped"} and stopped == [1]
    m.force_sync = lambda: (_ for _ in ()).throw(RuntimeError("db down"))
    r = c.post("/api/embeddings/sync")
    assert r.status_code == 500 and r.json == {"status": "error", "message": "db down"}


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from facerecognition_infrenceengine_amd import FaceAnalysis, GalleryMatcher
    from facerecognition_infrenceengine_amd._lib import FrError
    with pytest.raises(FrError):
        FaceAnalysis(name="x").prepare(ctx_id=0)
    with pytest.raises(FrError):
        GalleryMatcher("cuda:0")


def test_camera_manager_batches_sources_and_keeps_reference_queue_semantics():
    """CameraManager host logic with fake cameras and a fake processor (no GPU): one frame per source per turn, in
    source order; frame queues of 2 drop when full; results are (source, frame) pairs; start/stop are idempotent."""
    import queue
    import time
    from facerecognition_infrenceengine_amd.camera import CameraManager

    class FakeCap:
        def __init__(self, source, n):
            self.source, self.n, self.i, self.exhausted = source, n, 0, False

        def read(self):
            if self.i >= self.n:
                self.exhausted = True
                time.sleep(0.001)
                return False, None
            self.i += 1
            return True, np.full((4, 6, 3), self.source * 10 + self.i % 10, np.uint8)

        def release(self):
            self.released = True

    class FakeProcessor:
        def __init__(self):
            self.batches = []

        def recognize_batch(self, frames, company_id):
            self.batches.append((len(frames), company_id))
            time.sleep(0.002)
            return [[{"tag": int(f[0, 0, 0])}] for f in frames]

        def annotate(self, frame, res):
            out = frame.copy(); out[0, 0, 0] = 255 - res[0]["tag"]
            return out

    proc = FakeProcessor()
    caps = {}
    cm = CameraManager(embedding_manager=None, processor=proc,
                       capture_factory=lambda s: caps.setdefault(s, FakeCap(s, 40)))
    # take_batch on hand-filled queues: order and "at most one per source"
    cm.running = True
    cm.frame_queues = {s: queue.Queue(maxsize=2) for s in (3, 1, 2)}
    cm.result_queue = queue.Queue(maxsize=10)
    cm.frame_queues[1].put(np.zeros((4, 6, 3), np.uint8)); cm.frame_queues[1].put(np.ones((4, 6, 3), np.uint8))
    cm.frame_queues[3].put(np.full((4, 6, 3), 7, np.uint8))
    b = cm.take_batch([3, 1, 2])
    assert [s for s, _ in b] == [3, 1] and int(b[1][1][0, 0, 0]) == 0 and cm.frame_queues[1].qsize() == 1
    res = cm.process_batch(b, "acme")
    assert proc.batches == [(2, "acme")] and [r[0]["tag"] for r in res] == [7, 0]
    assert [cm.result_queue.get_nowait()[0] for _ in range(2)] == [3, 1]
    cm.running = False
    # full run with threads
    seen = []
    cm.start_cameras([0, 1, 2], "acme", display=lambda s, f: seen.append((s, int(f[0, 0, 0]))))
    cm.start_cameras([0, 1, 2], "acme")                      # second start is a no-op
    deadline = time.time() + 10
    while time.time() < deadline and not all(getattr(c, "exhausted", False) for c in caps.values()):
        time.sleep(0.01)
    time.sleep(0.1)
    cm.stop_cameras(); cm.stop_cameras()
    assert cm.stats["frames"] >= 3 and cm.stats["largest_batch"] >= 2 and all(n <= 3 for n, _ in proc.batches)
    assert len(seen) >= 3 and {s for s, _ in seen} <= {0, 1, 2}
    assert all(getattr(c, "released", False) for c in caps.values())


def test_camera_manager_backs_off_on_failed_reads_and_drains_results_by_default():
    """A camera whose read() fails must not spin (capture threads share the interpreter with the batching loop): the
    reads back off, the capture is re-opened after ``reopen_after`` failures and the source is given up after
    ``dead_after``; without a display the default consumer keeps the latest frame per source, so the result queue
    (maxsize 10) never fills and nothing is dropped; start without a capture_factory raises in the caller."""
    import time
    import pytest
    from facerecognition_infrenceengine_amd.camera import CameraManager

    class FlakyCap:
        opened = 0

        def __init__(self, source):
            FlakyCap.opened += 1
            self.source, self.reads, self.generation = source, 0, FlakyCap.opened

        def read(self):
            self.reads += 1
            if self.source == "dead" or (self.source == "flaky" and self.generation == 1):
                return False, None                              # a dropped stream: fails immediately, every time
            time.sleep(0.001)                                   # a live camera blocks in read() (and yields the interpreter)
            return True, np.full((4, 6, 3), 5, np.uint8)

        def release(self):
            pass

    class Proc:
        def recognize_batch(self, frames, company_id):
            time.sleep(0.001)
            return [[] for _ in frames]

        def annotate(self, frame, res):
            return frame

    with pytest.raises(RuntimeError):
        CameraManager(None, processor=Proc()).start_cameras([0], "acme")
    caps = []
    cm = CameraManager(None, processor=Proc(), capture_factory=lambda s: caps.append(FlakyCap(s)) or caps[-1],
                       reopen_after=3, dead_after=8)
    cm.start_cameras(["flaky", "dead", "good"], "acme")
    deadline = time.time() + 30
    while time.time() < deadline and not (cm.stats["dead_sources"] and cm.latest_frame("flaky") is not None
                                          and cm.stats["frames"] > 30):
        time.sleep(0.01)
    cm.stop_cameras()
    assert cm.stats["dead_sources"] == ["dead"]
    assert cm.latest_frame("flaky") is not None and cm.latest_frame("good") is not None and cm.latest_frame("dead") is None
    first_flaky = next(c for c in caps if c.source == "flaky")
    assert first_flaky.reads == 3                               # re-opened after 3 failures, not spun on
    dead_reads = sum(c.reads for c in caps if c.source == "dead")
    assert dead_reads == 8                                      # 8 backed-off reads in total, then given up
    assert cm.stats["frames"] > 30 and cm.stats["dropped_results"] == 0


def test_decode_image_is_the_imdecode_of_the_enrolment_path():
    """ingest.decode_image: encoded bytes -> BGR uint8 (trainingServer.py:219-221).  PNG round trip is lossless and
    must be exact incl. the channel order; JPEG decodes to within the codec's error; garbage gives None."""
    import io
    from PIL import Image
    from facerecognition_infrenceengine_amd.ingest import decode_image
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(bgr[:, :, ::-1]).save(buf, format="PNG")
    got = decode_image(buf.getvalue())
    assert got.dtype == np.uint8 and got.flags["C_CONTIGUOUS"] and np.array_equal(got, bgr)
    smooth = np.clip(np.add.outer(np.arange(64), np.arange(80))[:, :, None] + np.array([0, 40, 90]), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(smooth[:, :, ::-1]).save(buf, format="JPEG", quality=95)
    got = decode_image(buf.getvalue())
    assert got.shape == smooth.shape and np.abs(got.astype(int) - smooth.astype(int)).max() <= 6
    grey = io.BytesIO()
    Image.fromarray(smooth[:, :, 0]).save(grey, format="PNG")                 # IMREAD_COLOR: grey -> 3 equal channels
    g3 = decode_image(grey.getvalue())
    assert g3.shape == smooth.shape and np.array_equal(g3[:, :, 0], g3[:, :, 2])
    assert decode_image(b"not an image") is None
