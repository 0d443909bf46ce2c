"""Application classes of the reference's inference script, kept name-for-name so it drops in:
``EmbeddingManager`` (/root/reference/infrenceServer.py:36-398), ``FaceRecognitionProcessor``
(:400-563) and the counting-path ``CameraProcessor`` (/root/reference/peopleCount.py:822-896).

Differences that are the point of this build:
* the gallery is a device-resident ``[N,512]`` matrix per company, rebuilt on sync, not a Python
  dict scanned per face (rows a-7/a-9 of SURVEY.md section 8);
* the store is INJECTED (the reference connects to a remote MongoDB at import,
  infrenceServer.py:682 / db/__init__.py:7-9 - SURVEY.md F5);
* row order is explicit and deterministic: employees in store order, then visitors
  (infrenceServer.py:264,288); ties go to the lowest row (the reference iterates a ``set``, so its
  own tie order is random: infrenceServer.py:357-373).
"""
import logging
import pickle
import threading
from datetime import datetime

import numpy as np

logger = logging.getLogger(__name__)


class InMemoryStore:
    """Store protocol used by EmbeddingManager (what the Mongo collections + GridFS buckets of
    /root/reference/db/__init__.py:11-26 provide).  Documents use the reference's field names."""

    def __init__(self):
        self.employees, self.visitors = [], []          # documents, insertion order = cursor order
        self.employee_blobs, self.visitor_blobs = {}, {}

    def add_employee(self, _id, company_id, embedding, name="Unknown", status="active", blacklisted=False, **extra):
        self.employee_blobs[_id] = pickle.dumps(np.asarray(embedding, np.float32))
        doc = {"_id": _id, "companyId": company_id, "status": status, "blacklisted": blacklisted,
               "employeeName": name, "lastUpdated": datetime.utcnow(),
               "employeeEmbeddings": {"buffalo_l": {"embeddingId": _id, "status": "done"}}}
        doc.update(extra)
        self.employees.append(doc)
        return doc

    def add_visitor(self, _id, company_id, embedding, name="Unknown", status="done", **extra):
        self.visitor_blobs[_id] = pickle.dumps(np.asarray(embedding, np.float32))
        doc = {"_id": _id, "companyId": company_id, "visitorName": name, "lastUpdated": datetime.utcnow(),
               "visitorEmbeddings": {"buffalo_l": {"embeddingId": _id, "status": status}}}
        doc.update(extra)
        self.visitors.append(doc)
        return doc

    # ---- protocol
    def find_employees(self, since=None):
        """status == done embeddings of active, non-blacklisted employees (infrenceServer.py:66-73,215-219)."""
        return [d for d in self.employees
                if d.get("employeeEmbeddings", {}).get("buffalo_l", {}).get("status") == "done"
                and d.get("status") == "active" and not d.get("blacklisted")
                and (since is None or d["lastUpdated"] > since)]

    def find_visitors(self, since=None):
        return [d for d in self.visitors
                if d.get("visitorEmbeddings", {}).get("buffalo_l", {}).get("status") == "done"
                and (since is None or d["lastUpdated"] > since)]

    def find_inactive_employee_ids(self):
        """infrenceServer.py:236-242."""
        return [str(d["_id"]) for d in self.employees if d.get("status") != "active" or d.get("blacklisted")]

    def read_employee_embedding(self, embedding_id):
        return self.employee_blobs[embedding_id]

    def read_visitor_embedding(self, embedding_id):
        return self.visitor_blobs[embedding_id]

    def company_member_ids(self, company_id):
        """(active non-blacklisted employee ids, visitor ids) of a company, store order
        (infrenceServer.py:351-367)."""
        emp = [str(d["_id"]) for d in self.employees
               if d["companyId"] == company_id and d.get("status") == "active" and not d.get("blacklisted")]
        vis = [str(d["_id"]) for d in self.visitors if d["companyId"] == company_id]
        return emp, vis


class EmbeddingManager:
    """Same public surface as infrenceServer.py:36-398; ``store`` replaces the Mongo connection."""

    def __init__(self, mongodb_uri=None, database_name=None, store=None, device="cuda:0", sync_interval=30):
        if store is None:
            raise ValueError("EmbeddingManager needs store=<store object> (the reference's module-level MongoDB "
                             "connection is not reproduced; see INTEGRATION.md for a pymongo adapter)")
        self.store, self.device = store, device
        self.embeddings = {}                 # id -> unit float32[512]; dict order = row order
        self.employee_metadata = {}
        self.embeddings_lock = threading.Lock()
        self.last_sync_time = None
        self.is_initial_load = True
        self.sync_interval = sync_interval
        self.sync_thread, self.running = None, False
        self._gallery = None                 # DeviceGallery: one slab of rows in HBM, updated in place
        self._dirty, self._gone = {}, set()  # ids changed / removed since the slab was last brought up to date
        self._matchers = {}                  # company id -> (GalleryView, metadata), rebuilt after a sync
        self._initial_load()

    # ---- ingest (infrenceServer.py:260-341): unpickle, divide by the norm
    def _ingest(self, docs, kind):
        for d in docs:
            try:
                _id = str(d["_id"])
                entry = d[f"{kind}Embeddings"]["buffalo_l"]
                read = self.store.read_employee_embedding if kind == "employee" else self.store.read_visitor_embedding
                embedding = pickle.loads(read(entry["embeddingId"]))
                self.embeddings[_id] = embedding / np.linalg.norm(embedding)
                self._dirty[_id] = None
                self._gone.discard(_id)
                if kind == "employee":
                    self.employee_metadata[_id] = {
                        "name": d.get("employeeName", "Unknown"), "employeeId": d.get("employeeId", "Unknown"),
                        "email": d.get("employeeEmail", ""), "mobile": d.get("employeeMobile", ""),
                        "type": "employee", "lastUpdated": d.get("lastUpdated", datetime.utcnow())}
                else:
                    self.employee_metadata[_id] = {"name": d.get("visitorName", "Unknown"), "type": "visitor",
                                                   "lastUpdated": d.get("lastUpdated", datetime.utcnow())}
            except Exception as e:          # the reference logs and continues (infrenceServer.py:285-286)
                logger.error("error loading %s embedding for %s: %s", kind, d.get("_id"), e)

    def _load_updated_embeddings(self, employees, visitors):
        with self.embeddings_lock:
            self._ingest(employees, "employee")
            self._ingest(visitors, "visitor")
            self._matchers.clear()

    def _initial_load(self):
        self._load_updated_embeddings(self.store.find_employees(), self.store.find_visitors())
        self.last_sync_time = datetime.utcnow()
        self.is_initial_load = False

    def _remove_inactive_embeddings(self):
        with self.embeddings_lock:
            for _id in self.store.find_inactive_employee_ids():
                if _id in self.embeddings:
                    del self.embeddings[_id]
                    self.employee_metadata.pop(_id, None)
                    self._dirty.pop(_id, None)
                    self._gone.add(_id)
            self._matchers.clear()

    def _sync_embeddings(self):
        """infrenceServer.py:185-232: incremental, keyed on lastUpdated."""
        since = self.last_sync_time
        now = datetime.utcnow()
        self._load_updated_embeddings(self.store.find_employees(since), self.store.find_visitors(since))
        self._remove_inactive_embeddings()
        self.last_sync_time = now

    def force_sync(self):
        self._sync_embeddings()

    def start_sync(self):
        if self.sync_thread is None or not self.sync_thread.is_alive():
            self.running = True
            self._stop = threading.Event()
            self.sync_thread = threading.Thread(target=self._sync_loop, daemon=True)
            self.sync_thread.start()

    def stop_sync(self):
        self.running = False
        if self.sync_thread:
            self._stop.set()
            self.sync_thread.join(timeout=5)

    def _sync_loop(self):
        while self.running and not self._stop.wait(self.sync_interval):
            try:
                self._sync_embeddings()
            except Exception as e:
                logger.error("error in sync loop: %s", e)

    # ---- views
    def _company_rows(self, company_id):
        emp, vis = self.store.company_member_ids(company_id)
        seen = set(emp)
        ids = [i for i in emp if i in self.embeddings] + [i for i in vis if i in self.embeddings and i not in seen]
        return ids

    def get_embeddings_for_company(self, company_id):
        """(dict id -> float32[512], dict id -> metadata), as infrenceServer.py:343-380."""
        with self.embeddings_lock:
            ids = self._company_rows(company_id)
            return ({i: self.embeddings[i] for i in ids}, {i: self.employee_metadata[i] for i in ids})

    def get_all(self):
        """peopleCount.py:816-819."""
        with self.embeddings_lock:
            return dict(self.embeddings), dict(self.employee_metadata)

    def _flush_to_device(self):
        """Bring the device slab up to date: only rows that changed since the last flush travel (in-place row
        writes); removed people free their slots.  Called under the lock."""
        from .gallery import DeviceGallery
        if self._gallery is None:
            self._gallery = DeviceGallery(self.device, capacity=max(1024, 2 * len(self.embeddings)))
        if self._gone:
            self._gallery.remove(self._gone)
            self._gone = set()
        if self._dirty:
            ids = list(self._dirty)
            self._gallery.upsert(ids, np.stack([self.embeddings[i] for i in ids]), normalise=False)  # unit at ingest
            self._dirty = {}

    def get_matcher_for_company(self, company_id):
        """(view, metadata): the company's rows as an ordered view of the device-resident slab.  The membership
        queries the reference issues per frame (infrenceServer.py:351-367) run once per sync; rows are never
        copied per company, and a sync rewrites only the rows whose documents changed."""
        with self.embeddings_lock:
            self._flush_to_device()
            hit = self._matchers.get(company_id)
            if hit is None or hit[0].generation != self._gallery.generation:
                ids = self._company_rows(company_id) if company_id is not None else list(self.embeddings)
                hit = (self._gallery.view(ids), {i: self.employee_metadata[i] for i in ids})
                self._matchers[company_id] = hit
            return hit

    def get_stats(self):
        """infrenceServer.py:386-398."""
        with self.embeddings_lock:
            employees = sum(1 for m in self.employee_metadata.values() if m["type"] == "employee")
            visitors = sum(1 for m in self.employee_metadata.values() if m["type"] == "visitor")
            return {"total_embeddings": len(self.embeddings), "employees": employees, "visitors": visitors,
                    "last_sync": self.last_sync_time.isoformat() if self.last_sync_time else None,
                    "initial_load_complete": not self.is_initial_load}


class FaceRecognitionProcessor:
    """infrenceServer.py:400-563."""

    def __init__(self, embedding_manager, face_detector=None):
        self.embedding_manager = embedding_manager
        self.face_detector = face_detector
        self.detection_threshold = 0.3        # set but never read in the reference (:406)
        self.recognition_threshold = 0.4

    def initialize_detector(self):
        if self.face_detector is None:
            from .face_analysis import FaceAnalysis
            self.face_detector = FaceAnalysis(name="buffalo_l",
                                              providers=["CUDAExecutionProvider", "CPUExecutionProvider"])
            self.face_detector.prepare(ctx_id=0)

    def recognize(self, frame, company_id):
        """Structured form: list of dicts {bbox int[4], person_info, det_score, recognition_score, person_id}."""
        if self.face_detector is None:
            self.initialize_detector()
        matcher, metadata = self.embedding_manager.get_matcher_for_company(company_id)
        if len(matcher) == 0:
            logger.warning("No embeddings found for company %s", company_id)
            return None
        r = _detect_embed(self.face_detector, frame)
        matcher, metadata, idx, score = _match_fresh(self.embedding_manager, company_id, matcher, metadata,
                                                     r["normed_embedding"])  # renormalise + scan (:532-542)
        dec = matcher.decide_device(idx, score, self.recognition_threshold)  # :545
        idx, score, dec = idx.cpu().numpy(), score.cpu().numpy(), dec.cpu().numpy()
        bbox = r["bbox"].cpu().numpy().astype(int)                           # :531 truncation
        det = r["det_score"].cpu().numpy()
        out = []
        for f in range(len(idx)):
            if dec[f] == 1:
                pid = matcher.ids[idx[f]]
                out.append({"bbox": bbox[f], "person_id": pid, "person_info": metadata[pid],
                            "det_score": float(det[f]), "recognition_score": score[f]})
            else:
                out.append({"bbox": bbox[f], "person_id": None, "person_info": {"name": "Unknown", "type": "unknown"},
                            "det_score": float(det[f]), "recognition_score": 0})
        return out

    def recognize_batch(self, frames, company_id):
        """Batch form for the camera batcher (``camera.CameraManager``): ``frames`` = list of same-sized BGR uint8
        frames (one per camera).  ONE pass of the sync-free slot pipeline over the whole batch (pinned staging ->
        copy stream -> detect -> align -> embed -> company view scan -> decision), one host synchronisation at the
        end.  Returns a list (per frame) of lists of the dicts ``recognize`` returns, or None when the company has
        no gallery (the reference returns the frame untouched then, infrenceServer.py:523-525)."""
        import torch
        from .ingest import FrameIngest
        if self.face_detector is None:
            self.initialize_detector()
        det = self.face_detector
        matcher, metadata = self.embedding_manager.get_matcher_for_company(company_id)
        if len(matcher) == 0:
            logger.warning("No embeddings found for company %s", company_id)
            return None
        arr = [np.ascontiguousarray(np.asarray(f)) for f in frames]
        n, (h, w) = len(arr), arr[0].shape[:2]
        if any(a.shape != (h, w, 3) or a.dtype != np.uint8 for a in arr):
            raise ValueError("frames of a batch must be uint8 [H,W,3] of one size")
        stages = self.__dict__.setdefault("_stages", {})
        key = (n, h, w)
        if key not in stages:
            while len(stages) >= 8:                       # a few (cameras present, frame size) shapes at most: drop the oldest ring
                stages.pop(next(iter(stages)))
            stages[key] = [FrameIngest(n, h, w, det.device, depth=2), 0]
        ring, turn = stages[key]
        stages[key][1] = turn + 1
        with det._lock, torch.cuda.device(det.device):
            host = ring.host_buffer(turn)
            for i, a in enumerate(arr):
                host[i] = a
            dev, ready = ring.upload(turn)
            r = det.detect_embed_slots(dev, ready_event=ready, compact_embed=True)      # results go to the host anyway
            ring.release(turn)
        matcher, metadata, idx, score = _match_fresh(self.embedding_manager, company_id, matcher, metadata,
                                                     r["normed_embedding"])
        dec = matcher.decide_device(idx, score, self.recognition_threshold)
        cap = r["bbox"].shape[1]
        counts = r["counts"].cpu().numpy()                                   # the one synchronisation
        idx, score, dec = (t.cpu().numpy().reshape(n, cap) for t in (idx, score, dec))
        bbox = r["bbox"].cpu().numpy().astype(int)                           # :531 truncation
        dscore = r["det_score"].cpu().numpy()
        out = []
        for f in range(n):
            faces = []
            for j in range(int(counts[f])):
                if dec[f, j] == 1:
                    pid = matcher.ids[idx[f, j]]
                    faces.append({"bbox": bbox[f, j], "person_id": pid, "person_info": metadata[pid],
                                  "det_score": float(dscore[f, j]), "recognition_score": score[f, j]})
                else:
                    faces.append({"bbox": bbox[f, j], "person_id": None,
                                  "person_info": {"name": "Unknown", "type": "unknown"},
                                  "det_score": float(dscore[f, j]), "recognition_score": 0})
            out.append(faces)
        return out

    def annotate(self, frame, results):
        """Draw ``recognize`` / ``recognize_batch`` results onto a frame (colours of infrenceServer.py:546-553)."""
        for r in results or ():
            t = r["person_info"]["type"]
            color = (0, 255, 0) if t == "employee" else (0, 255, 255) if t == "visitor" else (0, 0, 255)
            frame = self.draw_enhanced_bounding_box(frame, r["bbox"], color, r["person_info"], r["det_score"],
                                                    r["recognition_score"])
        return frame

    def draw_enhanced_bounding_box(self, frame, bbox, color, person_info, detection_score, recognition_score):
        """The reference's HUD (infrenceServer.py:418-513) is presentation and out of scope; this
        draws a plain 2-px box so ``recognize_faces`` still returns a marked frame."""
        h, w = frame.shape[:2]
        x1, y1, x2, y2 = [int(v) for v in bbox]
        x1, x2 = max(0, min(x1, w - 1)), max(0, min(x2, w - 1))
        y1, y2 = max(0, min(y1, h - 1)), max(0, min(y2, h - 1))
        frame[y1:y1 + 2, x1:x2 + 1] = color; frame[max(y2 - 1, 0):y2 + 1, x1:x2 + 1] = color
        frame[y1:y2 + 1, x1:x1 + 2] = color; frame[y1:y2 + 1, max(x2 - 1, 0):x2 + 1] = color
        return frame

    def recognize_faces(self, frame, company_id):
        try:
            results = self.recognize(frame, company_id)
            if results is None:
                return frame
            frame = self.annotate(frame, results)
        except Exception as e:                # errors are logged and swallowed (:560-563)
            logger.error("Error during face recognition: %s", e)
        return frame


class CameraProcessor:
    """Counting path, peopleCount.py:822-896: whole gallery, 0.45 recognised / < 0.35 unknown,
    the [0.35, 0.45) band is dropped."""

    def __init__(self, embedding_manager, manager, face_detector=None):
        self.embedding_manager, self.manager, self.face_detector = embedding_manager, manager, face_detector
        self.recognition_threshold = 0.45
        self.unknown_threshold = 0.35

    def initialize_detector(self):
        if self.face_detector is None:
            from .face_analysis import FaceAnalysis
            self.face_detector = FaceAnalysis(name="buffalo_l").prepare(ctx_id=0)

    def process_frame(self, frame, camera_id):
        if self.face_detector is None:
            self.initialize_detector()
        matcher, metadata = self.embedding_manager.get_matcher_for_company(None)
        if len(matcher) == 0:
            return {"faces": 0, "recognized": 0, "unknown": 0}
        timestamp = datetime.utcnow()
        stats = {"faces": 0, "recognized": 0, "unknown": 0}
        try:
            r = _detect_embed(self.face_detector, frame)
            matcher, metadata, idx, score = _match_fresh(self.embedding_manager, None, matcher, metadata,
                                                         r["normed_embedding"])
            dec = matcher.decide_device(idx, score, self.recognition_threshold, self.unknown_threshold).cpu().numpy()
            idx, score = idx.cpu().numpy(), score.cpu().numpy()
            stats["faces"] = len(idx)
            q = None
            for f in range(len(idx)):
                if dec[f] == 1:
                    pid = matcher.ids[idx[f]]
                    self.manager.process_detection(pid, metadata[pid], camera_id, timestamp, float(score[f]))
                    stats["recognized"] += 1
                elif dec[f] == 0:
                    if q is None:             # re-normalised query rows, as handed on by peopleCount.py:863,884
                        e = r["normed_embedding"].cpu().numpy()
                        q = e / np.linalg.norm(e, axis=1, keepdims=True)
                        bb = r["bbox"].cpu().numpy().astype(int)
                    self.manager.process_unknown_detection(camera_id, timestamp, q[f], bb[f].tolist())
                    stats["unknown"] += 1
        except Exception as e:
            logger.error("Error in face detection: %s", e)
        return stats


def _detect_embed(detector, frame):
    """detect -> align -> embed of one frame under the engine's lock: one engine may be shared by threads
    (/root/reference/trainingServer.py:115,227) and ``get_batch`` takes the same lock."""
    lock = getattr(detector, "_lock", None)
    dev = _to_device(frame, detector.device)
    if lock is None:
        return detector.detect_embed_device(dev)
    with lock:
        return detector.detect_embed_device(dev)


def _match_fresh(manager, company_id, matcher, metadata, Q):
    """Scan through the company's view; a sync on another thread may have changed the slab's membership since the
    view was fetched (its generation moved on): fetch the current view once and retry instead of dropping the frame."""
    from .gallery import StaleViewError
    try:
        idx, score = matcher.match_device(Q)
    except StaleViewError:
        matcher, metadata = manager.get_matcher_for_company(company_id)
        idx, score = matcher.match_device(Q)
    return matcher, metadata, idx, score


def _to_device(frame, device):
    import torch
    a = np.ascontiguousarray(np.asarray(frame))
    if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:
        raise ValueError("frame must be uint8 [H,W,3] BGR")
    return torch.from_numpy(a).to(device)[None]
