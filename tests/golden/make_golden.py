"""Generate tests/golden/*.npz.  Run HERE (container with /root/reference), commit the output.

Two kinds of vectors:

1. REFERENCE-PINNED (match / decision / gallery row / enrolment / unknown cluster):
   the reference's own methods are AST-extracted from /root/reference (module
   top-levels are never executed: they import absent packages and dial a remote
   database) and run with the third-party model and the database
   replaced by in-memory fakes.  Their inputs and outputs are stored.
2. ORACLE-DEFINED (r100 / mtcnn / align): outputs of oracle/ with seeded synthetic
   weights ("parity unpinned" at the insightface boundary).

Only arrays are written; no reference source text is stored.
"""
import ast
import os
import pickle
import sys
import types
from collections import OrderedDict, deque
from datetime import datetime
from typing import Dict, List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


class _Log:
    def __getattr__(self, name):
        return lambda *a, **k: None


def extract(path, class_name, names, extra_ns=None):
    """Compile selected methods / a whole class out of a reference file."""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    ns = {"np": np, "logger": _Log(), "List": List, "Tuple": Tuple, "Dict": Dict,
          "Optional": Optional, "datetime": datetime, "deque": deque, "pickle": pickle}
    ns.update(extra_ns or {})
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            if names is None:
                mod = ast.Module(body=[node], type_ignores=[])
                exec(compile(mod, path, "exec"), ns)
                return ns[class_name]
            out = {}
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name in names:
                    mod = ast.Module(body=[fn], type_ignores=[])
                    exec(compile(mod, path, "exec"), ns)
                    out[fn.name] = ns[fn.name]
            return out
    raise KeyError(class_name)


class FakeFace:
    def __init__(self, bbox, emb, det):
        self.bbox = np.asarray(bbox, np.float32)
        self.normed_embedding = emb
        self.det_score = det


def unit(v):
    return (v / np.linalg.norm(v)).astype(np.float32)


def make_match():
    rng = np.random.default_rng(7)
    out = {}
    for tag, N in (("g100", 100), ("g1000", 1000)):
        G = rng.standard_normal((N, 512)).astype(np.float32)
        G /= np.linalg.norm(G, axis=1, keepdims=True)
        Q = rng.standard_normal((8, 512)).astype(np.float32)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        # planted matches (score ~0.9), exact-tie duplicate rows, threshold neighbours
        G[11] = unit(Q[0] + 0.02 * rng.standard_normal(512))
        G[40] = unit(Q[1] + 0.02 * rng.standard_normal(512))
        G[17] = unit(Q[2] + 0.02 * rng.standard_normal(512)); G[63] = G[17]   # exact tie -> first wins
        r = unit(rng.standard_normal(512) - Q[3] * 0)                           # orthogonalise
        r = unit(r - np.dot(r, Q[3]) * Q[3])
        for qi, c in ((3, 0.4 - 1e-4), (4, 0.4 + 1e-4), (5, 0.45 + 1e-4), (6, 0.35 - 1e-4)):
            o = unit(rng.standard_normal(512)); o = unit(o - np.dot(o, Q[qi]) * Q[qi])
            G[70 + qi] = unit(c * Q[qi] + np.sqrt(1 - c * c) * o)
        # Q[7]: no match (random), Q rows are "normed_embedding" as insightface hands them over:
        # unit up to float32 rounding; perturb scale by a few ulp to exercise the re-normalise
        Qn = (Q * np.float32(1.0000002)).astype(np.float32)
        ids = [f"id{n:05d}" for n in range(N)]
        embeddings = OrderedDict((ids[n], G[n]) for n in range(N))
        metadata = OrderedDict((ids[n], {"name": f"P{n}", "type": "employee" if n % 3 else "visitor",
                                         "pid": n}) for n in range(N))
        faces = [FakeFace([10.7 + f, 20.2, 110.9, 140.5], Qn[f], 0.9) for f in range(8)]

        # --- live path: FaceRecognitionProcessor.recognize_faces
        fn = extract("infrenceServer.py", "FaceRecognitionProcessor", {"recognize_faces"})["recognize_faces"]
        cap = []

        def draw(frame, bbox, color, person_info, det_score, rec_score):
            cap.append((person_info.get("pid", -1), float(rec_score), [int(b) for b in bbox], color))
            return frame
        me = types.SimpleNamespace(
            face_detector=types.SimpleNamespace(get=lambda fr: faces),
            embedding_manager=types.SimpleNamespace(
                get_embeddings_for_company=lambda cid: (embeddings, metadata)),
            recognition_threshold=0.4, draw_enhanced_bounding_box=draw,
            initialize_detector=lambda: None)
        fn(me, np.zeros((4, 4, 3), np.uint8), "c0")
        out[f"{tag}_G"] = G; out[f"{tag}_Q"] = Qn
        out[f"{tag}_live_pid"] = np.asarray([c[0] for c in cap], np.int64)
        out[f"{tag}_live_score"] = np.asarray([c[1] for c in cap], np.float32)
        out[f"{tag}_live_bbox_int"] = np.asarray([c[2] for c in cap], np.int64)

        # --- counting path: CameraProcessor.process_frame
        fn2 = extract("peopleCount.py", "CameraProcessor", {"process_frame"})["process_frame"]
        rec, unk = [], []
        mgr = types.SimpleNamespace(
            process_detection=lambda pid, info, cam, ts, score: rec.append((info["pid"], score)),
            process_unknown_detection=lambda cam, ts, emb, bbox: unk.append((emb.copy(), bbox)))
        me2 = types.SimpleNamespace(
            face_detector=types.SimpleNamespace(get=lambda fr: faces),
            embedding_manager=types.SimpleNamespace(get_all=lambda: (embeddings, metadata)),
            manager=mgr, recognition_threshold=0.45, unknown_threshold=0.35,
            initialize_detector=lambda: None)
        stats = fn2(me2, np.zeros((4, 4, 3), np.uint8), "cam0")
        out[f"{tag}_count_stats"] = np.asarray([stats["faces"], stats["recognized"], stats["unknown"]], np.int64)
        out[f"{tag}_count_rec_pid"] = np.asarray([r_[0] for r_ in rec], np.int64)
        out[f"{tag}_count_rec_score"] = np.asarray([r_[1] for r_ in rec], np.float32)
        out[f"{tag}_count_unknown_emb"] = np.asarray([u[0] for u in unk], np.float32).reshape(-1, 512)
    np.savez_compressed(os.path.join(HERE, "match_kat.npz"), **out)


class _File:
    def __init__(self, b): self.b = b
    def read(self): return self.b


class _FS(dict):
    def get(self, k): return _File(self[k])


def make_gallery_row_and_enrol():
    rng = np.random.default_rng(11)
    base = unit(rng.standard_normal(512))
    poses = [unit(base + 0.045 * rng.standard_normal(512)) for _ in range(3)]
    # trainingServer.py:355,393 (inline statements, restated literally)
    avg_embedding = np.mean(poses, axis=0)
    embedding_bytes = pickle.dumps(avg_embedding)
    # infrenceServer.py:260-341 via the reference's own method
    load = extract("infrenceServer.py", "EmbeddingManager", {"_load_updated_embeddings"},
                   {"ObjectId": lambda x: x})["_load_updated_embeddings"]
    import threading
    me = types.SimpleNamespace(embeddings_lock=threading.Lock(), embeddings=OrderedDict(), employee_metadata={},
                               employee_embedding_fs=_FS(e1=embedding_bytes),
                               visitor_embedding_fs=_FS(v1=embedding_bytes))
    emp = [{"_id": "E1", "employeeEmbeddings": {"buffalo_l": {"embeddingId": "e1"}}, "employeeName": "A"}]
    vis = [{"_id": "V1", "visitorEmbeddings": {"buffalo_l": {"embeddingId": "v1", "status": "done"}}},
           {"_id": "V2", "visitorEmbeddings": {"buffalo_l": {"embeddingId": "v1", "status": "queued"}}}]
    load(me, emp, vis)
    assert list(me.embeddings) == ["E1", "V1"]
    out = {"poses": np.asarray(poses, np.float32), "avg": avg_embedding,
           "blob": np.frombuffer(embedding_bytes, np.uint8), "row": me.embeddings["E1"],
           "row_visitor": me.embeddings["V1"]}
    np.savez_compressed(os.path.join(HERE, "gallery_row_kat.npz"), **out)

    # ---- enrolment: trainingServer.py:170-247
    fns = extract("trainingServer.py", "FaceEmbeddingWorker",
                  {"_check_image_similarity", "_check_duplicate_face", "_process_image"},
                  {"ObjectId": object, "Collection": object, "GridFS": object,
                   "cv2": types.SimpleNamespace(IMREAD_COLOR=1, imdecode=lambda b, f: np.zeros((8, 8, 3), np.uint8))})
    cfg = types.SimpleNamespace(similarity_threshold=0.4, duplicate_threshold=0.4)
    me = types.SimpleNamespace(config=cfg)
    other = unit(rng.standard_normal(512))
    o = unit(rng.standard_normal(512)); o = unit(o - np.dot(o, base) * base)
    near_lo = unit(0.3999 * base + np.sqrt(1 - 0.3999 ** 2) * o)
    near_hi = unit(0.4001 * base + np.sqrt(1 - 0.4001 ** 2) * o)
    sets = {"same": poses, "one_off": [poses[0], poses[1], other], "single": [poses[0]],
            "edge_lo": [base, near_lo], "edge_hi": [base, near_hi]}
    out = {}
    for k, s in sets.items():
        ok, pair = fns["_check_image_similarity"](me, s)
        out[f"sim_{k}_in"] = np.asarray(s, np.float32)
        out[f"sim_{k}_ok"] = np.asarray([int(ok)], np.int64)
        out[f"sim_{k}_pair"] = np.asarray(pair if pair else (-1, -1), np.int64)
    # duplicate check against a stored company gallery (rows are pickled means, not unit)
    stored = [np.mean([unit(rng.standard_normal(512)) for _ in range(3)], axis=0) for _ in range(20)]
    stored[13] = np.mean([unit(base + 0.04 * rng.standard_normal(512)) for _ in range(3)], axis=0)
    fs = _FS({i: pickle.dumps(s) for i, s in enumerate(stored)})
    docs = [{"_id": i, "employee": f"EMP{i}", "employeeEmbeddings": {"buffalo_l": {"embeddingId": i}}}
            for i in range(20)]
    coll = types.SimpleNamespace(find=lambda q: iter(docs))
    fns["_check_duplicate_face"].__globals__["employee_embedding_fs"] = fs
    fns["_check_duplicate_face"].__globals__["visitor_embedding_fs"] = fs
    for k, v in (("dup", avg_embedding), ("nodup", np.mean([other, unit(rng.standard_normal(512))], axis=0))):
        is_dup, dup_id = fns["_check_duplicate_face"](me, v, "cid", coll, "employee")
        out[f"{k}_new"] = np.asarray(v, np.float32)
        out[f"{k}_is"] = np.asarray([int(is_dup)], np.int64)
        out[f"{k}_idx"] = np.asarray([int(dup_id[3:]) if dup_id else -1], np.int64)
    out["stored"] = np.asarray(stored, np.float32)
    # largest face rule through _process_image
    bbs = np.asarray([[0, 0, 10, 10], [5, 5, 40, 30], [1, 1, 36, 26], [2, 2, 37, 27]], np.float32)
    embs = [unit(rng.standard_normal(512)) for _ in bbs]
    me.face_detector = types.SimpleNamespace(get=lambda im: [FakeFace(b, e, 0.9) for b, e in zip(bbs, embs)])
    got = fns["_process_image"](me, "img", _FS(img=b"\x00"), "front")
    out["largest_bboxes"] = bbs
    out["largest_idx"] = np.asarray([[np.array_equal(got, e) for e in embs].index(True)], np.int64)
    np.savez_compressed(os.path.join(HERE, "enrol_kat.npz"), **out)

    # ---- UnknownPerson clustering: peopleCount.py:52-91
    UP = extract("peopleCount.py", "UnknownPerson", None)
    centres = [unit(rng.standard_normal(512)) for _ in range(3)]
    seq = [unit(centres[i % 3] + 0.03 * rng.standard_normal(512) * (1 + (i % 5 == 4))) for i in range(40)]
    clusters, assign, sims = [], [], []
    for e in seq:   # peopleCount.py:441-449 loop restated around the extracted class
        hit = None
        for k, c in enumerate(clusters):
            if c.compute_similarity(e) >= 0.65:
                hit = k
                break
        if hit is None:
            clusters.append(UP("u", "c", datetime(2025, 1, 1), "cam", e, [0, 0, 1, 1])); hit = len(clusters) - 1
        else:
            clusters[hit].update(datetime(2025, 1, 1), "cam", e, [0, 0, 1, 1])
        assign.append(hit)
    np.savez_compressed(os.path.join(HERE, "unknown_kat.npz"), seq=np.asarray(seq, np.float32),
                        assign=np.asarray(assign, np.int64),
                        final_avg=np.asarray([c.avg_embedding for c in clusters], np.float32),
                        counts=np.asarray([c.detection_count for c in clusters], np.int64))


def synth_frame(h, w, seed):
    """Structured synthetic BGR frame (low-pass noise + blobs), uint8."""
    rng = np.random.default_rng(seed)
    import torch
    import torch.nn.functional as F
    low = torch.from_numpy(rng.random((1, 3, max(h // 16, 2), max(w // 16, 2)), dtype=np.float32))
    mid = torch.from_numpy(rng.random((1, 3, max(h // 4, 2), max(w // 4, 2)), dtype=np.float32))
    img = 0.6 * F.interpolate(low, (h, w), mode="bicubic", align_corners=False) + \
        0.3 * F.interpolate(mid, (h, w), mode="bilinear", align_corners=False) + \
        0.1 * torch.from_numpy(rng.random((1, 3, h, w), dtype=np.float32))
    return np.ascontiguousarray((img[0].permute(1, 2, 0).clamp(0, 1) * 255).round().to(torch.uint8).numpy())


def make_nets():
    import torch
    from facerecognition_infrenceengine_amd import weights
    from oracle import nets, detect, align
    torch.set_num_threads(8)
    # ---- align
    rng = np.random.default_rng(3)
    frame = synth_frame(240, 320, 5)
    kps = np.stack([align.ARCFACE_DST * s + t + rng.normal(0, 1.5, (5, 2))
                    for s, t in ((1.1, (60, 40)), (0.7, (150, 90)), (1.6, (-20, 30)))]).astype(np.float32)
    crops, Ms = zip(*[align.norm_crop(frame, k) for k in kps])
    np.savez_compressed(os.path.join(HERE, "align_kat.npz"), frame=frame, kps=kps,
                        M=np.asarray(Ms, np.float64), crops=np.asarray(crops, np.uint8))
    # ---- r100 (weights regenerated from the seed, not stored)
    st = weights.synth_iresnet_state("r100", seed=1234)
    x = torch.from_numpy(np.stack([align.crop_to_net(c) for c in crops[:2]]))
    taps = {}
    emb = nets.iresnet_forward(st, x, weights.IRESNET_LAYERS["r100"], taps)
    sub = {k.replace(".", "_"): v[:, ::8, ::3, ::3].numpy() for k, v in taps.items()}
    np.savez_compressed(os.path.join(HERE, "r100_kat.npz"), seed=np.asarray([1234]), x=x.numpy(),
                        embedding=emb.numpy(), **{"tap_" + k: v for k, v in sub.items()})
    # ---- mtcnn on one 320x240 frame
    p, r, o = weights.synth_mtcnn_states(seed=4321)
    fr = synth_frame(240, 320, 9)
    tr = {}
    b, s, k = detect.detect(fr, p, r, o, trace=tr)
    save = {"frame": fr, "bbox": b, "score": s, "kps": k, "seed": np.asarray([4321])}
    for i, (pp, rr) in enumerate(zip(tr["pnet_prob"], tr["pnet_reg"])):
        save[f"pnet_prob_{i}"] = pp; save[f"pnet_reg_{i}"] = rr
    for key in ("stage1_boxes", "stage1_scores", "rnet_score", "rnet_reg", "stage2_boxes", "stage2_scores",
                "onet_score", "onet_reg", "onet_lm"):
        if key in tr:
            save[key] = tr[key]
    np.savez_compressed(os.path.join(HERE, "mtcnn_kat.npz"), **save)
    print("mtcnn faces:", len(s), {k_: v.shape for k_, v in save.items() if k_.startswith("stage")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["match", "enrol", "nets"]
    if "match" in which:
        make_match()
    if "enrol" in which:
        make_gallery_row_and_enrol()
    if "nets" in which:
        make_nets()
    print("golden vectors written to", HERE)
