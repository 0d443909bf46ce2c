"""GPU end-to-end parity through the reference-shaped API: FaceAnalysis.get + gallery match vs the
full CPU oracle (detect -> align -> embed -> match) on the same inputs.
north_star bar: embeddings within 1e-3 cosine, identical top-1 ids."""
import os
import sys
import warnings
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import align as oalign, detect as odetect, match as omatch, nets as onets

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


@pytest.fixture(scope="module")
def app():
    from facerecognition_infrenceengine_amd import FaceAnalysis
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = FaceAnalysis(name="buffalo_l", providers=["CUDAExecutionProvider", "CPUExecutionProvider"])
        a.prepare(ctx_id=0)
    assert a.synthetic
    return a


def oracle_pipeline(frame):
    from facerecognition_infrenceengine_amd import weights
    p, r, o = weights.synth_mtcnn_states()
    st = weights.synth_iresnet_state("r100")
    b, s, k = odetect.detect(frame, p, r, o)
    if len(s) == 0:
        return b, s, k, np.zeros((0, 512), np.float32)
    crops = [oalign.norm_crop(frame, kk)[0] for kk in k]
    x = torch.from_numpy(np.stack([oalign.crop_to_net(c) for c in crops]))
    emb = onets.iresnet_forward(st, x, weights.IRESNET_LAYERS["r100"]).numpy()
    return b, s, k, emb


@pytest.mark.parametrize("hw,seed", [((480, 640), 0), ((240, 320), 4)])
def test_get_matches_oracle_pipeline(app, hw, seed):
    """C1 of BASELINE.json: single 640x480 frame, detect+embed+match vs a 100-row gallery."""
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import GalleryMatcher
    frame = synth_frame(hw[0], hw[1], seed)
    ob, os_, ok, oemb = oracle_pipeline(frame)
    faces = app.get(frame)
    assert len(faces) == len(os_) and len(faces) >= 1
    emb = np.stack([f.embedding for f in faces]); normed = np.stack([f.normed_embedding for f in faces])
    np.testing.assert_allclose(np.stack([f.bbox for f in faces]), ob, atol=5e-3)
    np.testing.assert_allclose(np.stack([f.kps for f in faces]), ok, atol=5e-3)
    np.testing.assert_allclose([f.det_score for f in faces], os_, atol=5e-5)
    cos = (emb * oemb).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(oemb, axis=1))
    assert (1 - cos).max() < 1e-3, cos
    assert faces[0].bbox.dtype == np.float32 and faces[0].kps.shape == (5, 2) and normed.dtype == np.float32
    # 100-row gallery with rows planted from the ORACLE embeddings (+ noise), rest random
    rng = np.random.default_rng(1)
    G = rng.standard_normal((100, 512)).astype(np.float32)
    on = oemb / np.linalg.norm(oemb, axis=1, keepdims=True)
    rows = rng.permutation(100)[:len(on)]
    for f, r in enumerate(rows[: max(1, len(on) - 1)]):           # leave the last face unplanted (-> unknown)
        G[r] = on[f] + 0.02 * rng.standard_normal(512)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    gal = OrderedDict((i, G[i]) for i in range(100))
    want = []
    for f in range(len(on)):                                      # literal reference loop on the oracle rows
        bid, bs = omatch.linear_scan(omatch.renormalise(on[f].astype(np.float32)), gal)
        want.append(bid if bs >= 0.4 else -1)
    m = GalleryMatcher("cuda:0")
    m.set_rows(list(range(100)), G, normalise=False)
    ids, score, idx = m.match(normed, thr=0.4)
    got = [-1 if i is None else i for i in ids]
    assert got == want                                            # identical top-1 ids / unknown decisions


def test_processor_end_to_end(app):
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd.processor import (CameraProcessor, EmbeddingManager,
                                                              FaceRecognitionProcessor, InMemoryStore)
    frame = synth_frame(240, 320, 4)
    faces = app.get(frame)
    store = InMemoryStore()
    rng = np.random.default_rng(3)
    for i in range(30):
        store.add_employee(f"e{i}", "acme", rng.standard_normal(512), name=f"E{i}")
    # enrol face 0 as a (mean-of-poses, un-normalised) row, as trainingServer.py:355 stores it
    poses = [faces[0].normed_embedding + 0.01 * rng.standard_normal(512).astype(np.float32) for _ in range(3)]
    store.add_employee("target", "acme", np.mean(poses, axis=0), name="Target")
    store.add_employee("other_co", "globex", faces[0].normed_embedding, name="Elsewhere")
    mgr = EmbeddingManager(store=store)
    proc = FaceRecognitionProcessor(mgr, face_detector=app)
    res = proc.recognize(frame, "acme")
    assert len(res) == len(faces)
    assert res[0]["person_id"] == "target" and res[0]["recognition_score"] > 0.9
    assert all(r["person_id"] in (None, "target") for r in res)
    assert np.array_equal(res[0]["bbox"], faces[0].bbox.astype(int))
    out = proc.recognize_faces(frame.copy(), "acme")
    assert out.shape == frame.shape and (out != frame).any()        # boxes drawn
    assert proc.recognize(frame, "nobody") is None                   # empty gallery short-circuit (:523-525)
    bad = proc.recognize_faces(np.zeros((4, 4), np.uint8), "acme")   # errors are swallowed (:560-563)
    assert bad.shape == (4, 4)

    class Mgr:
        def __init__(self): self.rec, self.unk = [], []
        def process_detection(self, pid, info, cam, ts, score): self.rec.append((pid, score))
        def process_unknown_detection(self, cam, ts, emb, bbox): self.unk.append((emb, bbox))
    cm = Mgr()
    stats = CameraProcessor(mgr, cm, face_detector=app).process_frame(frame, "cam0")
    assert stats["faces"] == len(faces) and stats["recognized"] >= 1
    assert cm.rec[0][0] in ("target", "other_co") and stats["recognized"] + stats["unknown"] <= stats["faces"]
    for emb, bbox in cm.unk:
        assert abs(np.linalg.norm(emb) - 1) < 1e-5 and len(bbox) == 4


def test_get_is_thread_safe(app):
    """One engine shared by 3 threads, as the enrolment worker does (trainingServer.py:115,227)."""
    import threading
    from make_golden import synth_frame
    frames = [synth_frame(120, 160, s) for s in (1, 2, 3)]
    ref = [app.get(f) for f in frames]
    out = [None] * 3

    def work(i):
        out[i] = app.get(frames[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for a, b in zip(ref, out):
        assert len(a) == len(b)
        for fa, fb in zip(a, b):
            assert np.array_equal(fa.embedding, fb.embedding)


def test_slot_path_equals_compact_path(app):
    """The sync-free fixed-slot pipeline (bench / streaming) gives the same faces as the compact one."""
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (4, 5, 6)]))
    dev = torch.from_numpy(frs).cuda()
    a = app.detect_embed_device(dev)
    b = app.detect_embed_slots(dev)
    cnt = b["counts"].cpu().tolist()
    assert cnt == a["counts"]
    cap = b["bbox"].shape[1]
    rows = torch.tensor([f * cap + j for f, n in enumerate(cnt) for j in range(n)], device="cuda")
    from facerecognition_infrenceengine_amd.iresnet import LOW_BATCH, SMALL_BATCH
    mode = lambda B: 0 if B <= LOW_BATCH else (1 if B <= SMALL_BATCH else 2)
    if mode(sum(cnt)) == mode(len(cnt) * cap):          # same split-K mode of the embed network: the same bits
        assert torch.equal(b["embedding"][rows], a["embedding"])
    else:
        cos = torch.nn.functional.cosine_similarity(b["embedding"][rows], a["embedding"])
        assert float((1 - cos).max()) < 1e-5
    assert torch.equal(b["bbox"].reshape(-1, 4)[rows], a["bbox"])


def test_two_stream_overlap_equals_single_stream(app):
    """Detector on its own HIP stream (overlapping the previous batch's embed): same results, batch after batch."""
    from make_golden import synth_frame
    batches = [torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s + k) for s in (10, 20)]))).cuda()
               for k in range(3)]
    want = [app.detect_embed_slots(b) for b in batches]
    torch.cuda.synchronize()
    s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    with torch.cuda.stream(s_emb):
        for _ in range(2):                      # second round reuses freed blocks across the streams
            got = [app.detect_embed_slots(b, det_stream=s_det) for b in batches]
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert torch.equal(w["counts"], g["counts"])
        cap = w["bbox"].shape[1]
        valid = (torch.arange(cap, device="cuda")[None, :] < w["counts"][:, None]).reshape(-1)   # empty slots: undefined
        assert valid.any()
        for k in ("bbox", "kps", "det_score", "embedding", "normed_embedding"):
            a, b = w[k].reshape(valid.numel(), -1)[valid], g[k].reshape(valid.numel(), -1)[valid]
            assert torch.equal(a, b), k


def test_detector_is_bit_stable_beside_the_embedder(app):
    """Regression for a cross-stream hazard measured on MI355X: with packed-f32 VALU ops in the detector's kernels,
    lanes 48-63 of their results went stale whenever the embedder's conv kernels shared the SIMDs;
    the library is built without those ops.  Detector outputs must be bit-identical to a solo run while the embed
    convs run on a second stream."""
    from make_golden import synth_frame
    batches = [torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s + k) for s in (10, 20)]))).cuda()
               for k in range(3)]
    from facerecognition_infrenceengine_amd.distributed import HipOps
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    crops = (torch.rand((64, 112, 112, 8), device="cuda") * 2 - 1).half()
    crops[..., 3:] = 0
    want = [app.det.detect_batch(b) for b in batches]
    emb0 = app.rec.forward(crops)[0].clone()
    # the sharded match's exchange runs on the embed stream too: its glue and arithmetic are libfrhip.so kernels
    # (no torch arithmetic beside the convs); one rank's worth of it here, the two all-gathers being byte moves
    gm = GalleryMatcher("cuda:0")
    gm.set_rows(range(4096), torch.randn((4096, 512), device="cuda"), normalise=True)
    ops = HipOps(gm, 0)

    def exchange(e):
        Qn = ops.renormalise(e)
        allq = ops.pack_queries(Qn, 64)
        cnt = ops.gathered_counts(allq, 1, 64)
        idx, score = ops.scan(allq, counts=cnt, seg_len=65)
        return ops.reduce(ops.pack(idx, score), 1, 65, 0, 64)
    idx0, score0 = exchange(emb0)
    idx0, score0 = idx0.clone(), score0.clone()
    torch.cuda.synchronize()
    s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(8):
        got, embs, ids = [], [], []
        for b in batches:
            with torch.cuda.stream(s_emb):
                embs.append(app.rec.forward(crops)[0])
                ids.append(exchange(embs[-1]))
            with torch.cuda.stream(s_det):
                got.append(app.det.detect_batch(b))
        torch.cuda.synchronize()
        for e in embs:
            assert torch.equal(e, emb0)
        for i, sc in ids:
            assert torch.equal(i, idx0) and torch.equal(sc, score0)
        for w, g in zip(want, got):
            assert torch.equal(w[3], g[3])
            cap = w[0].shape[1]
            valid = (torch.arange(cap, device="cuda")[None, :] < w[3][:, None]).reshape(-1)
            for a, b in zip(w[:3], g[:3]):
                assert torch.equal(a.reshape(valid.numel(), -1)[valid], b.reshape(valid.numel(), -1)[valid]), rep


def test_frame_ingest_ring_feeds_the_pipeline(app):
    """Pinned ring -> copy stream -> detector waits on the upload event only."""
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd.ingest import FrameIngest
    ing = FrameIngest(2, 240, 320, "cuda:0", depth=2)
    sets = [np.ascontiguousarray(np.stack([synth_frame(240, 320, s + k) for s in (10, 20)])) for k in range(4)]
    want = [app.detect_embed_slots(torch.from_numpy(f).cuda()) for f in sets]
    torch.cuda.synchronize()
    s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    with torch.cuda.stream(s_emb):
        for k, f in enumerate(sets):                        # 4 batches through a 2-slot ring: slots are reused
            if k >= ing.depth:
                got[k - ing.depth]["counts"].cpu()          # the caller owns host-buffer reuse: batch k-2 is done
            ing.host_buffer(k)[...] = f
            frames, ready = ing.upload(k)
            got.append(app.detect_embed_slots(frames, det_stream=s_det, ready_event=ready))
            ing.release(k)
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        assert torch.equal(w["counts"], g["counts"]) and int(w["counts"].sum()) > 0
        cap = w["bbox"].shape[1]
        valid = (torch.arange(cap, device="cuda")[None, :] < w["counts"][:, None]).reshape(-1)
        for k in ("bbox", "det_score", "embedding"):
            assert torch.equal(w[k].reshape(valid.numel(), -1)[valid], g[k].reshape(valid.numel(), -1)[valid]), k


def test_slot_path_with_compact_embed(app):
    """detect_embed_slots(compact_embed=True) embeds only the slots that hold a face (one sync on the counts) and keeps
    the slot layout: detector outputs identical, embeddings of the valid slots equal to the all-slots run (same bits in
    the same split-K mode, f16 rounding noise otherwise), empty slots finite."""
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd.iresnet import LOW_BATCH, SMALL_BATCH
    frs = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (4, 5, 6, 7)]))).cuda()
    a = app.detect_embed_slots(frs)
    b = app.detect_embed_slots(frs, compact_embed=True)
    assert torch.equal(a["counts"], b["counts"])
    cap = a["bbox"].shape[1]
    cnt = a["counts"].cpu()
    valid = (torch.arange(cap)[None, :] < cnt[:, None]).reshape(-1).cuda()
    for k in ("bbox", "kps", "det_score"):               # empty slots hold whatever the allocator left there
        assert torch.equal(a[k].reshape(valid.numel(), -1)[valid], b[k].reshape(valid.numel(), -1)[valid]), k
    assert int(valid.sum()) >= 2 and bool(torch.isfinite(b["normed_embedding"]).all())
    mode = lambda B: 0 if B <= LOW_BATCH else (1 if B <= SMALL_BATCH else 2)
    if mode(int(valid.sum())) == mode(valid.numel()):
        assert torch.equal(a["embedding"][valid], b["embedding"][valid])
    else:
        cos = torch.nn.functional.cosine_similarity(a["normed_embedding"][valid], b["normed_embedding"][valid])
        assert float((1 - cos).max()) < 1e-5


def test_frame_ingest_decodes_into_the_pinned_ring():
    """FrameIngest.decode_into: encoded bytes (the enrolment path's GridFS blobs, trainingServer.py:219-221) land in the
    slot's pinned buffer and reach the device unchanged; undecodable bytes and a wrong picture size are refused."""
    import io
    from PIL import Image
    from facerecognition_infrenceengine_amd.ingest import FrameIngest
    rng = np.random.default_rng(4)
    imgs = [rng.integers(0, 256, (60, 80, 3), dtype=np.uint8) for _ in range(2)]
    ing = FrameIngest(2, 60, 80, "cuda:0", depth=2)
    for i, im in enumerate(imgs):
        buf = io.BytesIO()
        Image.fromarray(im[:, :, ::-1]).save(buf, format="PNG")           # PNG holds RGB; frames are BGR
        assert ing.decode_into(0, i, buf.getvalue())
    assert not ing.decode_into(0, 0, b"garbage")
    small = io.BytesIO()
    Image.fromarray(imgs[0][:30, :40, ::-1].copy()).save(small, format="PNG")
    assert not ing.decode_into(0, 0, small.getvalue())
    dev, ready = ing.upload(0)
    ready.synchronize()
    assert np.array_equal(dev.cpu().numpy(), np.stack(imgs))


def test_graph_replay_equals_eager(app):
    """enable_graphs(): the captured slot pipeline gives the same Face lists as the eager path, call after call.
    The embed network sums its small-batch split-K slices in a batch-size MODE (<= 8 faces, <= 48, more: iresnet.py):
    with cap_o = 4 the eager call (its faces) and the captured one (4 slots) run in the same mode and must agree bit
    for bit; with the default 16 slots the modes differ and the embeddings agree to f16 rounding noise."""
    from make_golden import synth_frame
    frames = [synth_frame(240, 320, s) for s in (4, 5, 6)]
    for eng, exact in ((app.clone_with(cap_o=4), True), (app, False)):
        eager = [eng.get(f) for f in frames]
        try:
            eng.enable_graphs(True)
            for _ in range(2):
                for f, want in zip(frames, eager):
                    got = eng.get(f)
                    assert len(got) == len(want) and len(got) >= 1
                    for a, b in zip(want, got):
                        assert np.array_equal(a.bbox, b.bbox) and np.array_equal(a.kps, b.kps)
                        assert a.det_score == b.det_score
                        if exact:
                            assert np.array_equal(a.embedding, b.embedding)
                            assert np.array_equal(a.normed_embedding, b.normed_embedding)
                        else:
                            assert 1.0 - float(a.normed_embedding @ b.normed_embedding) < 1e-5
        finally:
            eng.enable_graphs(False)


def test_graph_replay_of_a_single_frame_embeds_only_the_slots_it_needs(app):
    """A single frame on an engine with several face slots is captured in two parts (detector | align + embed for 1, 2, 4, 8 ...
    slots, chosen by the face count): a frame with a few faces takes a small embed graph - the same batch-size mode as the
    eager call - and the Face lists agree bit for bit; frames with other counts reuse or add graphs."""
    from make_golden import synth_frame
    eng = app.clone_with(cap_o=16, thresholds=(0.6, 0.7, 0.78))       # a stricter O-Net threshold: 2 - 5 faces per frame, 16 slots
    frames = [synth_frame(240, 320, s) for s in (4, 5, 6, 7, 8)]
    eager = [eng.get(f) for f in frames]
    counts = sorted(len(e) for e in eager)
    assert 1 <= counts[0] and counts[-1] <= 8 and len(set(counts)) >= 3, counts
    try:
        eng.enable_graphs(True)
        for _ in range(2):
            for f, want in zip(frames, eager):
                got = eng.get(f)
                assert len(got) == len(want)
                for a, b in zip(want, got):
                    assert np.array_equal(a.bbox, b.bbox) and np.array_equal(a.kps, b.kps) and a.det_score == b.det_score
                    assert np.array_equal(a.embedding, b.embedding) and np.array_equal(a.normed_embedding, b.normed_embedding)
        pipe = eng._graphs[(1, 240, 320, 3)]
        assert pipe.split and 1 <= len(pipe.embed) <= 4 and max(pipe.embed) <= 8
    finally:
        eng.enable_graphs(False)


def test_model_pack_on_disk_round_trips_through_load_state(app, tmp_path):
    """The drop-in loads `<root>/models/<name>/arcface_<arch>.{safetensors,pt}` + `mtcnn_{pnet,rnet,onet}.pt`
    (weights.load_state).  A pack written from the synthetic state dicts must give the synthetic engine's results
    bit for bit, and the engine must report that it is NOT running on fallback weights."""
    from safetensors.torch import save_file
    from facerecognition_infrenceengine_amd import FaceAnalysis, weights
    from make_golden import synth_frame
    d = tmp_path / "models" / "mypack"
    d.mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in weights.synth_iresnet_state("r100").items()}, str(d / "arcface_r100.safetensors"))
    for n, st in zip(("pnet", "rnet", "onet"), weights.synth_mtcnn_states()):
        torch.save({"state_dict": st}, str(d / f"mtcnn_{n}.pt"))              # wrapped form is unwrapped by load_state
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                         # the synthetic-fallback warning must NOT fire
        b = FaceAnalysis(name="mypack", root=str(tmp_path)).prepare(ctx_id=0)
    assert b.synthetic is False
    frame = synth_frame(240, 320, 4)
    fa, fb = app.get(frame), b.get(frame)
    assert len(fa) == len(fb) >= 1
    for x, y in zip(fa, fb):
        assert np.array_equal(x.embedding, y.embedding) and np.array_equal(x.bbox, y.bbox)


def test_model_pack_with_an_onnx_recognition_network(tmp_path):
    """The reference's pack ships its ArcFace network as ONNX (buffalo_l/w600k_r50.onnx, infrenceServer.py:412-416).  A
    pack holding an exporter-style r50 graph (BatchNorms folded into the convs) next to a file that is not an IResNet
    must come up as r50 with that network: embeddings of the same crops equal those of an engine built from the
    original state dict within f16 folding noise, and match the fp32 oracle."""
    from facerecognition_infrenceengine_amd import FaceAnalysis, weights
    from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
    from oracle import nets
    from tests.helpers import onnx_write as ow
    d = tmp_path / "models" / "buffalo_like"
    d.mkdir(parents=True)
    st = weights.synth_iresnet_state("r50", seed=11)
    ow.write_iresnet_onnx(d / "w600k_r50.onnx", {k: v.numpy() for k, v in st.items()}, "r50", fold_bn=True)
    (d / "det_10g.onnx").write_bytes(ow._vi(1, 7) + ow._ld(7, ow._ld(11, ow._ld(1, b"x"))))     # parses, is no IResNet
    for n, s in zip(("pnet", "rnet", "onet"), weights.synth_mtcnn_states()):
        torch.save(s, str(d / f"mtcnn_{n}.pt"))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        b = FaceAnalysis(name="buffalo_like", root=str(tmp_path)).prepare(ctx_id=0)
    assert b.arch == "r50" and b.synthetic is False
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 3, 112, 112, generator=g).clamp(-1, 1)
    xp = torch.zeros(4, 112, 112, 8, dtype=torch.float16)
    xp[..., :3] = x.permute(0, 2, 3, 1).to(torch.float16)
    xd = xp.cuda()
    emb_pack, _ = b.rec.forward(xd)
    emb_orig, _ = IResNetHIP(st, "r50", "cuda:0").forward(xd)
    want = nets.iresnet_forward(st, xp[..., :3].float().permute(0, 3, 1, 2), nets.IRESNET_LAYERS["r50"])
    cos = torch.nn.functional.cosine_similarity
    assert (1 - cos(emb_pack.cpu(), emb_orig.cpu())).max().item() < 1e-4
    assert (1 - cos(emb_pack.cpu(), want)).max().item() < 1e-3


def test_processor_threads_share_one_engine_while_the_gallery_syncs(app):
    """FaceRecognitionProcessor.recognize from 3 threads on ONE engine while another thread keeps force_sync()ing a
    store whose membership changes: detection runs under the engine lock, a stale company view is re-fetched and
    retried, so every call returns results (the reference swallows exceptions and would return an unmarked frame)."""
    import threading
    from facerecognition_infrenceengine_amd.processor import EmbeddingManager, FaceRecognitionProcessor, InMemoryStore
    from make_golden import synth_frame
    rng = np.random.default_rng(3)
    store = InMemoryStore()
    for i in range(50):
        store.add_employee(f"e{i}", "acme", rng.standard_normal(512).astype(np.float32), name=f"E{i}")
    mgr = EmbeddingManager(store=store, device="cuda:0")
    proc = FaceRecognitionProcessor(mgr, face_detector=app)
    frames = [synth_frame(240, 320, s) for s in (4, 5, 6)]
    ref = [proc.recognize(f, "acme") for f in frames]
    assert all(r is not None for r in ref) and sum(len(r) for r in ref) >= 1
    stop, errors, done = threading.Event(), [], [0, 0, 0]

    def churn():
        k = 0
        while not stop.is_set():
            store.add_employee(f"x{k}", "acme", rng.standard_normal(512).astype(np.float32))   # membership change
            mgr.force_sync()
            k += 1

    def work(i):
        try:
            for _ in range(12):
                r = proc.recognize(frames[i], "acme")
                assert r is not None and len(r) == len(ref[i])
                assert all(np.array_equal(a["bbox"], b["bbox"]) for a, b in zip(r, ref[i]))
                done[i] += 1
        except Exception as e:                                  # noqa: BLE001 - reported below
            errors.append(repr(e))
    tc = threading.Thread(target=churn)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    tc.start(); [t.start() for t in ts]; [t.join() for t in ts]
    stop.set(); tc.join()
    assert not errors and done == [12, 12, 12], (errors, done)


def test_c2_geometry_1080p_frame_vs_oracle(app):
    """BASELINE config C2's frame geometry (1080 x 1920: 12 pyramid levels) end to end against the CPU oracle:
    detect -> align -> embed -> match of ONE full-HD frame through the reference-shaped get()."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
    from make_golden import synth_frame
    assert len(pyramid_scales(1080, 1920)) == 12
    frame = synth_frame(1080, 1920, 31)
    ob, os_, ok, oemb = oracle_pipeline(frame)
    faces = app.get(frame)
    assert len(faces) == len(os_) >= 1
    np.testing.assert_allclose(np.stack([f.bbox for f in faces]), ob, atol=1e-2)
    np.testing.assert_allclose(np.array([f.det_score for f in faces]), os_, atol=5e-5)
    np.testing.assert_allclose(np.stack([f.kps for f in faces]), ok, atol=1e-2)
    emb = np.stack([f.embedding for f in faces])
    cos = (emb * oemb).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(oemb, axis=1))
    assert (1 - cos).max() < 1e-3, cos
    rng = np.random.default_rng(8)
    G = rng.standard_normal((10_000, 512)).astype(np.float32)
    rows = rng.choice(10_000, len(faces), replace=False)
    G[rows] = oemb / np.linalg.norm(oemb, axis=1, keepdims=True) + 0.02 * rng.standard_normal(oemb.shape).astype(np.float32)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    m = GalleryMatcher("cuda:0")
    m.set_rows(list(range(10_000)), G, normalise=False)
    ids, score, idx = m.match(np.stack([f.normed_embedding for f in faces]))
    oi, _ = omatch.match_rows_fast(np.stack([omatch.renormalise(e / np.linalg.norm(e)) for e in oemb]), G)
    assert np.array_equal(idx, oi) and np.array_equal(idx, rows)          # identical top-1 ids, the planted rows


def test_eight_frame_batch_slot_path_vs_oracle(app):
    """Eight small frames through the sync-free slot path, every frame against the oracle (16.6 Mpixel and below run the all-f32
    detector on level streams; the batch arithmetic is under the oracle in the *_batch_path_* tests at the end of this file)."""
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(360, 640, 40 + i) for i in range(8)]))
    r = app.detect_embed_slots(torch.from_numpy(frs).cuda())
    counts = r["counts"].cpu().numpy()
    cap = r["bbox"].shape[1]
    emb = r["embedding"].cpu().numpy().reshape(8, cap, 512)
    total = 0
    for i in range(8):
        ob, os_, ok, oemb = oracle_pipeline(frs[i])
        n = min(len(os_), cap)
        assert counts[i] == n
        total += n
        if n:
            np.testing.assert_allclose(r["bbox"][i, :n].cpu().numpy(), ob[:n], atol=5e-3)
            np.testing.assert_allclose(r["det_score"][i, :n].cpu().numpy(), os_[:n], atol=5e-5)
            e = emb[i, :n]
            cos = (e * oemb[:n]).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(oemb[:n], axis=1))
            assert (1 - cos).max() < 1e-3, (i, cos)
    assert total >= 4


def test_bench_geometry_1080p_batch_two_streams_vs_oracle(app):
    """What bench.py times, at its own geometry: a batch of full-HD frames through the sync-free slot path with the
    detector on its own HIP stream beside the previous call's embed convs (det_stream=...), every frame against the
    CPU oracle: boxes, scores, landmarks, embeddings, and top-1 ids against a gallery with planted rows."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 60 + i) for i in range(4)]))
    dev = torch.from_numpy(frs).cuda()
    s_det, s_emb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s_emb):
        app.detect_embed_slots(dev, det_stream=s_det)            # a previous step in flight: its convs overlap ...
        r = app.detect_embed_slots(dev, det_stream=s_det)        # ... this step's detector
    torch.cuda.synchronize()
    counts = r["counts"].cpu().numpy()
    cap = r["bbox"].shape[1]
    emb = r["embedding"].cpu().numpy().reshape(4, cap, 512)
    nrm = r["normed_embedding"].cpu().numpy().reshape(4, cap, 512)
    rng = np.random.default_rng(9)
    G = rng.standard_normal((10_000, 512)).astype(np.float32)
    oracle_emb, queries, planted = [], [], []
    for i in range(4):
        ob, os_, ok, oemb = oracle_pipeline(frs[i])
        n = min(len(os_), cap)
        assert counts[i] == n >= 1
        np.testing.assert_allclose(r["bbox"][i, :n].cpu().numpy(), ob[:n], atol=1e-2)
        np.testing.assert_allclose(r["det_score"][i, :n].cpu().numpy(), os_[:n], atol=5e-5)
        np.testing.assert_allclose(r["kps"][i, :n].cpu().numpy(), ok[:n], atol=1e-2)
        e = emb[i, :n]
        cos = (e * oemb[:n]).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(oemb[:n], axis=1))
        assert (1 - cos).max() < 1e-3, (i, cos)
        for j in range(n):
            row = int(rng.integers(0, 10_000))
            while row in planted:
                row = int(rng.integers(0, 10_000))
            planted.append(row)
            G[row] = oemb[j] / np.linalg.norm(oemb[j]) + 0.02 * rng.standard_normal(512).astype(np.float32)
            oracle_emb.append(oemb[j]); queries.append(nrm[i, j])
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    m = GalleryMatcher("cuda:0")
    m.set_rows(list(range(10_000)), G, normalise=False)
    ids, score, idx = m.match(np.stack(queries))
    oi, _ = omatch.match_rows_fast(np.stack([omatch.renormalise(e / np.linalg.norm(e)) for e in oracle_emb]), G)
    assert np.array_equal(idx, oi)                               # identical top-1 ids vs the oracle's own embeddings


def test_camera_batcher_equals_per_frame_recognition(app):
    """N camera queues drained into ONE batch per turn must give, per source, exactly what the
    reference's per-frame caller gets (recognize(frame) one frame at a time, infrenceServer.py:603-622): same faces,
    same truncated boxes, same person ids, and per-company gallery views."""
    import queue
    from facerecognition_infrenceengine_amd.camera import CameraManager
    from facerecognition_infrenceengine_amd.processor import EmbeddingManager, FaceRecognitionProcessor, InMemoryStore
    from make_golden import synth_frame
    frames = [synth_frame(240, 320, s) for s in (4, 5, 6, 21)]
    # gallery: the faces of frames 0 and 2 are enrolled (company acme), plus noise rows and another company's rows
    rng = np.random.default_rng(9)
    store = InMemoryStore()
    for i in range(30):
        store.add_employee(f"n{i}", "acme" if i % 2 else "other", rng.standard_normal(512).astype(np.float32), name=f"N{i}")
    for k in (0, 2):
        for j, f in enumerate(app.get(frames[k])):
            store.add_employee(f"face{k}_{j}", "acme", f.embedding, name=f"F{k}{j}")
    mgr = EmbeddingManager(store=store, device="cuda:0")
    proc = FaceRecognitionProcessor(mgr, face_detector=app)
    want = [proc.recognize(f, "acme") for f in frames]
    assert sum(r["person_id"] is not None for res in want for r in res) >= 1
    cm = CameraManager(mgr, processor=proc)
    cm.frame_queues = {s: queue.Queue(maxsize=2) for s in range(4)}
    cm.result_queue = queue.Queue(maxsize=10)
    cm.running = True
    for s in range(4):
        cm.frame_queues[s].put(frames[s].copy())
    batch = cm.take_batch([0, 1, 2, 3])
    assert len(batch) == 4
    got = cm.process_batch(batch, "acme")
    cm.running = False
    assert cm.stats["largest_batch"] == 4
    for a, b in zip(want, got):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x["bbox"], y["bbox"]) and x["person_id"] == y["person_id"]
            assert x["det_score"] == y["det_score"]
            # the embed net picks its kernels by batch-size mode (<= 8, <= 48 faces, more): the per-frame calls and the
            # batch (its faces together) may run in different modes; scores then agree to f16-conv rounding
            assert abs(float(x["recognition_score"]) - float(y["recognition_score"])) < 5e-4
    outs = [cm.result_queue.get_nowait() for _ in range(4)]
    assert [s for s, _ in outs] == [0, 1, 2, 3] and all(o.shape == (240, 320, 3) for _, o in outs)
    assert proc.recognize_batch(frames[:2], "nobody") is None          # unknown company: no gallery, frames untouched


def _oracle_frame(frame, cap_o=16, embed=True):
    """oracle_pipeline with the engine's face-slot count (the oracle's ``cap_o`` is the detector's last capacity)"""
    from facerecognition_infrenceengine_amd import weights
    p, r, o = weights.synth_mtcnn_states()
    st = weights.synth_iresnet_state("r100")
    b, s, k = odetect.detect(frame, p, r, o, cap_o=cap_o)
    if not embed:
        return b, s, k, None
    crops = [oalign.norm_crop(frame, kk)[0] for kk in k]
    x = torch.from_numpy(np.stack([oalign.crop_to_net(c) for c in crops]))
    return b, s, k, onets.iresnet_forward(st, x, weights.IRESNET_LAYERS["r100"]).numpy()


def test_c5_pipeline_calibrate_fp8_embed_fp8_scan_vs_oracle():
    """BASELINE config C5 through the PRODUCT entry (/root/reference/infrenceServer.py:528-552):
    1080p frames -> FaceAnalysis.calibrate_fp8 (fp8 body convs, calibrated on the faces the engine itself detects) ->
    detect_embed_slots -> GalleryMatcher(scan="f8") (fp8 coarse scan + exact f32 re-rank), every frame against the CPU
    oracle: counts / boxes / scores / landmarks as on the f16 path (the detector does not change), embeddings within
    north_star's 1 - cos < 1e-3, top-1 ids and the 0.4 decisions equal to the literal loop on the ORACLE's embeddings.
    Gallery: 20 000 rows with one planted row per face (oracle embedding + N(0, 0.02)); four extra queries - random unit
    rows - must come back "unknown" through the same fp8 scan (their best score is far below 0.4)."""
    from facerecognition_infrenceengine_amd import FaceAnalysis
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    from make_golden import synth_frame
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a8 = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
    calib = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 90 + i) for i in range(4)]))
    assert a8.calibrate_fp8(calib) == 63 and a8.rec.fp8            # 16 calibration faces: 4 frames x 4 slots
    frs = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 60 + i) for i in range(4)]))   # not the calibration frames
    r = a8.detect_embed_slots(torch.from_numpy(frs).cuda())
    torch.cuda.synchronize()
    counts, cap = r["counts"].cpu().numpy(), r["bbox"].shape[1]
    emb = r["embedding"].cpu().numpy().reshape(4, cap, 512)
    rng = np.random.default_rng(19)
    N = 20_000
    G = rng.standard_normal((N, 512)).astype(np.float32)
    oracle_q, slots, worst = [], [], 0.0
    free = list(rng.permutation(N))
    for i in range(4):
        ob, os_, ok, oemb = _oracle_frame(frs[i], cap_o=4)
        n = len(os_)
        assert counts[i] == n >= 2
        np.testing.assert_allclose(r["bbox"][i, :n].cpu().numpy(), ob, atol=1e-2)
        np.testing.assert_allclose(r["det_score"][i, :n].cpu().numpy(), os_, atol=5e-5)
        np.testing.assert_allclose(r["kps"][i, :n].cpu().numpy(), ok, atol=1e-2)
        e = emb[i, :n]
        cos = (e * oemb).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(oemb, axis=1))
        worst = max(worst, float((1 - cos).max()))
        for j in range(n):
            on = oemb[j] / np.linalg.norm(oemb[j])
            G[free.pop()] = on + 0.02 * rng.standard_normal(512).astype(np.float32)
            oracle_q.append(on.astype(np.float32)); slots.append(i * cap + j)
    print(f"\nC5 pipeline, {len(slots)} faces of 4 x 1080p: fp8 embed 1-cos vs fp32 oracle max {worst:.3e}")
    assert worst < 1e-3
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    gal = OrderedDict((i, G[i]) for i in range(N))
    want_id, want_dec = [], []
    for q in oracle_q:                                            # literal reference loop + decision on the oracle's rows
        bid, bs = omatch.linear_scan(omatch.renormalise(q), gal)
        want_id.append(bid); want_dec.append(omatch.decide_live(str(bid), bs)[0] is not None)
    m = GalleryMatcher("cuda:0", scan="f8")
    m.set_rows(list(range(N)), G, normalise=False)
    strangers = rng.standard_normal((4, 512)).astype(np.float32)
    strangers /= np.linalg.norm(strangers, axis=1, keepdims=True)
    for q in strangers:
        bid, bs = omatch.linear_scan(omatch.renormalise(q), gal)
        want_id.append(bid); want_dec.append(omatch.decide_live(str(bid), bs)[0] is not None)
    Q = torch.cat([r["normed_embedding"][torch.tensor(slots, device="cuda")], torch.from_numpy(strangers).cuda()]).contiguous()
    idx, score = m.match_device(Q)
    dec = m.decide_device(idx, score, 0.4).cpu().numpy()
    got_dec = [bool(d == 1) for d in dec]
    assert got_dec == want_dec and all(want_dec[:-4]) and not any(want_dec[-4:])
    # what the reference hands on is (id | None, score): the id of a face it decides "unknown" is dropped (:549-552), and
    # among random rows that id is a near-tie argmax which an embedding 1e-3 away may legitimately resolve otherwise
    assert [i for i, d in zip(idx.cpu().tolist(), got_dec) if d] == [i for i, d in zip(want_id, want_dec) if d]


def test_c3_4k_frame_sixteen_slots_through_get_vs_oracle():
    """BASELINE config C3's frame (2160 x 3840: 14 pyramid levels, capacity overflow at the large levels) through the
    reference-shaped get() with sixteen face slots, end to end against the CPU oracle INCLUDING embeddings and ids (VERDICT
    r3 item 1b; the detector-only check is test_4k_frame_capacity_overflow_vs_oracle)."""
    from facerecognition_infrenceengine_amd import FaceAnalysis
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
    from make_golden import synth_frame
    assert len(pyramid_scales(2160, 3840)) == 14
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a16 = FaceAnalysis(name="synthetic", arch="r100", cap_o=16).prepare(ctx_id=0)
    frame = synth_frame(2160, 3840, 77)
    ob, os_, ok, oemb = _oracle_frame(frame, cap_o=16)
    faces = a16.get(frame)
    assert len(faces) == len(os_) >= 8
    np.testing.assert_allclose(np.stack([f.bbox for f in faces]), ob, atol=2e-2)
    np.testing.assert_allclose(np.array([f.det_score for f in faces]), os_, atol=5e-5)
    np.testing.assert_allclose(np.stack([f.kps for f in faces]), ok, atol=2e-2)
    emb = np.stack([f.embedding for f in faces])
    cos = (emb * oemb).sum(1) / (np.linalg.norm(emb, axis=1) * np.linalg.norm(oemb, axis=1))
    assert (1 - cos).max() < 1e-3, cos
    rng = np.random.default_rng(5)
    G = rng.standard_normal((10_000, 512)).astype(np.float32)
    rows = rng.choice(10_000, len(faces), replace=False)
    G[rows] = oemb / np.linalg.norm(oemb, axis=1, keepdims=True) + 0.02 * rng.standard_normal(oemb.shape).astype(np.float32)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    m = GalleryMatcher("cuda:0")
    m.set_rows(list(range(10_000)), G, normalise=False)
    ids, score, idx = m.match(np.stack([f.normed_embedding for f in faces]))
    oi, _ = omatch.match_rows_fast(np.stack([omatch.renormalise(e / np.linalg.norm(e)) for e in oemb]), G)
    assert np.array_equal(idx, oi) and np.array_equal(idx, rows)
    # the same frame through the sync-free slot path, two at once (C3's batch form): identical faces
    r = a16.detect_embed_slots(torch.from_numpy(np.stack([frame, frame])).cuda())
    n = len(faces)
    assert r["counts"].cpu().tolist() == [n, n]
    e2 = r["embedding"].cpu().numpy().reshape(2, 16, 512)[:, :n]
    for f in range(2):
        c2 = (e2[f] * oemb).sum(1) / (np.linalg.norm(e2[f], axis=1) * np.linalg.norm(oemb, axis=1))
        assert (1 - c2).max() < 1e-3
        np.testing.assert_allclose(r["bbox"][f, :n].cpu().numpy(), ob, atol=2e-2)


# ------------------------------------------------------------------------------------------------------------------------------
# The path bench.py times, under the suite's oracle (/root/reference/infrenceServer.py:528): batches of
# >= MTCNNHIP.batch_min_pixels pixels (22 M: 11 full-HD frames, 3 4K frames) take the BATCH detector - band-only exact P-Net pass, P-Net conv1 of the levels with
# >= split_pconv1_min_px map pixels and every R-/O-Net layer on the f16 matrix cores with split-precision operands, exact f32
# passes at the thresholds - and these tests ASSERT that path was the one taken (MTCNNHIP._tls.path) besides comparing every
# frame with the CPU oracle, so that a change of the gates cannot silently turn them into tests of the all-f32 detector.
def _oracle_threads(n=16):
    """The CPU oracle's convs are fastest on ~16 torch threads whatever the host has (bench.py cpu_baseline_sweep)."""
    old = torch.get_num_threads()
    torch.set_num_threads(min(n, os.cpu_count() or n))
    return old


def _assert_batch_path(det, nframes, nlevels, mfma_levels_min):
    p = det._tls.path
    assert p["frames"] == nframes and p["batch"] and p["unfused_levels"] == 0, p
    assert p["fused_levels"] == p["band_levels"] == nlevels * p["chunks"], p
    assert len(p["pconv1_mfma_levels"]) >= mfma_levels_min * p["chunks"], p
    assert all(h * w >= det.split_pconv1_min_px for h, w in p["pconv1_mfma_levels"]), p
    assert p["split_ro"], p
    lists = {net: int(lc[0]) for net, lc in det._ro_lists.items()}          # crops the exact f32 R-/O-Net pass took (last chunk)
    assert set(lists) == {0, 1} and all(0 <= lists[n] <= det.ro_list_cap[n] for n in lists), (lists, det.ro_list_cap)
    return p, lists


def _check_slots_vs_oracle(r, frs, which, cap_o, px_tol, cos_tol=1e-3, embed=None):
    """frames ``which`` of the slot-path result ``r`` against the CPU oracle (``embed``: the frames whose faces also go through
    the oracle's fp32 r100 - default all of ``which``); returns [(slot, oracle embedding)] of those"""
    counts = r["counts"].cpu().numpy()
    cap = r["bbox"].shape[1]
    emb = r["embedding"].cpu().numpy().reshape(len(counts), cap, 512)
    out, worst = [], 0.0
    old = _oracle_threads()
    try:
        for i in which:
            with_emb = embed is None or i in embed
            ob, os_, ok, oemb = _oracle_frame(frs[i], cap_o=cap_o, embed=with_emb)
            n = len(os_)
            assert counts[i] == n >= 1, (i, counts[i], n)
            np.testing.assert_allclose(r["bbox"][i, :n].cpu().numpy(), ob, atol=px_tol)
            np.testing.assert_allclose(r["det_score"][i, :n].cpu().numpy(), os_, atol=5e-5)
            np.testing.assert_allclose(r["kps"][i, :n].cpu().numpy(), ok, atol=px_tol)
            if not with_emb:
                continue
            e = emb[i, :n]
            cos = (e * oemb).sum(1) / (np.linalg.norm(e, axis=1) * np.linalg.norm(oemb, axis=1))
            worst = max(worst, float((1 - cos).max()))
            assert (1 - cos).max() < cos_tol, (i, cos)
            out += [(i * cap + j, oemb[j]) for j in range(n)]
    finally:
        torch.set_num_threads(old)
    return out, worst


def _planted_ids_equal(r, faces, rows, seed, scan="f32"):
    """top-1 ids of the GPU's embeddings == the oracle's own, against a gallery with one planted row per face"""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((rows, 512)).astype(np.float32)
    planted = rng.choice(rows, len(faces), replace=False)
    oemb = np.stack([e for _, e in faces])
    G[planted] = oemb / np.linalg.norm(oemb, axis=1, keepdims=True) + 0.02 * rng.standard_normal(oemb.shape).astype(np.float32)
    G /= np.linalg.norm(G, axis=1, keepdims=True)
    m = GalleryMatcher("cuda:0", scan=scan)
    m.set_rows(list(range(rows)), G, normalise=False)
    Q = r["normed_embedding"][torch.tensor([s for s, _ in faces], device="cuda")].contiguous()
    idx, score = m.match_device(Q)
    oi, _ = omatch.match_rows_fast(np.stack([omatch.renormalise(e / np.linalg.norm(e)) for e in oemb]), G)
    assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(oi, planted)
    dec = m.decide_device(idx, score, 0.4).cpu().numpy()
    assert (dec == 1).all()


def test_c2_batch_path_twelve_1080p_frames_vs_oracle_and_path_taken(app):
    """BASELINE config C2's step at a fifth of its batch: 12 x 1080p frames (24.9 Mpixel: the smallest full-HD batch on the batch
    path) through detect_embed_slots with every default left alone; EVERY frame's detections against the CPU oracle (counts,
    boxes / landmarks, scores) and the faces of six frames through its fp32 r100 as well (embeddings, top-1 ids on planted rows);
    the path asserted: all 12 levels through the band-only exact pass, level 0 (pooled conv1 map 323 x 575) through P-Net conv1
    on the f16 matrix cores with exact tiles under the band, the R-/O-Net split cascade with its exact lists inside their
    capacities, and a non-zero number of exactly re-evaluated P-Net cells."""
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 200 + i) for i in range(12)]))
    det = app.det
    assert 8 * 1080 * 1920 < det.batch_min_pixels <= 12 * 1080 * 1920
    det.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
    try:
        r = app.detect_embed_slots(torch.from_numpy(frs).cuda())
        torch.cuda.synchronize()
        refined = int(det.refined_cells[0])
    finally:
        det.refined_cells = None
    p, lists = _assert_batch_path(det, 12, 12, 1)
    assert max(p["pconv1_mfma_levels"]) == (323, 575), p
    assert refined > 0
    faces, worst = _check_slots_vs_oracle(r, frs, range(12), det.cap_o, 1e-2, embed=(0, 2, 5, 7, 8, 11))
    print(f"\nC2 batch path, 12 x 1080p: {len(faces)} faces vs oracle, max 1-cos {worst:.2e}; exact P-Net cells {refined}, "
          f"exact R-/O-Net crops {lists}, path {p}")
    assert len(faces) >= 24
    _planted_ids_equal(r, faces, 10_000, 21)


def test_c3_batch_path_eight_4k_frames_vs_oracle_and_path_taken():
    """BASELINE config C3's batch form: 8 x 4K frames, sixteen face slots per frame, through detect_embed_slots (the 128-face
    embed forward = the stage-14 kernel's smallest batch); frames are independent, so two of the eight go through the CPU
    oracle (a 4K frame costs it ~20 GFLOP of P-Net alone).  Path asserted as for C2 (14 levels; cap_scale overflows at the
    large levels, where the oracle keeps the first 2048 cells in raster order as the kernels do)."""
    from facerecognition_infrenceengine_amd import FaceAnalysis
    from make_golden import synth_frame
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a16 = FaceAnalysis(name="synthetic", arch="r100", cap_o=16).prepare(ctx_id=0)
    frs = np.ascontiguousarray(np.stack([synth_frame(2160, 3840, 300 + i) for i in range(8)]))
    r = a16.detect_embed_slots(torch.from_numpy(frs).cuda())
    torch.cuda.synchronize()
    p, lists = _assert_batch_path(a16.det, 8, 14, 3)
    assert p["chunks"] == 1
    assert int(r["counts"].sum()) >= 64
    faces, worst = _check_slots_vs_oracle(r, frs, (1, 6), 16, 2e-2)
    print(f"\nC3 batch path, 8 x 4K: {len(faces)} faces of frames 1, 6 vs oracle, max 1-cos {worst:.2e}; exact crops {lists}, path {p}")
    _planted_ids_equal(r, faces, 10_000, 22)


def test_c5_batch_path_twelve_1080p_frames_fp8_vs_oracle_and_path_taken():
    """BASELINE config C5 on the batch path: calibrate_fp8 (on other frames), then 12 x 1080p frames with four face slots each -
    the detector on its batch path (asserted), the embed net with 63 fp8 convs, the fp8 coarse scan + exact re-rank; five of the
    twelve frames against the CPU oracle: detector outputs as on the f16 path, fp8 embeddings within north_star's 1 - cos < 1e-3 of
    the fp32 oracle, ids equal on planted rows."""
    from facerecognition_infrenceengine_amd import FaceAnalysis
    from make_golden import synth_frame
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a8 = FaceAnalysis(name="synthetic", arch="r100", cap_o=4).prepare(ctx_id=0)
    calib = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 90 + i) for i in range(4)]))
    assert a8.calibrate_fp8(calib) == 63 and a8.rec.fp8
    frs = np.ascontiguousarray(np.stack([synth_frame(1080, 1920, 400 + i) for i in range(12)]))
    r = a8.detect_embed_slots(torch.from_numpy(frs).cuda())
    torch.cuda.synchronize()
    p, lists = _assert_batch_path(a8.det, 12, 12, 1)
    faces, worst = _check_slots_vs_oracle(r, frs, (0, 3, 5, 7, 10), 4, 1e-2)
    print(f"\nC5 batch path, 12 x 1080p: {len(faces)} faces vs oracle, fp8 embed max 1-cos {worst:.2e}; exact crops {lists}")
    assert len(faces) >= 8
    _planted_ids_equal(r, faces, 20_000, 23, scan="f8")


def test_sixty_four_4k_frames_are_cut_into_groups_that_fit_the_fused_pnet():
    """64 x 4K frames: level 0's split conv1 map (64 x 647 x 1151 x 64 B = 3.05e9 B) exceeds the fused P-Net's 32-bit offsets.
    detect_batch cuts the batch into two groups of 32 (MTCNNHIP._tls.path["chunks"]) instead of dropping the level to the
    layer-by-layer f32 path: every frame's faces equal those of the same frame inside an 8-frame batch (the same batch path;
    frames are independent), and no level ran unfused."""
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    from make_golden import synth_frame
    det = MTCNNHIP(*weights.synth_mtcnn_states(), device="cuda:0", cap_o=4)
    eight = np.ascontiguousarray(np.stack([synth_frame(2160, 3840, 500 + i) for i in range(8)]))
    big = torch.from_numpy(eight).cuda().repeat(8, 1, 1, 1).contiguous()           # 64 frames, 1.6 GB
    want = det.detect_batch(big[:8].contiguous())
    assert det._tls.path["chunks"] == 1
    got = det.detect_batch(big)
    torch.cuda.synchronize()
    p = det._tls.path
    assert p["chunks"] == 2 and p["unfused_levels"] == 0 and p["fused_levels"] == 28 and p["split_ro"], p
    assert int(want[3].sum()) >= 16
    for f in range(64):
        n = int(want[3][f % 8])
        assert int(got[3][f]) == n
        for a, b in zip(want[:3], got[:3]):
            assert torch.equal(a[f % 8, :n], b[f, :n]), f
