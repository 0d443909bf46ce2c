"""GPU parity: MTCNN cascade + 5-point alignment through the C ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import align as oalign
from oracle import detect as odetect

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det():
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    return MTCNNHIP(*weights.synth_mtcnn_states(seed=4321), device="cuda:0")


def test_pyramid_scales_match_oracle():
    from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
    for h, w in ((480, 640), (1080, 1920), (2160, 3840), (120, 160), (13, 400)):
        assert pyramid_scales(h, w) == odetect.pyramid_scales(h, w)
    assert len(pyramid_scales(480, 640)) == 10 and len(pyramid_scales(1080, 1920)) == 12
    assert len(pyramid_scales(2160, 3840)) == 14


def test_pnet_maps_vs_golden(det, golden):
    d = golden("mtcnn_kat.npz")
    frame = torch.from_numpy(d["frame"]).cuda()[None].contiguous()
    tr = {}
    det.detect_batch(frame, trace=tr)
    H, W = d["frame"].shape[:2]
    assert len(tr["pnet_prob"]) == len(odetect.pyramid_scales(H, W))
    for i, (prob, head) in enumerate(zip(tr["pnet_prob"], tr["pnet_head"])):
        np.testing.assert_allclose(prob[0].cpu().numpy(), d[f"pnet_prob_{i}"], atol=2e-5)
        reg = head[0, :, :, 2:6].permute(2, 0, 1).cpu().numpy()
        np.testing.assert_allclose(reg, d[f"pnet_reg_{i}"], atol=2e-5)


def _check_stage(got_b, got_s, want_b, want_s, atol_box):
    assert len(got_s) == len(want_s), (len(got_s), len(want_s))
    np.testing.assert_allclose(got_s, want_s, atol=5e-5)
    np.testing.assert_allclose(got_b, want_b, atol=atol_box)


def test_cascade_stages_vs_golden(det, golden):
    d = golden("mtcnn_kat.npz")
    frame = torch.from_numpy(d["frame"]).cuda()[None].contiguous()
    tr = {}
    boxes, scores, kps, counts = det.detect_batch(frame, trace=tr)
    n1 = int(tr["stage1_counts"][0])
    _check_stage(tr["stage1_boxes"][0, :n1].cpu().numpy(), tr["stage1_scores"][0, :n1].cpu().numpy(),
                 d["stage1_boxes"], d["stage1_scores"], 2e-3)
    # R-Net scores for every stage-1 candidate (slot order == oracle order)
    np.testing.assert_allclose(tr["rnet_prob"][0, :n1].cpu().numpy(), d["rnet_score"], atol=5e-5)
    n2 = int(tr["stage2_counts"][0])
    _check_stage(tr["stage2_boxes"][0, :n2].cpu().numpy(), tr["stage2_scores"][0, :n2].cpu().numpy(),
                 d["stage2_boxes"], d["stage2_scores"], 5e-3)
    np.testing.assert_allclose(tr["onet_prob"][0, :n2].cpu().numpy(), d["onet_score"], atol=5e-5)
    n3 = int(counts[0])
    _check_stage(boxes[0, :n3].cpu().numpy(), scores[0, :n3].cpu().numpy(), d["bbox"], d["score"], 5e-3)
    np.testing.assert_allclose(kps[0, :n3].cpu().numpy(), d["kps"], atol=5e-3)
    assert n3 >= 1


@pytest.mark.parametrize("hw,seed", [((96, 128), 3), ((240, 320), 21), ((13, 200), 1), ((21, 200), 1), ((480, 640), 0)])
def test_cascade_vs_oracle_other_frames(det, hw, seed):
    """Fresh seeded frames (incl. frames whose pyramid has zero / one level and the 640x480 C1 case)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    fr = synth_frame(hw[0], hw[1], seed)
    p, r, o = weights.synth_mtcnn_states(seed=4321)
    ob, os_, ok = odetect.detect(fr, p, r, o)
    boxes, scores, kps, counts = det.detect_batch(torch.from_numpy(fr).cuda()[None].contiguous())
    n = int(counts[0])
    assert n == len(os_)
    if n:
        np.testing.assert_allclose(scores[0, :n].cpu().numpy(), os_, atol=5e-5)
        np.testing.assert_allclose(boxes[0, :n].cpu().numpy(), ob, atol=5e-3)
        np.testing.assert_allclose(kps[0, :n].cpu().numpy(), ok, atol=5e-3)


def test_batch_equals_single_frames(det):
    """Frames of a batch are independent (property used for sharding frames across GPUs)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(120, 160, s) for s in (1, 2, 3)]))
    bb, ss, kk, cc = det.detect_batch(torch.from_numpy(frs).cuda())
    for i in range(3):
        b1, s1, k1, c1 = det.detect_batch(torch.from_numpy(frs[i:i + 1]).cuda())
        assert int(c1[0]) == int(cc[i])
        n = int(c1[0])
        assert torch.equal(b1[0, :n], bb[i, :n]) and torch.equal(s1[0, :n], ss[i, :n])


def test_nms_kernel_vs_oracle(lib):
    from facerecognition_infrenceengine_amd import _lib
    rng = np.random.default_rng(5)
    for n, cap, mode, thr in ((0, 64, 0, 0.5), (1, 64, 0, 0.5), (300, 512, 0, 0.5), (1500, 2048, 0, 0.7),
                              (64, 64, 1, 0.7), (3000, 4096, 0, 0.7)):
        c = rng.uniform(0, 300, (n, 2)); wh = rng.uniform(10, 80, (n, 2))
        boxes = np.zeros((cap, 4), np.float32)
        boxes[:n] = np.floor(np.concatenate([c, c + wh], 1)).astype(np.float32)
        scores = np.zeros(cap, np.float32); scores[:n] = rng.uniform(0.6, 1.0, n).astype(np.float32)
        if n > 10:
            scores[5] = scores[9]                                  # exact score tie -> slot order
        order = np.argsort(-scores[:n], kind="stable")
        keep = odetect.nms(boxes[:n][order], scores[:n][order], thr, "min" if mode else "union")
        want = order[keep][:256]
        bd, sd = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
        aux = torch.arange(cap, dtype=torch.float32, device="cuda").reshape(cap, 1).contiguous()
        cnt = torch.tensor([n], dtype=torch.int32, device="cuda")
        bo = torch.empty((1, 256, 4), device="cuda"); so = torch.empty((1, 256), device="cuda")
        ao = torch.empty((1, 256, 1), device="cuda"); co = torch.empty(1, dtype=torch.int32, device="cuda")
        lib.fr_sort_nms(_lib.ptr(bd), _lib.ptr(sd), _lib.ptr(aux), 1, _lib.ptr(cnt), 1, 1, cap, 0, thr, mode, 256,
                        _lib.ptr(bo), _lib.ptr(so), _lib.ptr(ao), _lib.ptr(co), 256, _lib.stream_ptr())
        k = int(co[0])
        assert k == len(want)
        assert np.array_equal(ao[0, :k, 0].cpu().numpy().astype(np.int64), want)   # identical survivors, identical order


def test_align_vs_golden(lib, golden):
    from facerecognition_infrenceengine_amd import _lib
    d = golden("align_kat.npz")
    frame = torch.from_numpy(d["frame"]).cuda()[None].contiguous()
    kps = torch.from_numpy(d["kps"]).cuda()
    F = kps.shape[0]
    H, W = d["frame"].shape[:2]
    fidx = torch.zeros(F, dtype=torch.int32, device="cuda")
    out = torch.empty((F, 112, 112, 8), dtype=torch.float16, device="cuda")
    u8 = torch.empty((F, 112, 112, 3), dtype=torch.uint8, device="cuda")
    M = torch.empty((F, 2, 3), dtype=torch.float32, device="cuda")
    lib.fr_warp_affine_5pt(_lib.ptr(frame), 1, H, W, _lib.ptr(kps), _lib.ptr(fidx), None, F, 112, _lib.ptr(out),
                           _lib.ptr(u8), _lib.ptr(M), _lib.stream_ptr())
    np.testing.assert_allclose(M.cpu().numpy(), d["M"], rtol=1e-5, atol=1e-4)     # closed form == Umeyama SVD
    diff = np.abs(u8.cpu().numpy().astype(int) - d["crops"].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3                            # uint8 rounding at exact .5 only
    want = np.stack([oalign.crop_to_net(c) for c in u8.cpu().numpy()])            # [F,3,112,112] RGB
    got = out[..., :3].float().cpu().numpy().transpose(0, 3, 1, 2)
    np.testing.assert_allclose(got, want, atol=1e-3)                               # f16 storage
    assert float(out[..., 3:].abs().max()) == 0.0


def test_4k_frame_capacity_overflow_vs_oracle(det):
    """BASELINE config C3 (4K frame, 14 pyramid levels): level 0 has more P-Net candidates than CAP_SCALE, so
    the deterministic overflow rule (first CAP_SCALE cells in raster order) must match the oracle too."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import pyramid_scales
    fr = synth_frame(2160, 3840, 0)
    assert len(pyramid_scales(2160, 3840)) == 14
    p, r, o = weights.synth_mtcnn_states(seed=4321)
    tr = {}
    ob, os_, ok = odetect.detect(fr, p, r, o, trace=tr)
    assert int((tr["pnet_prob"][0] >= 0.6).sum()) > 2048            # the overflow path is really taken
    boxes, scores, kps, counts = det.detect_batch(torch.from_numpy(fr).cuda()[None].contiguous())
    n = int(counts[0])
    assert n == len(os_) and n >= 4
    np.testing.assert_allclose(scores[0, :n].cpu().numpy(), os_, atol=5e-5)
    np.testing.assert_allclose(boxes[0, :n].cpu().numpy(), ob, atol=2e-2)
    np.testing.assert_allclose(kps[0, :n].cpu().numpy(), ok, atol=2e-2)


def test_count_aware_rnet_onet_equal_full_computation(det, monkeypatch):
    """R-Net / O-Net compute only the crop slots that hold a candidate (device-side counts); the kept boxes must be
    bit-identical to computing every slot, on frames whose stages are far from their capacities and on a batch in
    which one frame has no candidate at all."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    frs = np.ascontiguousarray(np.stack([synth_frame(240, 320, s) for s in (4, 21, 5)]))
    frs[1] = 0                                                        # a blank frame: zero candidates
    x = torch.from_numpy(frs).cuda()
    b1, s1, k1, c1 = det.detect_batch(x)
    orig_r, orig_o = det.rnet, det.onet
    monkeypatch.setattr(det, "fused_crop", False)                     # reference: stand-alone crops, EVERY slot computed
    monkeypatch.setattr(det, "rnet", lambda t, B, counts=None, cap=0, x1=None: orig_r(t, B))
    monkeypatch.setattr(det, "onet", lambda t, B, counts=None, cap=0, x1=None: orig_o(t, B))
    b0, s0, k0, c0 = det.detect_batch(x)
    assert torch.equal(c0, c1) and int(c1[1]) == 0 and int(c1.sum()) >= 1
    for f in range(3):
        n = int(c1[f])
        assert torch.equal(b0[f, :n], b1[f, :n]) and torch.equal(s0[f, :n], s1[f, :n]) and torch.equal(k0[f, :n], k1[f, :n])


def test_fused_split_f16_pnet_equals_f32_pnet():
    """P-Net conv2+conv3+heads on the f16 matrix cores with split-precision operands + exact f32 re-evaluation of every
    cell that can pass the threshold (csrc/pnet_fused.hip) against the all-f32 MFMA path, level by level:
    * every head value within 1e-4 (the split-precision error bound the refine margin relies on, measured here);
    * every cell the f32 path keeps (prob >= 0.6) carries f32-accurate values (1e-6) and is kept by the fused path, and
      no other cell is: identical candidate sets;
    * the whole cascade returns bit-identical boxes / scores / landmarks on several frames, incl. a 1080p one."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    st = weights.synth_mtcnn_states(seed=4321)
    fused = MTCNNHIP(*st, device="cuda:0", fused_pnet=True)
    plain = MTCNNHIP(*st, device="cuda:0", fused_pnet=False)
    fused.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
    total = kept = 0
    worst = 0.0
    for hw, seed in (((240, 320), 21), ((480, 640), 0), ((1080, 1920), 31), ((96, 128), 3)):
        fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(hw[0], hw[1], seed + k) for k in range(2)]))).cuda()
        fused.p23_all_heads = True          # the level-by-level comparison reads the heads of EVERY cell
        for s in pyramid_scales(*hw):
            with torch.cuda.device("cuda:0"):
                fused._s = plain._s = torch.cuda.current_stream().cuda_stream
                hf, h1, w1 = fused.pnet_level(fr, s)
                hp, h2, w2 = plain.pnet_level(fr, s)
            assert (h1, w1) == (h2, w2) and hf.shape == hp.shape
            d = (hf - hp).abs()
            worst = max(worst, float(d.max()))
            keep_p = (hp[..., 1] - hp[..., 0]) >= np.log(1.5)
            keep_f = (hf[..., 1] - hf[..., 0]) >= np.log(1.5)
            assert torch.equal(keep_p, keep_f)
            if keep_p.any():
                assert float(d[keep_p].max()) == 0.0, float(d[keep_p].max())     # the exact pass runs the all-f32 path's own MFMA chains
            total += keep_p.numel(); kept += int(keep_p.sum())
        fused.p23_all_heads = False         # the product setting: only the re-evaluated cells' heads are written
        a = fused.detect_batch(fr); b = plain.detect_batch(fr)
        assert torch.equal(a[3], b[3])
        for f in range(fr.shape[0]):
            n = int(a[3][f])
            for x, y in zip(a[:3], b[:3]):
                assert torch.equal(x[f, :n], y[f, :n])
    assert worst < 1e-4, worst
    refined = int(fused.refined_cells[0])
    print(f"\nfused P-Net: max |head - f32 head| = {worst:.2e}; kept {kept} of {total} cells; re-evaluated exactly: {refined // 2} per pass")
    assert kept >= 100 and refined >= kept


def test_fused_pnet_exact_pass_with_every_cell_on_its_work_list():
    """The exact pass of the fused P-Net takes its cells from per-block lists the split-precision kernel fills
    (csrc/pnet_fused.hip).  With a face threshold near zero (almost) EVERY cell is on a list: full segments, many trips per
    wave, ragged tiles at the map borders, several frames.  Every listed cell must carry the all-f32 path's head values
    bit for bit, with and without the approximate head rows of the others."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    st = weights.synth_mtcnn_states(seed=77)
    thr = (1e-4, 0.7, 0.7)
    fused = MTCNNHIP(*st, device="cuda:0", fused_pnet=True, thresholds=thr)
    plain = MTCNNHIP(*st, device="cuda:0", fused_pnet=False, thresholds=thr)
    fused.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
    logit_thr = float(np.log(thr[0] / (1.0 - thr[0])))
    listed = total = 0
    for hw, nfr in (((131, 197), 3), ((480, 640), 2)):
        fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(hw[0], hw[1], 50 + k) for k in range(nfr)]))).cuda()
        for s in pyramid_scales(*hw)[:4]:
            for all_heads in (True, False):
                fused.p23_all_heads = all_heads
                with torch.cuda.device("cuda:0"):
                    fused._s = plain._s = torch.cuda.current_stream().cuda_stream
                    hf, h1, w1 = fused.pnet_level(fr, s)
                    dl = fused._dl[0][:fr.shape[0] * h1 * w1].reshape(fr.shape[0], h1, w1).clone()
                    hp, h2, w2 = plain.pnet_level(fr, s)
                torch.cuda.synchronize()
                on_list = dl >= logit_thr - fused.refine_margin
                assert torch.equal(hf[on_list], hp[on_list])
                # whatever the split-precision pass ruled out is below the threshold in the f32 path too
                assert bool(((hp[..., 1] - hp[..., 0])[~on_list] < logit_thr).all())
            listed += int(on_list.sum()); total += on_list.numel()
    assert listed > 0.9 * total, (listed, total)
    assert int(fused.refined_cells[0]) == 2 * listed


def test_eager_single_frame_call_list_equals_launch_by_launch():
    """An eager single-frame detect_batch of a frame shape seen twice before replays ONE recorded C call list
    (fr_detect_sequence, MTCNNHIP.use_sequence): same kernels, same arguments - but for the frame and the four result tensors,
    which are patched in - so boxes / scores / landmarks / counts equal the launch-by-launch call bit for bit, frame after
    frame, across a change of frame shape and back, and after a threshold changed (a new recording)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    st = weights.synth_mtcnn_states(seed=4321)
    seq = MTCNNHIP(*st, device="cuda:0")
    ref = MTCNNHIP(*st, device="cuda:0")
    ref.use_sequence = False

    def same(hw, seed):
        fr = torch.from_numpy(synth_frame(hw[0], hw[1], seed)[None]).cuda()
        a, b = seq.detect_batch(fr), ref.detect_batch(fr)
        torch.cuda.synchronize()
        assert torch.equal(a[3], b[3]) and int(a[3][0]) >= 1
        n = int(a[3][0])
        for x, y in zip(a[:3], b[:3]):
            assert torch.equal(x[0, :n], y[0, :n])
        return a

    kept = [same((240, 320), s) for s in range(20, 26)]            # calls 1, 2: eager (cache fill, recording); 3 ..: replayed
    assert len(seq._tls.seqs) == 1
    first = [t.clone() for t in kept[3]]
    for s in range(30, 33):
        same((200, 264), s)                                       # another frame shape: its own cache and list
    same((240, 320), 40)                                          # back: the first list is still valid
    assert all(torch.equal(x, y) for x, y in zip(first, kept[3])) # results handed out earlier were not overwritten
    assert len(seq._tls.seqs) == 2
    seq.thresholds = ref.thresholds = (0.6, 0.7, 0.75)            # a configuration change invalidates the recorded list
    for s in range(50, 54):
        same((240, 320), s)
    # two frames at once (still a single-frame-mode batch): lists are per batch shape
    fr2 = torch.from_numpy(np.stack([synth_frame(240, 320, 60), synth_frame(240, 320, 61)])).cuda()
    for _ in range(4):
        a, b = seq.detect_batch(fr2), ref.detect_batch(fr2)
        assert torch.equal(a[3], b[3])
        for f in range(2):
            n = int(a[3][f])
            for x, y in zip(a[:3], b[:3]):
                assert torch.equal(x[f, :n], y[f, :n])


def test_call_list_is_rerecorded_when_thresholds_or_caps_change_and_never_covers_other_entry_points():
    """Guards of the recorded call list: (i) ``thresholds`` and ``cap_o`` changed between two
    calls of ONE frame shape: the old list is dropped on the spot and a new one recorded - results keep following the
    launch-by-launch engine bit for bit, and the result tensors take the new ``cap_o`` shape; (ii) with ``fused_crop = False``
    the cascade launches fr_crop_resize_norm, which fr_detect_sequence cannot replay: nothing is recorded, four and more
    calls stay equal to ``use_sequence = False`` (a replay that skipped the crops would hand R-/O-Net the PREVIOUS
    frame's crops); switching the attribute back records afresh; (iii) a recording during which an entry point outside
    _lib.SEQ_FN is called is refused by the recorder itself; (iv) ``one_stream`` calls are never replayed."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    st = weights.synth_mtcnn_states(seed=4321)
    seq = MTCNNHIP(*st, device="cuda:0")
    ref = MTCNNHIP(*st, device="cuda:0")
    ref.use_sequence = False
    replays = []
    orig = seq._replay
    seq._replay = lambda s, f: (replays.append(1), orig(s, f))[1]

    def same(seed, hw=(240, 320)):
        fr = torch.from_numpy(synth_frame(hw[0], hw[1], seed)[None]).cuda()
        a, b = seq.detect_batch(fr), ref.detect_batch(fr)
        torch.cuda.synchronize()
        assert torch.equal(a[3], b[3]) and int(a[3][0]) >= 1
        n = int(a[3][0])
        for x, y in zip(a[:3], b[:3]):
            assert x.shape == y.shape and torch.equal(x[0, :n], y[0, :n])
        return a

    def setboth(**kw):
        for d in (seq, ref):
            for k, v in kw.items():
                setattr(d, k, v)

    for s in range(4):
        same(100 + s)
    assert len(replays) == 2 and len(seq._tls.seqs) == 1                    # calls 3 and 4 were replays
    old = next(iter(seq._tls.seqs.values()))
    # (i) thresholds + cap_o between two calls of the same frame shape
    setboth(thresholds=(0.6, 0.65, 0.72), cap_o=5)
    a = same(110)
    assert len(replays) == 2 and a[0].shape == (1, 5, 4)                    # not replayed: the old list went away ...
    lst = next(iter(seq._tls.seqs.values()), None)
    assert lst is not old
    for s in range(3):
        same(111 + s)
    assert len(replays) >= 4 and next(iter(seq._tls.seqs.values()))["cfg"][3] == (0.6, 0.65, 0.72)   # ... a new one is in use
    # (ii) the stand-alone crop kernel is outside SEQ_FN
    n0 = len(replays)
    setboth(fused_crop=False)
    for s in range(5):
        same(120 + s)
    assert len(replays) == n0 and not seq._tls.seqs                          # nothing replayed, nothing kept
    setboth(fused_crop=True)
    for s in range(4):
        same(130 + s)
    assert len(replays) > n0 and len(seq._tls.seqs) == 1                     # back: recorded afresh (the old list was dropped)
    # (iii) the recorder refuses a list when another entry point ran during the recording
    lib = _lib.load()
    lib.start_recording()
    x = torch.randn(4, 512, device="cuda")
    lib.fr_l2norm_rows_f32(_lib.ptr(x), _lib.ptr(x), 4, 512, _lib.stream_ptr())
    assert lib.stop_recording() is None and lib.recording_invalid() == "fr_l2norm_rows_f32"
    lib.start_recording()
    assert lib.stop_recording() == [] and lib.recording_invalid() is None
    # (iv) one_stream (profiling: every level on the caller's stream) is honoured, i.e. never replayed
    n1 = len(replays)
    setboth(one_stream=True)
    for s in range(4):
        same(140 + s)
    assert len(replays) == n1


@pytest.mark.parametrize("negative_slopes", [False, True])
def test_pnet_conv1_kernel_vs_oracle_and_16x16x4_form(negative_slopes):
    """P-Net conv1 (+ pyramid resize, PReLU, 2x2 ceil pool) runs as its own 4x4x1-MFMA kernel (csrc/pnet_conv1.hip).
    Against the oracle's resize + conv + PReLU + pool in f32 (tolerance: summation order), and bit for bit against the
    16x16x4-MFMA form of the same layer (layer id 3: same fma chain) - pooled map and split-f16 copy, on levels whose
    sizes exercise ragged tiles, one-row / one-column maps and the last-frame pull-back.  With some NEGATIVE PReLU
    slopes the kernel may not pool before it activates (the max pool only commutes with a non-decreasing function)."""
    import math
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    st = weights.synth_mtcnn_states(seed=99)
    if negative_slopes:
        st[0]["prelu1.weight"] = st[0]["prelu1.weight"] * torch.tensor([1, -1, 1, -0.5, 1, 1, -2, 1, 1, -1.0])
    d = MTCNNHIP(*st, device="cuda:0")
    p1, lib = d.p1, d.lib
    g = torch.Generator(device="cuda").manual_seed(5)
    w1, b1, s1 = (st[0][k].float() for k in ("conv1.weight", "conv1.bias", "prelu1.weight"))
    for (N, H, W) in [(2, 120, 160), (1, 250, 333), (3, 37, 53), (1, 480, 640)]:
        frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device="cuda", dtype=torch.uint8)
        for sc in pyramid_scales(H, W):
            hs, ws = int(math.ceil(H * sc)), int(math.ceil(W * sc))
            if hs < 3 or ws < 3:
                continue
            h, w = p1.out_hw(hs, ws)
            outs = []
            for layer in (0, 3):
                y = torch.full((N, h, w, 12), float("nan"), dtype=torch.float32, device="cuda")
                xs = torch.full((N, h, w, 64), 0x7f, dtype=torch.uint8, device="cuda")
                rc = lib.fr_dconv_mfma_f32(layer, None, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope), _lib.ptr(y),
                                           N, hs, ws, None, None, _lib.ptr(frames), H, W, None, 0, _lib.ptr(xs),
                                           _lib.stream_ptr())
                assert rc == 0
                outs.append((y, xs))
            torch.cuda.synchronize()
            assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)), (N, H, W, hs, ws)
            assert torch.equal(outs[0][1], outs[1][1]), (N, H, W, hs, ws)
            # oracle: level image -> conv -> PReLU -> ceil-mode 2x2 pool (first frame)
            rgb = frames[0].cpu().numpy()[:, :, ::-1].astype(np.float32)
            x = odetect._to_net(odetect.resize_bilinear(rgb, hs, ws))
            c = torch.nn.functional.conv2d(x, w1, b1)
            c = torch.where(c > 0, c, c * s1[None, :, None, None])
            want = torch.nn.functional.max_pool2d(c, 2, 2, ceil_mode=True)[0].permute(1, 2, 0).numpy()
            got = outs[0][0][0, :, :, :10].cpu().numpy()
            np.testing.assert_allclose(got, want, atol=3e-5, rtol=1e-5)
            assert float(outs[0][0][..., 10:].abs().max()) == 0.0
            # the f16 matrix-core form (fr_pnet_conv1_band mode 0, the batch path's band mode): split map = the f32 map within ~1e-6,
            # channels 10..15 zero, nothing written outside the map; its f32 view (tests only) is the map the halves encode
            y16 = torch.full((N, h, w, 12), float("nan"), dtype=torch.float32, device="cuda")
            xs16 = torch.full((N, h, w, 64), 0x7f, dtype=torch.uint8, device="cuda")
            assert lib.fr_pnet_conv1_band(0, _lib.ptr(frames), N, H, W, hs, ws, _lib.ptr(p1.w), _lib.ptr(p1.b), _lib.ptr(p1.slope),
                                          _lib.ptr(y16), _lib.ptr(xs16), None, None, 0, _lib.stream_ptr()) == 0
            torch.cuda.synchronize()
            hl = xs16.view(torch.float16).reshape(N, h, w, 2, 16).float()
            assert float(hl[..., 10:].abs().max()) == 0.0
            dec = hl[..., 0, :12] + hl[..., 1, :12]
            tol = 4e-6 * max(1.0, float(outs[0][0].abs().max()))
            assert float((dec - outs[0][0]).abs().max()) <= tol, (N, H, W, hs, ws, float((dec - outs[0][0]).abs().max()), tol)
            assert float((y16 - outs[0][0]).abs().max()) <= tol


@pytest.mark.parametrize("negative_slopes", [False, True])
def test_crop_conv1_equals_crop_then_layer(negative_slopes):
    """R-/O-Net's first layer runs fused with the crop (csrc/ro_conv1.hip: crop -> resize -> conv 3x3 on 4x4x1 MFMAs ->
    PReLU -> 3x3/s2 ceil pool, no crop tensor in HBM).  Must equal fr_crop_resize_norm followed by layer 10 / 20 of
    fr_dconv_mfma_f32 BIT FOR BIT on every valid slot: boxes inside the frame, sticking out of every side, one pixel
    wide, degenerate (empty), in the last frame (8-byte loads pulled back), with partially filled slot lists; also with
    negative PReLU slopes (no pooling before the activation then)."""
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    st = weights.synth_mtcnn_states(seed=77)
    if negative_slopes:
        st[1]["prelu1.weight"] = st[1]["prelu1.weight"] * torch.where(torch.arange(28) % 3 == 0, -1.0, 1.0)
        st[2]["prelu1.weight"] = st[2]["prelu1.weight"] * torch.where(torch.arange(32) % 5 == 0, -0.5, 1.0)
    d = MTCNNHIP(*st, device="cuda:0")
    lib = d.lib
    N, H, W, cap = 3, 97, 131, 40
    g = torch.Generator(device="cuda").manual_seed(11)
    frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device="cuda", dtype=torch.uint8)
    x1 = torch.rand((N, cap), generator=g, device="cuda") * (W + 40) - 30
    y1 = torch.rand((N, cap), generator=g, device="cuda") * (H + 40) - 30
    sz = torch.rand((N, cap), generator=g, device="cuda") * 70 + 1
    boxes = torch.stack([x1, y1, x1 + sz, y1 + sz * 1.2], -1).contiguous()
    boxes[0, 0] = torch.tensor([5.0, 5.0, 5.0, 60.0])                 # one pixel wide
    boxes[0, 1] = torch.tensor([50.0, 50.0, 40.0, 60.0])              # empty (x2 < x1)
    boxes[2, 2] = torch.tensor([W - 3.0, H - 3.0, W + 10.0, H + 9.0])   # last frame, bottom-right corner
    boxes[1, 3] = torch.tensor([-20.0, -20.0, W + 20.0, H + 20.0])    # larger than the frame
    counts = torch.tensor([cap, 17, 25], dtype=torch.int32, device="cuda")
    valid = (torch.arange(cap, device="cuda")[None, :] < counts[:, None]).reshape(-1)
    d._s = _lib.stream_ptr()
    for net, size, layer, c in ((0, 24, d.r1, 28), (1, 48, d.o1, 32)):
        crops = torch.empty((N * cap, size, size, 4), dtype=torch.float32, device="cuda")
        lib.fr_crop_resize_norm(_lib.ptr(frames), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, size, _lib.ptr(crops),
                                _lib.stream_ptr())
        want, ho, wo = d._dconv(crops, layer, N * cap, size, size, counts=counts, cap=cap)
        got = d.crop_conv1(net, frames, boxes, counts, cap)
        torch.cuda.synchronize()
        assert got.shape == want.shape == (N * cap, ho, wo, c)
        assert torch.equal(got[valid].view(torch.int32), want[valid].view(torch.int32)), net


@pytest.mark.parametrize("negative_slopes", [False, True])
def test_split_f16_conv2_layers_vs_f32_layers(negative_slopes):
    """R-/O-Net's second layer on the f16 matrix cores with split-precision operands (csrc/ro_conv2.hip) against the f32
    layers it replaces (fr_crop_conv1_f32 -> layer 11 / 21 of fr_dconv_mfma_f32), from the same frames and boxes: the
    split conv1 map is hi + lo of the f32 map to 2^-21 relative (hi / lo layout and the zero K padding checked directly), the
    pooled conv2 maps agree to ~1e-6 of the map's scale on every valid slot - boxes sticking out of every side, odd slot
    counts (the R-Net kernel takes crops in pairs), an empty frame, negative PReLU slopes (no pooling before the activation)."""
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    st = weights.synth_mtcnn_states(seed=78)
    if negative_slopes:
        st[1]["prelu2.weight"] = st[1]["prelu2.weight"] * torch.where(torch.arange(48) % 3 == 0, -1.0, 1.0)
        st[2]["prelu2.weight"] = st[2]["prelu2.weight"] * torch.where(torch.arange(64) % 5 == 0, -0.5, 1.0)
    d = MTCNNHIP(*st, device="cuda:0")
    lib = d.lib
    N, H, W, cap = 4, 97, 131, 24
    g = torch.Generator(device="cuda").manual_seed(12)
    frames = torch.randint(0, 256, (N, H, W, 3), generator=g, device="cuda", dtype=torch.uint8)
    x1 = torch.rand((N, cap), generator=g, device="cuda") * (W + 40) - 30
    y1 = torch.rand((N, cap), generator=g, device="cuda") * (H + 40) - 30
    sz = torch.rand((N, cap), generator=g, device="cuda") * 70 + 1
    boxes = torch.stack([x1, y1, x1 + sz, y1 + sz * 1.2], -1).contiguous()
    counts = torch.tensor([cap, 17, 0, 5], dtype=torch.int32, device="cuda")
    valid = (torch.arange(cap, device="cuda")[None, :] < counts[:, None]).reshape(-1)
    d._s = _lib.stream_ptr()
    for net, layer2, p1, c1, shape2 in ((0, d.r2, 11, 28, (4, 4, 48)), (1, d.o2, 23, 32, (10, 10, 64))):
        m1 = d.crop_conv1(net, frames, boxes, counts, cap)                           # f32 [N*cap, p1, p1, c1]
        want, _, _ = d._dconv(m1, layer2, N * cap, p1, p1, counts=counts, cap=cap)
        w1, b1, s1 = d._rc1 if net == 0 else d._oc1
        xs = torch.full((N * cap, p1 * p1, 128), 0x7f, dtype=torch.uint8, device="cuda")
        lib.fr_crop_conv1_split(net, _lib.ptr(frames), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, _lib.ptr(w1), _lib.ptr(b1),
                                _lib.ptr(s1), _lib.ptr(xs), 0, _lib.stream_ptr())
        d.split_conv1 = False
        got, _ = d.crop_conv12_split(net, frames, boxes, counts, cap)
        # ... and with the first layer's conv on the f16 matrix cores too (the product setting)
        xs16 = torch.full((N * cap, p1 * p1, 128), 0x7f, dtype=torch.uint8, device="cuda")
        lib.fr_crop_conv1_split(net, _lib.ptr(frames), N, H, W, _lib.ptr(boxes), _lib.ptr(counts), cap, _lib.ptr(w1), _lib.ptr(b1),
                                _lib.ptr(s1), _lib.ptr(xs16), 1, _lib.stream_ptr())
        d.split_conv1 = True
        got16, _ = d.crop_conv12_split(net, frames, boxes, counts, cap)
        torch.cuda.synchronize()
        hl16 = xs16.view(torch.float16).reshape(N * cap, p1, p1, 2, 32).float()
        m16 = (hl16[..., 0, :] + hl16[..., 1, :])[valid][..., :c1]
        e1 = float((m16 - m1[valid]).abs().max())
        assert e1 <= 4e-6 * float(m1[valid].abs().max()), (net, e1)                  # the f16-matrix-core conv1 map
        assert float(hl16[valid][..., c1:].abs().max() if c1 < 32 else 0.0) == 0.0
        e2 = float((got16[valid] - want[valid]).abs().max())
        print(f"\nnet {net}: conv1 on the f16 matrix cores: map max |d| {e1:.3e}; conv2 map behind it max |d| {e2:.3e}")
        assert e2 <= 1.5e-5 * float(want[valid].abs().max()), (net, e2)
        hl = xs.view(torch.float16).reshape(N * cap, p1, p1, 2, 32).float()
        hi, lo = hl[..., 0, :], hl[..., 1, :]
        assert torch.equal(hi[valid][..., :c1], m1[valid].half().float())             # hi = f16(x)
        assert float((hi[valid][..., :c1] + lo[valid][..., :c1] - m1[valid]).abs().max()) <= 2.0 ** -21 * float(m1[valid].abs().max())
        assert float(hl[valid][..., c1:].abs().max() if c1 < 32 else 0.0) == 0.0      # the K padding is zero
        assert got.shape == want.shape == (N * cap, *shape2)
        err = float((got[valid] - want[valid]).abs().max())
        scale = float(want[valid].abs().max())
        print(f"\nnet {net}: split-f16 conv2 vs f32 conv2: max |d| {err:.3e} at scale {scale:.3f}")
        assert err <= 4e-6 * scale, (net, err, scale)


def test_split_ro_cascade_vs_f32_cascade_and_exact_pass():
    """The cascade with the split-precision second layers (MTCNNHIP.split_ro, batches of >= 8 frames) against the all-f32
    cascade on eight frames: (i) default margin: the same faces in the same order, scores within 5e-6, boxes / landmarks
    within 1e-3 px (the split heads are ~1e-6 from the f32 ones; crops near the threshold are re-evaluated exactly, so the
    keep / reject decisions are the f32 ones); (ii) with the margin opened wide every valid crop goes through the exact
    pass (work lists sized to hold them all): the whole cascade is then BIT-identical to the f32 one - the list, the
    compact f32 layers and the scatter put every head row back in its slot; (iii) the exact pass's list is small at the
    default margin."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    st = weights.synth_mtcnn_states()
    split = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0)       # (8 small frames: below the default pixel gate of the batch path)
    ref = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0)
    ref.split_ro = False
    frs = np.ascontiguousarray(np.stack([synth_frame(360, 640, 40 + i) for i in range(8)]))
    frs[5] = 0                                                        # a frame with no candidate
    x = torch.from_numpy(frs).cuda()
    want = ref.detect_batch(x)
    got = split.detect_batch(x)
    torch.cuda.synchronize()
    assert torch.equal(got[3], want[3]) and int(want[3].sum()) >= 8 and int(want[3][5]) == 0
    listed = [int(split._ro_lists[k][0]) for k in (0, 1)]
    valid2 = 8 * 512
    print(f"\nsplit R-/O-Net: exact pass took {listed} crops (R-Net, O-Net) at margin {split.ro_margin}")
    assert listed[0] < valid2 // 20 and listed[1] < 64
    for f in range(8):
        n = int(want[3][f])
        assert float((got[1][f, :n] - want[1][f, :n]).abs().max() if n else 0.0) <= 5e-6
        assert float((got[0][f, :n] - want[0][f, :n]).abs().max() if n else 0.0) <= 1e-3
        assert float((got[2][f, :n] - want[2][f, :n]).abs().max() if n else 0.0) <= 1e-3
    split.ro_margin = 1e9
    split.ro_list_cap = (8 * 512, 8 * 64)
    exact = split.detect_batch(x)
    torch.cuda.synchronize()
    assert torch.equal(exact[3], want[3])
    for f in range(8):
        n = int(want[3][f])
        for a, b in zip(exact[:3], want[:3]):
            assert torch.equal(a[f, :n], b[f, :n])


def test_pnet_band_mode_vs_exact_kept_cells():
    """Batches of >= 8 frames re-evaluate exactly only the P-Net cells within ``refine_margin`` of the face threshold
    (MTCNNHIP.pnet_band): level by level against the all-f32 path - identical kept-cell sets (the decisions are f32 decisions),
    the cells inside the band carry the f32 path's bits, the kept cells above it its values to ~1e-5; far fewer cells go
    through the exact pass than with ``pnet_band = False``; and the whole cascade on eight frames returns the same faces in
    the same order within 5e-6 (scores) / 1e-3 px of the all-exact setting."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    st = weights.synth_mtcnn_states(seed=4321)
    band = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0)
    plain = MTCNNHIP(*st, device="cuda:0", fused_pnet=False, batch_min_pixels=0)
    exact = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0).set_exact(True)
    assert exact.pnet_band is False and exact.split_ro is False
    band.split_ro = exact.split_ro = plain.split_ro = False           # this test isolates the P-Net
    band.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
    exact.refined_cells = torch.zeros(1, dtype=torch.int32, device="cuda")
    hw = (360, 640)
    fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(hw[0], hw[1], 70 + k) for k in range(8)]))).cuda()
    lt = float(np.log(1.5))
    kept = inband = 0
    worst = 0.0
    for s in pyramid_scales(*hw):
        with torch.cuda.device("cuda:0"):
            band._s = plain._s = torch.cuda.current_stream().cuda_stream
            hb_, h1, w1 = band.pnet_level(fr, s)
            hp, h2, w2 = plain.pnet_level(fr, s)
        torch.cuda.synchronize()
        dp = hp[..., 1] - hp[..., 0]
        keep_p = dp >= lt
        dl = band._dl[0][:8 * h1 * w1].reshape(8, h1, w1)
        cand = dl >= lt - band.refine_margin                              # what fr_pnet_candidates reads
        assert bool((~cand | (hb_[..., 1] - hb_[..., 0] >= lt) == (~cand | keep_p)).all())       # same decisions on every candidate
        assert bool((dp[~cand] < lt).all())                               # and nothing the f32 path keeps was ruled out
        inb = cand & (dl <= lt + band.refine_margin)
        assert torch.equal(hb_[inb], hp[inb])                             # inside the band: the f32 path's bits
        if keep_p.any():
            worst = max(worst, float((hb_[keep_p] - hp[keep_p]).abs().max()))
        kept += int(keep_p.sum()); inband += int(inb.sum())
    assert kept >= 100 and worst < 1e-4, (kept, worst)
    a = band.detect_batch(fr)
    n_band = int(band.refined_cells[0])
    b = exact.detect_batch(fr)
    n_exact = int(exact.refined_cells[0])
    torch.cuda.synchronize()
    print(f"\nP-Net band mode: kept {kept} cells, {inband} inside the band; max |head - f32 head| of a kept cell {worst:.2e}; "
          f"exact pass: {n_band - inband} cells per cascade against {n_exact}")
    assert n_band - inband < n_exact // 5
    assert torch.equal(a[3], b[3]) and int(a[3].sum()) >= 8
    for f in range(8):
        n = int(a[3][f])
        if n:
            assert float((a[1][f, :n] - b[1][f, :n]).abs().max()) <= 5e-6
            assert float((a[0][f, :n] - b[0][f, :n]).abs().max()) <= 1e-3
            assert float((a[2][f, :n] - b[2][f, :n]).abs().max()) <= 1e-3


def test_all_levels_in_three_launches_equal_per_level_launches():
    """fr_pnet_finish_levels runs the P-Net's exact pass and the ordered candidate extraction for up to 16 pyramid levels per call
    (the product passes ONE level per call, from MTCNNHIP.pnet_level: the all-levels form was measured slower inside the pipeline
    and its host switch is gone).  The C entry keeps the general form; checked here through the C ABI: all levels of a 9-frame batch
    in one call against the per-level launches (fr_pnet23_split_f16's own exact pass + fr_pnet_candidates) - the same counts, the
    same cells in the same order, the same bits - with the band-only exact pass and with every kept cell re-evaluated, and with a
    frame that holds no candidate at all."""
    import math, sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    d = MTCNNHIP(*weights.synth_mtcnn_states(seed=4321), device="cuda:0")
    lib = d.lib
    frs = np.ascontiguousarray(np.stack([synth_frame(300, 500, 80 + k) for k in range(9)]))
    frs[4] = 0
    fr = torch.from_numpy(frs).cuda()
    N, H, W = 9, 300, 500
    t0, cs = d.thresholds[0], d.cap_scale
    lt = math.log(t0 / (1.0 - t0))
    s0 = torch.cuda.current_stream().cuda_stream
    scales = pyramid_scales(H, W)
    p23 = [_lib.ptr(t) for t in d._p23]
    for band in (True, False):
        hi = lt + d.refine_margin if band else float("-inf")
        lists, keep = {}, []
        for mode in ("per_level", "one_call"):
            nref = torch.zeros(1, dtype=torch.int32, device="cuda")
            lb, ls, lr, lc = (torch.zeros(len(scales), N, cs, 4, device="cuda"), torch.zeros(len(scales), N, cs, device="cuda"),
                              torch.zeros(len(scales), N, cs, 4, device="cuda"), torch.zeros(len(scales), N, dtype=torch.int32, device="cuda"))
            lv = (_lib.PnetLevel * len(scales))()
            with torch.cuda.device("cuda:0"):
                d._s = s0
                for li, sc in enumerate(scales):
                    hs, ws = int(math.ceil(H * sc)), int(math.ceil(W * sc))
                    h, w = d.p1.out_hw(hs, ws)
                    xs = torch.empty((N, h, w, 64), dtype=torch.uint8, device="cuda")
                    x, _, _ = d._dconv(None, d.p1, N, hs, ws, frames=fr, y_split=xs)
                    head = torch.empty((N, h - 4, w - 4, 6), device="cuda")
                    wsp = torch.empty(lib.fr_pnet23_workspace_bytes(N, h, w) // 4, device="cuda")
                    bc = torch.zeros(N * (-(-(h - 4) * (w - 4) // 256)), dtype=torch.int32, device="cuda")
                    lib.fr_pnet23_split_f16(_lib.ptr(x), _lib.ptr(xs), N, h, w, *p23, _lib.ptr(head), 0 if mode == "per_level" else 2,
                                            lt - d.refine_margin, hi, _lib.ptr(nref), _lib.ptr(wsp), wsp.numel() * 4, s0)
                    if mode == "per_level":
                        lib.fr_pnet_candidates(_lib.ptr(head), N, h - 4, w - 4, float(sc), t0, cs, _lib.ptr(lb[li]), _lib.ptr(ls[li]),
                                               _lib.ptr(lr[li]), _lib.ptr(lc[li]), _lib.ptr(bc), None, _lib.ptr(wsp), lt - d.refine_margin, s0)
                    else:
                        lv[li] = _lib.PnetLevel(x.data_ptr(), head.data_ptr(), wsp.data_ptr(), h, w, float(sc), lb[li].data_ptr(),
                                                ls[li].data_ptr(), lr[li].data_ptr(), lc[li].data_ptr(), bc.data_ptr())
                    keep.append((x, xs, head, wsp, bc))
                if mode == "one_call":
                    lib.fr_pnet_finish_levels(lv, len(scales), N, *p23, t0, cs, lt - d.refine_margin, _lib.ptr(nref), s0)
            torch.cuda.synchronize()
            lists[mode] = (lb, ls, lr, lc, int(nref[0]))
        (ab, as_, ar, ac, na), (bb, bs, br, bc_, nb) = lists["per_level"], lists["one_call"]
        assert torch.equal(ac, bc_) and int(ac.sum()) >= 100 and int(ac[:, 4].sum()) == 0
        assert na == nb > 0
        for li in range(len(scales)):
            for f in range(N):
                n = min(int(ac[li, f]), cs)
                assert torch.equal(ab[li, f, :n], bb[li, f, :n]) and torch.equal(as_[li, f, :n], bs[li, f, :n])
                assert torch.equal(ar[li, f, :n], br[li, f, :n])


def test_split_gemm_tail_layers_vs_f32_layers():
    """R-Net conv3 / dense4 and O-Net conv4 / dense5 as split-precision GEMMs on the f16 matrix cores (csrc/ro_gemm.hip) against
    the f32 layers of the same ids (fr_dconv_mfma_f32) on random maps: ~1e-6 of the output's scale on every valid slot, with
    partially filled and empty frames and a slot count that is no multiple of the kernel's 64-row tiles."""
    from facerecognition_infrenceengine_amd import _lib, weights
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP
    d = MTCNNHIP(*weights.synth_mtcnn_states(seed=79), device="cuda:0")
    d._s = _lib.stream_ptr()
    g = torch.Generator(device="cuda").manual_seed(5)
    N, cap = 5, 30
    counts = torch.tensor([cap, 7, 0, 19, 1], dtype=torch.int32, device="cuda")
    valid = (torch.arange(cap, device="cuda")[None, :] < counts[:, None]).reshape(-1)
    B = N * cap
    for lid, layer, shape_in, shape_out in ((12, d.r3, (4, 4, 48), (3, 3, 64)), (13, d.r4, (3, 3, 64), (1, 1, 128)),
                                            (22, d.o3, (10, 10, 64), (4, 4, 64)),      # conv3 + the fused 2x2/s2 pool
                                            (23, d.o4, (4, 4, 64), (3, 3, 128)), (24, d.o5, (3, 3, 128), (1, 1, 256))):
        x = torch.randn((B, *shape_in), generator=g, device="cuda") * 1.5
        want, _, _ = d._dconv(x, layer, B, shape_in[0], shape_in[1], counts=counts, cap=cap)
        got = d._gemm_split(lid, x, B, shape_out, counts, cap)
        torch.cuda.synchronize()
        assert got.shape == want.shape == (B, *shape_out)
        err, scale = float((got[valid] - want[valid]).abs().max()), float(want[valid].abs().max())
        print(f"\nlayer {lid}: split GEMM vs f32 layer: max |d| {err:.3e} at scale {scale:.3f}")
        assert err <= 4e-6 * scale, (lid, err, scale)


def test_pnet_conv1_on_matrix_cores_with_exact_tiles_under_the_band():
    """``split_pconv1`` (batches, band mode): P-Net conv1 runs on the f16 matrix cores too (split precision, map written only as
    hi/lo halves); the exact pass then needs exact f32 conv1 values under its cells' 5x5 windows, which the f32 conv1 kernel
    produces for just the 16x64 tiles those windows touch (fr_pnet_band_tiles -> fr_pnet_conv1_band mode 1).  Checked here:
    (1) the split map decodes to the f32 map within 2e-6 relative; (2) per level the candidate lists of the two settings agree -
    same counts, same cells, the cells inside the band bit for bit; (3) the marked tiles cover every band cell's window and the
    f32 map holds the f32 kernel's bits there; (4) the cascade on eight frames returns the same faces."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from make_golden import synth_frame
    from facerecognition_infrenceengine_amd import weights, _lib
    from facerecognition_infrenceengine_amd.mtcnn import MTCNNHIP, pyramid_scales
    st = weights.synth_mtcnn_states(seed=4321)
    mc = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0)
    ref = MTCNNHIP(*st, device="cuda:0", batch_min_pixels=0)
    assert mc.split_pconv1 and mc.pnet_band
    mc.split_pconv1_min_px = 25                                         # every level (the default takes the large ones only)
    ref.split_pconv1 = False
    mc.split_ro = ref.split_ro = False                                  # this test isolates the P-Net
    hw = (360, 640)
    N = 8
    fr = torch.from_numpy(np.ascontiguousarray(np.stack([synth_frame(hw[0], hw[1], 170 + k) for k in range(N)]))).cuda()
    lib = mc.lib
    # (1) the split map against the f32 map of the same kernel family
    hs, ws = 180, 320
    h, w = mc.p1.out_hw(hs, ws)
    xs = torch.zeros(N, h, w, 64, dtype=torch.uint8, device="cuda")
    y = torch.zeros(N, h, w, 12, device="cuda")
    s0 = torch.cuda.current_stream().cuda_stream
    lib.fr_pnet_conv1_band(0, _lib.ptr(fr), N, hw[0], hw[1], hs, ws, _lib.ptr(mc.p1.w), _lib.ptr(mc.p1.b), _lib.ptr(mc.p1.slope),
                           _lib.ptr(y), _lib.ptr(xs), None, None, 0, s0)
    with torch.cuda.device("cuda:0"):
        ref._s = s0
        yr, _, _ = ref._dconv(None, ref.p1, N, hs, ws, frames=fr)
    torch.cuda.synchronize()
    hl = xs.view(torch.float16).reshape(N, h, w, 2, 16)[..., :12].float()
    dec = hl[..., 0, :] + hl[..., 1, :]
    scale = float(yr.abs().max())
    assert float((dec - yr).abs().max()) <= 4e-6 * scale, (float((dec - yr).abs().max()), scale)
    assert float((y - yr).abs().max()) <= 4e-6 * scale
    # (2), (3) level by level
    t0 = mc.thresholds[0]
    lt = float(np.log(t0 / (1 - t0)))
    cs = mc.cap_scale
    inband = total = tiles_marked = tiles_all = 0
    for s in pyramid_scales(*hw):
        outs = []
        for d in (mc, ref):
            lb, ls, lr, lc = (torch.zeros(N, cs, 4, device="cuda"), torch.zeros(N, cs, device="cuda"),
                              torch.zeros(N, cs, 4, device="cuda"), torch.zeros(N, dtype=torch.int32, device="cuda"))
            with torch.cuda.device("cuda:0"):
                d._s = s0
                head, hc, wc = d.pnet_level(fr, s, cand=(float(s), t0, cs, lb, ls, lr, lc))
                if d is ref:
                    assert not d._tls.level_done
                    bc = torch.zeros(N * (-(-hc * wc // 256)), dtype=torch.int32, device="cuda")
                    lib.fr_pnet_candidates(_lib.ptr(head), N, hc, wc, float(s), t0, cs, _lib.ptr(lb), _lib.ptr(ls), _lib.ptr(lr),
                                           _lib.ptr(lc), _lib.ptr(bc), None, _lib.ptr(d._dl[0]), d._dl[1], s0)
                else:
                    assert d._tls.level_done
                    x1, _, _, wsp, tbuf, tiles, _ = d._tls.keep
            torch.cuda.synchronize()
            outs.append((lb, ls, lr, lc, head, d._dl[0][:N * hc * wc].reshape(N, hc, wc).clone()))
        (ab, as_, ar, ac, ah, adl), (bb, bs, br, bc_, bh, bdl) = outs
        assert torch.equal(ac, bc_), (s, ac, bc_)
        for f in range(N):
            n = min(int(ac[f]), cs)
            assert torch.equal(ab[f, :n], bb[f, :n])                      # the same cells in the same order
            sb = bs[f, :n]
            lg = torch.log(sb / (1 - sb))
            near = (lg - lt).abs() < 0.5 * mc.refine_margin              # well inside the band by the f32 path's own score
            assert torch.equal(as_[f, :n][near], sb[near]) and torch.equal(ar[f, :n][near], br[f, :n][near])
            assert float((as_[f, :n] - sb).abs().max() if n else 0.0) <= 5e-6
            assert float((ar[f, :n] - br[f, :n]).abs().max() if n else 0.0) <= 2e-5
            inband += int(near.sum()); total += n
        # (3) band cells of the matrix-core run: their windows' tiles are on the list and hold the f32 kernel's bits
        h1, w1 = hc + 4, wc + 4
        band = (adl >= lt - mc.refine_margin) & (adl <= lt + mc.refine_margin)
        cnt = int(tbuf[0])
        nt = tiles.numel()
        tiles_marked += cnt; tiles_all += nt
        listed = set(tiles[:cnt].tolist())
        assert len(listed) == cnt
        ry, rx = (h1 + 7) // 8, (w1 + 31) // 32
        hs_l, ws_l = int(np.ceil(hw[0] * s)), int(np.ceil(hw[1] * s))
        with torch.cuda.device("cuda:0"):
            yr, _, _ = ref._dconv(None, ref.p1, N, hs_l, ws_l, frames=fr)
        torch.cuda.synchronize()
        for n_, yy, xx in band.nonzero().tolist():
            for ty in {yy // 8, (yy + 4) // 8}:
                for tx in {xx // 32, (xx + 4) // 32}:
                    assert (n_ * ry + ty) * rx + tx in listed
            assert torch.equal(x1[n_, yy:yy + 5, xx:xx + 5], yr[n_, yy:yy + 5, xx:xx + 5])
    assert total >= 100 and inband >= 1, (total, inband)
    # (4) the cascade
    a = mc.detect_batch(fr)
    b = ref.detect_batch(fr)
    torch.cuda.synchronize()
    print(f"\nP-Net conv1 on the matrix cores: {total} candidates, {inband} inside the band bit-identical; "
          f"{tiles_marked} of {tiles_all} conv1 tiles recomputed in f32")
    assert torch.equal(a[3], b[3]) and int(a[3].sum()) >= 8
    for f in range(N):
        n = int(a[3][f])
        if n:
            assert float((a[1][f, :n] - b[1][f, :n]).abs().max()) <= 5e-6
            assert float((a[0][f, :n] - b[0][f, :n]).abs().max()) <= 1e-3
            assert float((a[2][f, :n] - b[2][f, :n]).abs().max()) <= 1e-3
