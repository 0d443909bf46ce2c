"""GPU parity: gallery match through the C ABI vs the reference-pinned oracle."""
import numpy as np
import pytest
import torch

from oracle import match as omatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    return GalleryMatcher("cuda:0")


@pytest.mark.parametrize("tag", ["g100", "g1000"])
def test_match_golden_ids_and_decisions(matcher, golden, tag):
    d = golden("match_kat.npz")
    G, Q = d[f"{tag}_G"], d[f"{tag}_Q"]
    matcher.set_rows(list(range(len(G))), G, normalise=False)
    ids, score, idx = matcher.match(Q, thr=0.4)
    exp = d[f"{tag}_live_pid"]
    got = np.asarray([-1 if i is None else i for i in ids])
    assert np.array_equal(got, exp)                      # identical top-1 ids + >= 0.4 decisions
    oi, os_ = omatch.match_rows(Q, G)
    assert np.array_equal(idx, oi)
    np.testing.assert_allclose(score, os_, atol=2e-6)    # f32 dot, different summation order
    assert idx[2] == 17                                   # exact tie 17/63 -> lowest row
    # counting-path band (peopleCount.py:876-887)
    ids2, _, _ = matcher.match(Q, thr=0.45, unknown_thr=0.35)
    rec = [i for i in ids2 if i is not None]
    assert rec == list(d[f"{tag}_count_rec_pid"])


def test_l2norm_rows_matches_numpy(lib):
    from facerecognition_infrenceengine_amd import _lib
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((37, 512)) * 3).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    out = torch.empty_like(xd)
    lib.fr_l2norm_rows_f32(_lib.ptr(xd), _lib.ptr(out), 37, 512, _lib.stream_ptr())
    ref = x / np.linalg.norm(x, axis=1, keepdims=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=3e-7, atol=1e-8)


@pytest.mark.parametrize("N,F", [(0, 3), (1, 1), (31, 5), (33, 40), (4097, 70), (10000, 256)])
def test_match_ragged_sizes(matcher, N, F):
    rng = np.random.default_rng(N + F)
    G = rng.standard_normal((N, 512)).astype(np.float32)
    if N:
        G /= np.linalg.norm(G, axis=1, keepdims=True)
    Q = rng.standard_normal((F, 512)).astype(np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if N > 3:
        G[N - 1] = Q[0]; G[N // 2] = Q[0]                 # duplicate rows: lowest index wins
    matcher.set_rows(list(range(N)), G, normalise=False)
    idx, score = matcher.match_device(torch.from_numpy(Q).cuda())
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    if N == 0:
        assert (idx == -1).all() and (score == -1).all()
        return
    oi, os_ = omatch.match_rows_fast(Q, G)
    assert np.array_equal(idx, oi)
    np.testing.assert_allclose(score, os_, atol=3e-6)
    if N > 3:
        assert idx[0] == N // 2


def test_full_size_property_planted_rows(matcher):
    """BASELINE config sizes (10k and 1M rows): every query has a planted row; ids must be
    exactly the planted rows (size-independent property, no CPU oracle at this size)."""
    g = torch.Generator(device="cuda").manual_seed(1)
    for N in (10_000, 1_000_000):
        G = torch.randn((N, 512), generator=g, device="cuda")
        G /= G.norm(dim=1, keepdim=True)
        rows = torch.randperm(N, generator=g, device="cuda")[:256]
        Q = G[rows] + 0.02 * torch.randn((256, 512), generator=g, device="cuda")
        matcher.set_rows(range(N), G, normalise=False)
        idx, score = matcher.match_device(Q)
        assert torch.equal(idx, rows)
        assert float(score.min()) > 0.8


@pytest.mark.parametrize("scan", ["f16", "f8"])
@pytest.mark.parametrize("N,F", [(1, 1), (33, 5), (63, 300), (4097, 130), (100_000, 256), (20_000, 700)])
def test_coarse_scan_with_f32_rerank_matches_f32_oracle(N, F, scan):
    """f16 / fp8 one-pass GEMM scan + exact f32 re-rank must give the SAME ids as the f32 oracle (incl. duplicate
    rows, a near-duplicate, row tiles that end inside a 64-row tile and query tiles that end inside a wave)."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    rng = np.random.default_rng(N * 7 + F)
    G = rng.standard_normal((N, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    Q = rng.standard_normal((F, 512)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if N > 3:
        G[N - 1] = Q[0]; G[N // 2] = Q[0]
        Q[1 % F] = G[N // 3] + 0.01 * rng.standard_normal(512).astype(np.float32)      # near-duplicate of one row
    m = GalleryMatcher("cuda:0", scan=scan)
    m.set_rows(list(range(N)), G, normalise=False)
    idx, score = m.match_device(torch.from_numpy(Q).cuda())
    oi, os_ = omatch.match_rows_fast(Q, G)
    assert np.array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_allclose(score.cpu().numpy(), os_, atol=3e-6)                     # scores are the f32 re-scores
    if N > 3:
        assert int(idx[0]) == N // 2                                                    # duplicate rows: lowest row


def test_fp8_scan_saturates_instead_of_nan_on_non_unit_rows():
    """fp8 rows are e4m3(256 g): an element beyond 1.75 leaves the e4m3 range, where the bare conversion gives NaN and
    a NaN coarse score fails every '>' - the row would never reach the exact re-rank even when it is the true best
    match.  The packing saturates at +-448 instead: a non-unit gallery row (set_rows(normalise=False)) and a
    non-unit query (renormalise=False) still come back with the f32 oracle's ids."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    rng = np.random.default_rng(31)
    N = 3000
    G = rng.standard_normal((N, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    Q = rng.standard_normal((6, 512)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    G[1234] = 3.0 * Q[0]                        # not unit: the true best row of query 0, elements up to ~0.5 ... fine,
    G[1234, 7] = 2.5                            # ... and one element past 1.75: 640 > 448 in the fp8 copy
    Q[1] = 4.0 * G[77]; Q[1, 3] = 2.2           # a non-unit query with an out-of-range element
    m = GalleryMatcher("cuda:0", scan="f8")
    m.set_rows(list(range(N)), G, normalise=False)
    idx, score = m.match_device(torch.from_numpy(Q).cuda(), renormalise=False)
    S = Q @ G.T                                 # renormalise=False: raw dots of the non-unit query
    oi = S.argmax(axis=1)
    assert np.array_equal(idx.cpu().numpy(), oi) and int(oi[0]) == 1234 and int(oi[1]) == 77
    np.testing.assert_allclose(score.cpu().numpy(), S[np.arange(6), oi], rtol=2e-6, atol=3e-6)
    assert not torch.isnan(score).any()


@pytest.mark.parametrize("scan,N", [("f16", 1_000_000), ("f8", 1_250_000)])
def test_coarse_scan_full_size(scan, N):
    """C4's whole 1 M-row gallery (f16) and one of C5's 10 M / 8 = 1.25 M-row fp8 shards, 2048 gathered queries
    (8 ranks x 256): planted rows must come back exactly (size-independent property), and for UNPLANTED queries -
    whose best rows are a near-tie of random rows, the hard case for a coarse scan - the ids must equal those of
    the exact f32 HIP scan (itself pinned to the oracle at smaller sizes) and of numpy for a few of them."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    g = torch.Generator(device="cuda").manual_seed(3)
    G = torch.randn((N, 512), generator=g, device="cuda"); G /= G.norm(dim=1, keepdim=True)
    rows = torch.randperm(N, generator=g, device="cuda")[:2048]                        # 8 ranks x 256 gathered queries
    Q = G[rows] + 0.02 * torch.randn((2048, 512), generator=g, device="cuda")
    Q[1024:] = torch.randn((1024, 512), generator=g, device="cuda")                    # unknown faces: no planted row
    m = GalleryMatcher("cuda:0", scan=scan)
    m.set_rows(range(N), G, normalise=False)
    idx, score = m.match_device(Q)
    assert torch.equal(idx[:1024], rows[:1024]) and float(score[:1024].min()) > 0.8
    exact = GalleryMatcher("cuda:0")
    exact.G, exact.ids = m.G, m.ids                                                    # same f32 rows, f32 scan
    ei, es = exact.match_device(Q[1024:1024 + 256])
    assert torch.equal(idx[1024:1024 + 256], ei)
    torch.testing.assert_close(score[1024:1024 + 256], es, atol=3e-6, rtol=0)
    Qn = Q[1024:1032] / Q[1024:1032].norm(dim=1, keepdim=True)
    ref = (G @ Qn.T).argmax(dim=0)
    assert torch.equal(idx[1024:1032], ref)


@pytest.mark.parametrize("scan", ["f32", "f16", "f8"])
def test_padding_slots_are_skipped(scan):
    """Gathered batch of the sharded match: 4 segments of 8 slots with counts (3, 0, 8, 1); real slots equal the
    unmasked result, padding slots report (-1, -1) whatever their rows hold (here: NaN)."""
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    rng = np.random.default_rng(11)
    G = rng.standard_normal((3000, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    Q = rng.standard_normal((32, 512)).astype(np.float32); Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    counts = np.array([3, 0, 8, 1], np.int32)
    real = (np.arange(32) % 8) < counts[np.arange(32) // 8]
    Qp = Q.copy(); Qp[~real] = np.nan
    m = GalleryMatcher("cuda:0", scan=scan)
    m.set_rows(list(range(3000)), G, normalise=False)
    idx, score = m.match_device(torch.from_numpy(Qp).cuda(), renormalise=False, row_offset=500,
                                counts=torch.from_numpy(counts).cuda(), seg_len=8)
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    oi, os_ = omatch.match_rows_fast(Q, G)
    assert np.array_equal(idx[real], oi[real] + 500)
    np.testing.assert_allclose(score[real], os_[real], atol=3e-6)
    assert (idx[~real] == -1).all() and (score[~real] == -1).all()


def test_two_matches_in_flight_on_two_streams(matcher):
    """One matcher driven from two HIP streams at once (bench.py --pipes 2): each call owns its scratch, so both
    results equal their solo runs (a shared per-object workspace let step i's reduce read step i+1's partials)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    N = 300_000
    G = torch.randn((N, 512), generator=g, device="cuda"); G /= G.norm(dim=1, keepdim=True)
    matcher.set_rows(range(N), G, normalise=False)
    Qa = torch.randn((256, 512), generator=g, device="cuda")
    Qb = torch.randn((40, 512), generator=g, device="cuda")
    solo_a, solo_b = matcher.match_device(Qa), matcher.match_device(Qb)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(4):
        with torch.cuda.stream(sa):
            ra = matcher.match_device(Qa)
        with torch.cuda.stream(sb):
            rb = matcher.match_device(Qb)
        torch.cuda.synchronize()
        assert torch.equal(ra[0], solo_a[0]) and torch.equal(ra[1], solo_a[1])
        assert torch.equal(rb[0], solo_b[0]) and torch.equal(rb[1], solo_b[1])
