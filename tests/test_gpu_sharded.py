"""Sharded match on ONE device (last row): R row shards of the gallery scanned by the HIP
kernels with a per-shard ``row_offset``, candidates packed and reduced by the HIP exchange kernels - ids must be
identical to the unsharded oracle for R in {1, 2, 4, 8}, including duplicate rows in different shards, an empty
shard and a rank with zero faces.  This is exactly what each rank of the N-GPU job computes; only the two
all-gathers (byte moves) are replaced by torch.cat here."""
import numpy as np
import pytest
import torch

from oracle import match as omatch

pytestmark = pytest.mark.gpu


def _simulate(G, Qs, q_max, scan):
    """Every 'rank' r holds shard r and the queries Qs[r]; returns per-rank (idx, score) numpy."""
    from facerecognition_infrenceengine_amd.distributed import (HipOps, gathered_counts, pack_candidates, pack_queries,
                                                                reduce_packed, shard_rows)
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    R, N = len(Qs), len(G)
    ops = []
    for r in range(R):
        lo, hi = shard_rows(N, R, r)
        m = GalleryMatcher("cuda:0", scan=scan)
        m.set_rows(range(lo, hi), G[lo:hi], normalise=False)
        ops.append(HipOps(m, lo))
    # step 0/1: renormalise own rows, pack (rows + zero padding + count row: fr_exchange_pack_queries), "all-gather"
    seg = q_max + 1
    sends = []
    for r in range(R):
        Qn = ops[r].renormalise(torch.from_numpy(Qs[r]).cuda())
        send = ops[r].pack_queries(Qn, q_max)
        assert torch.equal(send, pack_queries(Qn, q_max))                # HIP exchange glue == torch forms, bit for bit
        sends.append(send)
    allq = torch.cat(sends)
    cnt = ops[0].gathered_counts(allq, R, q_max)
    assert torch.equal(cnt, gathered_counts(allq, R, q_max)) and cnt.tolist() == [len(q) for q in Qs]
    # step 2: every rank scans its shard for all gathered slots IN PLACE (a segment = q_max slots + the count row, which
    # is padding); step 3: pack, "all-gather", reduce own slots
    packs = []
    for r in range(R):
        idx, score = ops[r].scan(allq, counts=cnt, seg_len=seg)
        assert int(idx.view(R, seg)[:, q_max].max()) == -1               # the count row is never a query
        p = ops[r].pack(idx, score)
        assert torch.equal(p, pack_candidates(idx, score))               # HIP pack == torch pack, bit for bit
        packs.append(p)
    allp = torch.cat(packs)
    out = []
    for r in range(R):
        F = len(Qs[r])
        bi, bs = ops[r].reduce(allp, R, R * seg, r * seg, F)
        ti, ts = reduce_packed(allp, R, R * seg, r * seg, F)             # HIP reduce == torch reduce
        assert torch.equal(bi, ti) and torch.equal(bs, ts)
        out.append((bi.cpu().numpy(), bs.cpu().numpy()))
    return out


@pytest.mark.parametrize("scan", ["f32", "f16", "f8"])
@pytest.mark.parametrize("R", [1, 2, 4, 8])
def test_sharded_hip_match_equals_unsharded_oracle(R, scan):
    rng = np.random.default_rng(100 + R)
    N, q_max = 5003, 16
    G = rng.standard_normal((N, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    fs = [(7 * r + 5) % (q_max + 1) for r in range(R)]
    if R > 1:
        fs[1] = 0                                                         # a rank with no faces this step
    Qs = []
    for f in fs:
        Q = rng.standard_normal((f, 512)).astype(np.float32)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True) * rng.uniform(0.5, 2.0, (f, 1)).astype(np.float32)  # not unit: a-6 must run
        Qs.append(Q)
    q0 = Qs[0][0] / np.linalg.norm(Qs[0][0])
    G[7] = q0; G[N - 3] = q0; G[N // 2] = q0                              # duplicates in DIFFERENT shards
    out = _simulate(G, Qs, q_max, scan)
    for r in range(R):
        if fs[r] == 0:
            assert len(out[r][0]) == 0
            continue
        Qn = np.stack([omatch.renormalise(q) for q in Qs[r]])
        oi, os_ = omatch.match_rows_fast(Qn, G)
        assert np.array_equal(out[r][0], oi), (r, out[r][0], oi)
        np.testing.assert_allclose(out[r][1], os_, atol=3e-6)
    assert out[0][0][0] == 7                                              # lowest global row among the duplicates


@pytest.mark.parametrize("scan", ["f32", "f16"])
def test_sharded_with_empty_shards_and_empty_gallery(scan):
    rng = np.random.default_rng(5)
    Q = rng.standard_normal((3, 512)).astype(np.float32)
    G = rng.standard_normal((2, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    out = _simulate(G, [Q, Q[:1], Q[:0], Q[:2]], 4, scan)                 # 2 rows over 4 ranks: two empty shards
    Qn = np.stack([omatch.renormalise(q) for q in Q])
    oi, _ = omatch.match_rows_fast(Qn, G)
    assert np.array_equal(out[0][0], oi) and np.array_equal(out[3][0], oi[:2])
    out = _simulate(G[:0], [Q, Q[:1]], 4, scan)                           # empty gallery: (-1, -1) everywhere
    assert (out[0][0] == -1).all() and (out[0][1] == -1).all() and (out[1][0] == -1).all()
