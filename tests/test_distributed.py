"""world_size-2 gloo test of the frame-sharded / gallery-row-sharded match exchange (host logic);
the local scan is the oracle here (the product wires the HIP scan into the same class)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import match as omatch


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class OracleOps:
    """CPU stand-in for distributed.HipOps: the oracle's arithmetic behind the same six methods, so the exchange
    logic (padding, counts, packing, reduce over shards) is what these gloo tests exercise."""

    def __init__(self, shard, lo):
        self.shard, self.lo = shard, lo

    def renormalise(self, Q):
        q = Q.numpy()
        return torch.from_numpy(np.stack([omatch.renormalise(r) for r in q]).astype(np.float32)) if len(q) else Q.clone()

    def scan(self, Q, counts=None, seg_len=0):
        q = Q.numpy().copy()
        if counts is not None:                                    # padding slots (incl. the count row) are never scanned
            pad = (np.arange(len(q)) % seg_len) >= counts.numpy()[np.arange(len(q)) // seg_len]
            q[pad] = 0
        assert not np.isnan(q).any(), "padding rows must not be divided by their norm"
        idx, score = omatch.match_rows_fast(q, self.shard) if len(self.shard) else (
            np.full(len(q), -1, np.int64), np.full(len(q), -1, np.float32))
        idx = np.where(idx >= 0, idx + self.lo, -1)
        if counts is not None:                                    # padding slots report (-1, -1), as the HIP scan
            pad = (np.arange(len(q)) % seg_len) >= counts.numpy()[np.arange(len(q)) // seg_len]
            idx[pad], score = -1, np.where(pad, np.float32(-1), score)
        return torch.from_numpy(idx), torch.from_numpy(score.astype(np.float32))

    def pack_queries(self, Qn, q_max):
        from facerecognition_infrenceengine_amd.distributed import pack_queries
        return pack_queries(Qn, q_max)

    def gathered_counts(self, allq, world, q_max):
        from facerecognition_infrenceengine_amd.distributed import gathered_counts
        return gathered_counts(allq, world, q_max)

    def pack(self, idx, score):
        from facerecognition_infrenceengine_amd.distributed import pack_candidates
        return pack_candidates(idx, score)

    def reduce(self, allp, world, n, q0, F):
        from facerecognition_infrenceengine_amd.distributed import reduce_packed
        return reduce_packed(allp, world, n, q0, F)


def _worker(rank, world, port, G, Qs, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from facerecognition_infrenceengine_amd.distributed import ShardedGalleryMatcher, shard_rows
    lo, hi = shard_rows(len(G), world, rank)
    shard = G[lo:hi]

    m = ShardedGalleryMatcher(OracleOps(shard, lo), q_max=8)
    idx, score = m.match(torch.from_numpy(Qs[rank]))
    out[rank] = (idx.numpy(), score.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run(world, N, fs):
    rng = np.random.default_rng(42 + N)
    G = rng.standard_normal((N, 512)).astype(np.float32)
    if N:
        G /= np.linalg.norm(G, axis=1, keepdims=True)
    Qs = []
    for f in fs:
        Q = rng.standard_normal((f, 512)).astype(np.float32)
        Q /= np.linalg.norm(Q, axis=1, keepdims=True)
        Qs.append(Q)
    if N > 40:
        G[3] = Qs[0][0]; G[N - 2] = Qs[0][0]          # duplicate rows in DIFFERENT shards: lowest row wins
        G[N - 5] = Qs[1][1] if fs[1] > 1 else G[N - 5]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), G, Qs, out), nprocs=world, join=True)
    for r in range(world):
        idx, score = out[r]
        if N == 0:
            assert (idx == -1).all()
            continue
        oi, os_ = omatch.match_rows_fast(Qs[r], G)
        assert np.array_equal(idx, oi), (r, idx, oi)
        np.testing.assert_allclose(score, os_, atol=1e-6)
    return out


def test_world2_sharded_match_equals_unsharded():
    out = _run(2, 101, [5, 3])
    assert out[0][0][0] == 3


def test_world2_ragged_and_empty_rank():
    _run(2, 64, [8, 0])        # one rank has no faces this step
    _run(2, 1, [2, 2])         # one shard is empty (1 row over 2 ranks)


def test_world4_matches_unsharded():
    _run(4, 257, [1, 8, 0, 4])


def test_world8_matches_unsharded():
    """the driver's largest run: 8 ranks, a rank without faces, ragged counts, 5 rows spread over 8 shards"""
    _run(8, 1001, [3, 0, 8, 1, 5, 2, 7, 4])
    _run(8, 5, [1, 1, 0, 2, 1, 0, 1, 1])          # three ranks hold an empty gallery shard


def test_shard_rows_partition():
    from facerecognition_infrenceengine_amd.distributed import shard_rows
    for n in (0, 1, 7, 10000, 1000003):
        for w in (1, 2, 4, 8):
            edges = [shard_rows(n, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_reduce_candidates_tie_rule():
    from facerecognition_infrenceengine_amd.distributed import reduce_candidates
    s = torch.tensor([[0.5, 0.9, -1.0], [0.5, 0.2, -1.0]])
    i = torch.tensor([[40, 7, -1], [12, 99, -1]])
    bi, bs = reduce_candidates(s, i)
    assert bi.tolist() == [12, 7, -1]
    assert torch.equal(bs, torch.tensor([0.5, 0.9, -1.0]))


def test_bench_refuses_ranks_that_share_a_device():
    """bench.py gathers {rank, device_index, pci_bus_id, uuid} of every rank after the timed region and exits (code 5) when two
    ranks report the same device: the first real multi-GPU record must prove that its N ranks sat on N GPUs.
    The rule itself, on gathered records as an 8-GPU node and a mis-launched job would produce them."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    good = [{"rank": r, "device_index": r, "pci_bus_id": f"0000:{5 + 16 * r:02x}:00", "uuid": f"GPU-{r:04d}"} for r in range(8)]
    assert bench.ranks_sharing_a_device(good) == []
    bad = [dict(d) for d in good]
    bad[5]["pci_bus_id"] = bad[2]["pci_bus_id"]            # LOCAL_RANK ignored: rank 5 landed on rank 2's GPU
    assert bench.ranks_sharing_a_device(bad) == [(2, 5)]
    no_pci = [{"rank": r, "device_index": 0, "uuid": "GPU-same"} for r in range(2)]
    assert bench.ranks_sharing_a_device(no_pci) == [(0, 1)]
    bare = [{"rank": 0, "device_index": 0}, {"rank": 1, "device_index": 1}, {"rank": 2, "device_index": 1}]
    assert bench.ranks_sharing_a_device(bare) == [(1, 2)]
