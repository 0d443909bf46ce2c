"""Multi-GPU match: frames are sharded across ranks (one process per GPU), the gallery is
row-sharded, and one exchange step joins them.

The reference's only parallelism is one OS process per camera, each scanning the whole gallery
(/root/reference/infrenceServer.py:606,641-646).  Here:

  1. all-gather of the per-rank query rows, padded to ``q_max`` with a count  (RCCL over xGMI;
     payload is KBs, so it is latency-bound: one collective, no bucketing);
  2. every rank scans ITS gallery shard for ALL gathered queries (one HBM pass over the shard);
  3. all-gather of the per-shard (score, global row) pairs; each rank reduces the candidates of
     its own queries: maximum score, lowest global row on exact ties - the same rule as the
     single-GPU scan (strict '>' in /root/reference/infrenceServer.py:538-542).

``local_scan(Q[F,512]) -> (idx int64[F] global rows or -1, score f32[F])`` is injected: the
product passes the HIP scan (GalleryMatcher.match_device with row_offset), the CPU ``gloo`` tests
pass the oracle.  This module contains no arithmetic besides the final max/tie reduce.
"""
import torch
import torch.distributed as dist


def shard_rows(n_rows, world_size, rank):
    """Contiguous row shard [lo, hi) of rank; the first n_rows % world ranks get one extra row."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_candidates(scores, idx):
    """scores f32 [R,F], idx i64 [R,F] (global rows, -1 = none) -> best (idx[F], score[F]):
    maximum score, lowest row index on exact ties; (-1, -1.0) when no shard has a candidate."""
    valid = idx >= 0
    s = torch.where(valid, scores, torch.full_like(scores, float("-inf")))
    best = s.max(dim=0).values
    big = torch.iinfo(torch.int64).max
    cand = torch.where(valid & (s == best[None, :]), idx, torch.full_like(idx, big))
    bi = cand.min(dim=0).values
    none = bi == big
    return torch.where(none, torch.full_like(bi, -1), bi), torch.where(none, torch.full_like(best, -1.0), best)


class ShardedGalleryMatcher:
    def __init__(self, local_scan, q_max, dim=512, group=None, force_exchange=False):
        """``force_exchange``: run both collectives even with one rank (rehearses the RCCL path on a 1-GPU box)."""
        self.local_scan, self.q_max, self.dim, self.group = local_scan, q_max, dim, group
        self.force_exchange = force_exchange
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def match(self, Q):
        """Q f32 [F_local,dim] on this rank's device (F_local <= q_max).
        Returns (idx i64[F_local] global rows, score f32[F_local]) for the local queries."""
        F = Q.shape[0]
        assert F <= self.q_max, "more local queries than q_max"
        if self.world == 1 and not self.force_exchange:
            return self.local_scan(Q)
        dev = Q.device
        # (1) gather queries; the count rides in an extra row so it stays ONE collective
        send = torch.zeros((self.q_max + 1, self.dim), dtype=torch.float32, device=dev)
        send[:F] = Q
        send[self.q_max:, :1].fill_(float(F))     # (a scalar __setitem__ copies from the host and SYNCHRONISES)
        allq = torch.empty((self.world * (self.q_max + 1), self.dim), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allq, send, group=self.group)      # concatenated along dim 0
        allq = allq.view(self.world, self.q_max + 1, self.dim)
        # (2) scan the local shard for every gathered query (padding rows included: fixed shape)
        flat = allq[:, :self.q_max].reshape(self.world * self.q_max, self.dim)
        idx, score = self.local_scan(flat)
        # (3) gather the per-shard candidates and reduce those of the local queries.  (score, row) travel as raw
        # bits in ONE int32 [n,3] tensor: bit copies only, no float conversion kernels in the exchange
        n = score.shape[0]
        pair = torch.empty((n, 3), dtype=torch.int32, device=dev)
        pair[:, 0] = score.contiguous().view(torch.int32)
        pair[:, 1:] = idx.contiguous().view(torch.int32).view(n, 2)
        allp = torch.empty((self.world * n, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allp, pair, group=self.group)
        mine = allp.view(self.world, n, 3)[:, self.rank * self.q_max:self.rank * self.q_max + F]
        sc = torch.empty((self.world, F), dtype=torch.int32, device=dev).copy_(mine[..., 0]).view(torch.float32)
        ix = torch.empty((self.world, F, 2), dtype=torch.int32, device=dev).copy_(mine[..., 1:]).view(torch.int64)
        ix = ix.reshape(self.world, F)
        return reduce_candidates(sc, ix)
