"""Multi-GPU match: frames are sharded across ranks (one process per GPU), the gallery is
row-sharded, and one exchange step joins them.

The reference's only parallelism is one OS process per camera, each scanning the whole gallery
(/root/reference/infrenceServer.py:606,641-646).  Here, per step:

  0. every rank re-normalises ITS OWN query rows (a-6, /root/reference/infrenceServer.py:532) - before the
     exchange, so zero padding rows are never divided by their norm;
  1. ONE all-gather of the per-rank query rows, padded to ``q_max``, the count riding in an extra row
     (RCCL over xGMI; the payload is KBs, so it is latency-bound: one collective, no bucketing);
  2. every rank scans ITS gallery shard for the gathered queries (one HBM pass over the shard); the gathered
     counts tell the scan which query slots are padding (``counts`` / ``seg_len`` of the scan ops);
  3. ONE all-gather of the per-shard candidates, packed as int32 (score bits, row lo, row hi); each rank
     reduces the candidates of its own queries: maximum score, lowest global row on exact ties - the same
     rule as the single-GPU scan (strict '>' in /root/reference/infrenceServer.py:538-542).

The arithmetic is injected as an ``ops`` object: ``HipOps`` (the product: libfrhip.so kernels through
``GalleryMatcher``; fails without a HIP device) or, in the CPU ``gloo`` tests, an oracle-backed stand-in with
the same six methods.  This module itself only moves bytes: it launches no torch kernel between the embedder's
output and the ids (zeros / slice-assign / fill_ / casts used to run on the embed stream beside the other pipe's convs).

Query rows travel as f32 (2 KB/row), not f16: the scan's final scores are exact f32 dots of the SAME f32
query the single-GPU path uses, which is what keeps top-1 ids bit-identical to the CPU loop; at 256 faces
per rank the collective is 512 KB, far below where xGMI bandwidth (not latency) would matter.
"""
import torch
import torch.distributed as dist


def shard_rows(n_rows, world_size, rank):
    """Contiguous row shard [lo, hi) of rank; the first n_rows % world ranks get one extra row."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_candidates(idx, score):
    """(idx i64[n], score f32[n]) -> int32 [n,3] = (score bits, row lo, row hi).  Torch form of
    fr_match_pack_candidates: used by the CPU gloo tests and as the cross-check of the kernel."""
    n = score.shape[0]
    pair = torch.empty((n, 3), dtype=torch.int32, device=score.device)
    pair[:, 0] = score.contiguous().view(torch.int32)
    pair[:, 1:] = idx.contiguous().view(torch.int32).view(n, 2)
    return pair


def pack_queries(Qn, q_max):
    """Qn f32 [F,D] -> send f32 [q_max+1, D]: the rows, zero rows up to q_max, and a last row whose first element is
    F.  Torch form of fr_exchange_pack_queries (CPU gloo tests, cross-check of the kernel)."""
    F, D = Qn.shape
    send = torch.zeros((q_max + 1, D), dtype=torch.float32, device=Qn.device)
    send[:F] = Qn
    send[q_max:, :1].fill_(float(F))          # (a scalar __setitem__ copies from the host and SYNCHRONISES)
    return send


def gathered_counts(allq, world, q_max):
    """allq f32 [world*(q_max+1), D] -> int32 [world].  Torch form of fr_exchange_counts."""
    return allq.view(world, q_max + 1, -1)[:, q_max, 0].to(torch.int32).clamp(0, q_max).contiguous()


def reduce_candidates(scores, idx):
    """scores f32 [R,F], idx i64 [R,F] (global rows, -1 = none) -> best (idx[F], score[F]):
    maximum score, lowest row index on exact ties; (-1, -1.0) when no shard has a candidate.
    Torch form of fr_match_reduce_shards (CPU gloo tests, cross-check of the kernel)."""
    valid = idx >= 0
    s = torch.where(valid, scores, torch.full_like(scores, float("-inf")))
    best = s.max(dim=0).values
    big = torch.iinfo(torch.int64).max
    cand = torch.where(valid & (s == best[None, :]), idx, torch.full_like(idx, big))
    bi = cand.min(dim=0).values
    none = bi == big
    return torch.where(none, torch.full_like(bi, -1), bi), torch.where(none, torch.full_like(best, -1.0), best)


def reduce_packed(allp, world, n, q0, F):
    """Torch reduce of gathered packed candidates int32 [world*n,3] for queries [q0, q0+F)."""
    mine = allp.view(world, n, 3)[:, q0:q0 + F]
    sc = torch.empty((world, F), dtype=torch.int32, device=allp.device).copy_(mine[..., 0]).view(torch.float32)
    ix = torch.empty((world, F, 2), dtype=torch.int32, device=allp.device).copy_(mine[..., 1:]).view(torch.int64)
    ix = ix.reshape(world, F)
    return reduce_candidates(sc, ix)


class HipOps:
    """The product's arithmetic for the exchange: libfrhip.so kernels on this rank's GPU."""

    def __init__(self, matcher, row_lo):
        from . import _lib
        self._lib, self.lib = _lib, matcher.lib
        self.matcher, self.row_lo, self.device = matcher, int(row_lo), matcher.device

    def renormalise(self, Q):
        out = torch.empty_like(Q)
        if Q.shape[0]:
            with torch.cuda.device(self.device):
                self.lib.fr_l2norm_rows_f32(self._lib.ptr(Q), self._lib.ptr(out), Q.shape[0], Q.shape[1],
                                            self._lib.stream_ptr())
        return out

    def scan(self, Q, counts=None, seg_len=0):
        """Q: unit query rows f32 [n,512]; returns global rows (shard row + row_lo) or -1."""
        return self.matcher.match_device(Q, renormalise=False, row_offset=self.row_lo, counts=counts, seg_len=seg_len)

    def pack_queries(self, Qn, q_max):
        send = torch.empty((q_max + 1, Qn.shape[1]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_exchange_pack_queries(self._lib.ptr(Qn), Qn.shape[0], q_max, Qn.shape[1], self._lib.ptr(send),
                                              self._lib.stream_ptr())
        return send

    def gathered_counts(self, allq, world, q_max):
        counts = torch.empty(world, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_exchange_counts(self._lib.ptr(allq), world, q_max, allq.shape[1], self._lib.ptr(counts),
                                        self._lib.stream_ptr())
        return counts

    def pack(self, idx, score):
        n = score.shape[0]
        cand = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_match_pack_candidates(self._lib.ptr(idx), self._lib.ptr(score), n, self._lib.ptr(cand),
                                              self._lib.stream_ptr())
        return cand

    def reduce(self, allp, world, n, q0, F):
        idx = torch.empty(F, dtype=torch.int64, device=self.device)
        score = torch.empty(F, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self.lib.fr_match_reduce_shards(self._lib.ptr(allp), world, n, q0, F, self._lib.ptr(idx),
                                            self._lib.ptr(score), self._lib.stream_ptr())
        return idx, score


class ShardedGalleryMatcher:
    def __init__(self, ops, q_max, dim=512, group=None, force_exchange=False):
        """``ops``: renormalise / pack_queries / gathered_counts / scan / pack / reduce (``HipOps`` in the product).
        ``force_exchange``: run both collectives even with one rank (rehearses the RCCL path on a 1-GPU box)."""
        self.ops, self.q_max, self.dim, self.group = ops, q_max, dim, group
        self.force_exchange = force_exchange
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.timing = None          # a list: every exchange appends its five HIP events (bench.py's exchange_ms; diagnostics)

    def _stamp(self, marks, dev):
        if marks is not None and dev.type == "cuda":
            e = torch.cuda.Event(enable_timing=True)
            e.record(torch.cuda.current_stream(dev))
            marks.append(e)

    def exchange_ms(self):
        """Median milliseconds of the exchange's four parts over the matches timed since ``timing = []`` was set (the caller
        synchronises first): gather_q = pack + the all-gather of the query rows (includes waiting for the slowest rank's
        embedder), scan = count row + this rank's shard scan, gather_c = pack + the all-gather of the candidates,
        reduce.  None when nothing was timed."""
        rows = [[m[k].elapsed_time(m[k + 1]) for k in range(4)] for m in (self.timing or []) if len(m) == 5]
        if not rows:
            return None
        med = torch.tensor(rows, dtype=torch.float64).median(dim=0).values.tolist()
        return {k: round(v, 4) for k, v in zip(("gather_q", "scan", "gather_c", "reduce"), med)} | {"samples": len(rows)}

    def match(self, Q):
        """Q f32 [F_local,dim] ``normed_embedding`` rows on this rank's device (F_local <= q_max).
        Returns (idx i64[F_local] global rows, score f32[F_local]) for the local queries."""
        F = Q.shape[0]
        assert F <= self.q_max, "more local queries than q_max"
        Qn = self.ops.renormalise(Q.to(torch.float32).contiguous())
        if self.world == 1 and not self.force_exchange:
            return self.ops.scan(Qn)
        dev = Q.device
        seg = self.q_max + 1
        marks = [] if self.timing is not None else None
        self._stamp(marks, dev)
        # (1) gather queries; the count rides in an extra row so it stays ONE collective
        send = self.ops.pack_queries(Qn, self.q_max)
        allq = torch.empty((self.world * seg, self.dim), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(allq, send, group=self.group)      # concatenated along dim 0
        self._stamp(marks, dev)
        counts = self.ops.gathered_counts(allq, self.world, self.q_max)   # gathered face counts, on the device
        # (2) scan the local shard for every gathered query slot IN PLACE: a rank's segment is its q_max query slots +
        # the count row, which is slot q_max >= count, i.e. padding like every slot past the count (skipped by the scan)
        idx, score = self.ops.scan(allq, counts=counts, seg_len=seg)
        self._stamp(marks, dev)
        # (3) gather the per-shard candidates and reduce those of the local queries
        n = score.shape[0]
        pair = self.ops.pack(idx, score)
        allp = torch.empty((self.world * n, 3), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allp, pair, group=self.group)
        self._stamp(marks, dev)
        out = self.ops.reduce(allp, self.world, n, self.rank * seg, F)
        self._stamp(marks, dev)
        if marks:
            self.timing.append(marks)
        return out
