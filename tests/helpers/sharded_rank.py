"""One rank of the 2-process sharded-match test (launched by tests/test_gpu_multiproc.py through
torch.distributed.run).  Every rank uses cuda:0 and the gloo backend (two ranks cannot share one device under
RCCL); what runs on the GPU is exactly the product: HipOps = libfrhip.so scan with row_offset, pack and reduce
kernels.  Exits non-zero when its ids differ from the unsharded oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    scan = sys.argv[1] if len(sys.argv) > 1 else "f32"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from facerecognition_infrenceengine_amd.distributed import HipOps, ShardedGalleryMatcher, shard_rows
    from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
    from oracle import match as omatch
    rng = np.random.default_rng(7)                      # same data on every rank
    N, q_max = 20011, 24
    G = rng.standard_normal((N, 512)).astype(np.float32); G /= np.linalg.norm(G, axis=1, keepdims=True)
    fs = [17, 0, 24, 5][:world] if world > 2 else [17, 9]
    Qs = [rng.standard_normal((f, 512)).astype(np.float32) for f in fs]
    q0 = Qs[0][0] / np.linalg.norm(Qs[0][0])
    G[11] = q0; G[N - 2] = q0                            # duplicate rows in different shards: lowest row wins
    lo, hi = shard_rows(N, world, rank)
    gm = GalleryMatcher("cuda:0", scan=scan)
    gm.set_rows(range(lo, hi), G[lo:hi], normalise=False)
    m = ShardedGalleryMatcher(HipOps(gm, lo), q_max)
    ok = True
    for step in range(3):                                # a few steps: buffers are re-used across collectives
        idx, score = m.match(torch.from_numpy(Qs[rank]).cuda())
        idx, score = idx.cpu().numpy(), score.cpu().numpy()
        if fs[rank]:
            Qn = np.stack([omatch.renormalise(q) for q in Qs[rank]])
            oi, os_ = omatch.match_rows_fast(Qn, G)
            ok &= bool(np.array_equal(idx, oi)) and bool(np.allclose(score, os_, atol=3e-6))
            if rank == 0:
                ok &= int(idx[0]) == 11
        else:
            ok &= len(idx) == 0
    flag = torch.tensor([0 if ok else 1])
    dist.all_reduce(flag)
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {'ok' if ok else 'MISMATCH'}", flush=True)
    sys.exit(int(flag.item() != 0))


if __name__ == "__main__":
    main()
