"""Dev tool: which statement of the exchange path blocks the host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
import torch, torch.distributed as dist
from facerecognition_infrenceengine_amd.distributed import reduce_candidates
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
A = torch.randn((8192, 8192), device=dev)
Q = torch.randn((256, 512), device=dev)
q_max, dim, world, rank, F = 256, 512, 1, 0, 256
def body(log):
    t = time.perf_counter()
    def lap(name):
        nonlocal t
        n = time.perf_counter(); log.append((name, (n - t) * 1e3)); t = n
    send = torch.zeros((q_max + 1, dim), dtype=torch.float32, device=dev); lap("zeros")
    send[:F] = Q; lap("send[:F]=Q")
    send[q_max, 0] = float(F); lap("send[q_max,0]=F")
    allq = torch.empty((world * (q_max + 1), dim), dtype=torch.float32, device=dev); lap("empty")
    dist.all_gather_into_tensor(allq, send); lap("all_gather 1")
    allq = allq.view(world, q_max + 1, dim)
    flat = allq[:, :q_max].reshape(world * q_max, dim); lap("reshape")
    score = flat[:, 0].contiguous(); idx = torch.arange(flat.shape[0], device=dev); lap("fake scan")
    n = score.shape[0]
    pair = torch.empty((n, 3), dtype=torch.int32, device=dev)
    pair[:, 0] = score.contiguous().view(torch.int32); lap("pair0")
    pair[:, 1:] = idx.contiguous().view(torch.int32).view(n, 2); lap("pair1")
    allp = torch.empty((world * n, 3), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(allp, pair); lap("all_gather 2")
    mine = allp.view(world, n, 3)[:, rank * q_max:rank * q_max + F]
    sc = torch.empty((world, F), dtype=torch.int32, device=dev).copy_(mine[..., 0]).view(torch.float32); lap("sc")
    ix = torch.empty((world, F, 2), dtype=torch.int32, device=dev).copy_(mine[..., 1:]).view(torch.int64).reshape(world, F); lap("ix")
    r = reduce_candidates(sc, ix); lap("reduce")
body([]); torch.cuda.synchronize()
for _ in range(14):
    A @ A
log = []
body(log)
torch.cuda.synchronize()
for name, ms in log:
    print(f"{name:18s} {ms:8.3f} ms")
dist.destroy_process_group()
