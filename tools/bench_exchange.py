"""Dev tool: cost of the sharded-match exchange (RCCL, one rank) alone and beside a busy stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
from facerecognition_infrenceengine_amd import GalleryMatcher
from facerecognition_infrenceengine_amd.distributed import ShardedGalleryMatcher
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
G = torch.randn((10000, 512), device=dev)
gm = GalleryMatcher(dev); gm.set_rows(range(10000), G)
Q = torch.randn((256, 512), device=dev)
for force in (False, True):
    sh = ShardedGalleryMatcher(lambda q: gm.match_device(q, renormalise=True), 256, force_exchange=force)
    for _ in range(5):
        sh.match(Q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        sh.match(Q)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"force_exchange={force}: host enqueue {1e3*(t1-t0)/50:.3f} ms/call, total {1e3*(t2-t0)/50:.3f} ms/call")
# does a collective stall an unrelated busy stream?
A = torch.randn((8192, 8192), device=dev)
sb = torch.cuda.Stream()
def busy(n):
    with torch.cuda.stream(sb):
        for _ in range(n):
            A @ A
busy(2); torch.cuda.synchronize()
for force in (False, True):
    sh = ShardedGalleryMatcher(lambda q: gm.match_device(q, renormalise=True), 256, force_exchange=force)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    busy(20)
    for _ in range(20):
        sh.match(Q)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"busy stream + 20 matches, force_exchange={force}: {1e3*(t1-t0):.2f} ms")
for force in (False, True):
    sh = ShardedGalleryMatcher(lambda q: gm.match_device(q, renormalise=True), 256, force_exchange=force)
    sh.match(Q); torch.cuda.synchronize()
    for _ in range(14):
        A @ A
    t0 = time.perf_counter(); r = sh.match(Q); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"match behind 100 ms of queued work, force_exchange={force}: host returned after {1e3*(t1-t0):.2f} ms (drained {1e3*(t2-t0):.2f})")
# is the collective call itself host-blocking?  queue ~100 ms of work on the current stream, then time the calls
x = torch.randn((256, 512), device=dev); out = torch.empty((256, 512), device=dev)
for name, fn in (("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out, x)),
                 ("all_reduce", lambda: dist.all_reduce(x)),
                 ("copy_", lambda: out.copy_(x))):
    torch.cuda.synchronize()
    for _ in range(14):
        A @ A
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host returned after {1e3*(t1-t0):.2f} ms; stream drained after {1e3*(t2-t0):.2f} ms")
dist.destroy_process_group()
