"""Dev tool: do two streams still run concurrently once RCCL collectives are issued on one of them?"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, torch.distributed as dist
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
a = torch.randn((32, 1 << 20), device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn((256, 512), device=dev); out = torch.empty((256, 512), device=dev)

def run(two, coll, n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        for i in range(n):
            torch.cumsum(a, dim=1)
            if coll and i % 10 == 5:
                dist.all_gather_into_tensor(out, x)
    if two:
        with torch.cuda.stream(s2):
            for i in range(n):
                torch.cumsum(a, dim=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

run(True, False, 10)
print("before init: one stream %.1f ms, two streams %.1f ms" % (run(False, False), run(True, False)))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("after init : one stream %.1f ms, two streams %.1f ms" % (run(False, False), run(True, False)))
print("with collectives on s1: one stream %.1f ms, two streams %.1f ms" % (run(False, True), run(True, True)))
print("afterwards, no collectives: one stream %.1f ms, two streams %.1f ms" % (run(False, False), run(True, False)))
dist.destroy_process_group()
