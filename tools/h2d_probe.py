import torch, time
x = torch.empty(398*1024*1024, dtype=torch.uint8).pin_memory()
d = torch.empty_like(x, device="cuda")
s = torch.cuda.Stream()
for _ in range(2):
    with torch.cuda.stream(s): d.copy_(x, non_blocking=True)
s.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s): d.copy_(x, non_blocking=True)
s.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"pinned H2D 398 MiB: {dt*1e3:.2f} ms = {x.numel()/dt/1e9:.1f} GB/s")
