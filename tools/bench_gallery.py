"""Dev tool: gallery scan throughput (HBM roofline: N*512*4 bytes per 32-query group per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd.gallery import GalleryMatcher
m = GalleryMatcher("cuda:0")
for N in (10_000, 1_000_000):
    G = torch.randn((N, 512), device="cuda"); G /= G.norm(dim=1, keepdim=True)
    m.set_rows(range(N), G, normalise=False)
    for F in (1, 32, 256):
        Q = G[torch.randint(0, N, (F,), device="cuda")] + 0.02 * torch.randn((F, 512), device="cuda")
        for _ in range(3): m.match_device(Q)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20 if N < 100000 else 5
        e0.record()
        for _ in range(it): m.match_device(Q)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / it
        groups = (F + 31) // 32
        print(f"N={N} F={F}: {ms*1e3:9.1f} us  unique {N*2048/ms/1e6:8.1f} GB/s  streamed(x{groups} groups) {groups*N*2048/ms/1e6:8.1f} GB/s  {2*F*N*512/ms/1e9:7.2f} TFLOP/s")
m16 = GalleryMatcher("cuda:0", f16_scan=True)
N = 1_000_000
G = torch.randn((N, 512), device="cuda"); G /= G.norm(dim=1, keepdim=True)
m16.set_rows(range(N), G, normalise=False)
for F in (32, 256, 2048):
    Q = G[torch.randint(0, N, (F,), device="cuda")] + 0.02 * torch.randn((F, 512), device="cuda")
    for _ in range(2): m16.match_device(Q)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): m16.match_device(Q)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    passes = (F + 127) // 128
    print(f"f16 N={N} F={F}: {ms*1e3:9.1f} us  streamed(x{passes} passes of 1.02 GB) {passes*N*1024/ms/1e6:8.1f} GB/s  {2*F*N*512/ms/1e9:7.2f} TFLOP/s")
