"""What the box gives a vendor-library GEMM and a plain copy ("confirm the peaks with a measured copy /
GEMM on the box").  torch.matmul (hipBLASLt / rocBLAS) in f16 at a large square shape and at the GEMM shape of the
dominant conv (M = 50 176 pixels, N = 256 couts, K = 2 304), fp8 where torch offers it, and a device-to-device copy."""
import time
import torch

dev = "cuda:0"


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(8192, 8192, 8192), (16384, 16384, 8192), (50176, 256, 2304), (50176, 512, 2304), (200704, 128, 1152)]:
    a = torch.randn((M, K), generator=g, device=dev, dtype=torch.float16)
    b = torch.randn((N, K), generator=g, device=dev, dtype=torch.float16)
    t = bench(lambda: torch.matmul(a, b.t()))
    print(f"f16 GEMM {M}x{N}x{K}: {t * 1e6:9.1f} us  {2 * M * N * K / t / 1e12:8.1f} TFLOP/s", flush=True)
    del a, b
x = torch.empty(1 << 30, dtype=torch.float32, device=dev)          # 4 GiB
y = torch.empty_like(x)
t = bench(lambda: y.copy_(x), n=10)
print(f"d2d copy 4 GiB: {t * 1e3:.2f} ms  {2 * x.numel() * 4 / t / 1e12:.2f} TB/s (read + write)")
t = bench(lambda: x.add_(1.0), n=10)
print(f"in-place add 4 GiB: {t * 1e3:.2f} ms  {2 * x.numel() * 4 / t / 1e12:.2f} TB/s (read + write)")
