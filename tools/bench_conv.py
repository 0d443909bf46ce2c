"""Dev tool: time ONE conv layer shape (default: IResNet stage-3 body conv, 256 faces) back to back."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import _lib

def run(B, H, W, Cin, Cout, k=3, stride=1, pad=1, iters=50, tag=""):
    lib = _lib.load()
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = (torch.randn((B, H, W, Cin), device="cuda")).half()
    w = (torch.randn((Cout, k * k * Cin), device="cuda") * 0.02).half()
    y = torch.empty((B, Ho, Wo, Cout), dtype=torch.float16, device="cuda")
    bias = torch.zeros(Cout, device="cuda")
    a = _lib.ConvArgs(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), _lib.ptr(bias), None, None, None,
                      B, H, W, Cin, Cout, k, k, stride, pad, Ho, Wo, 0, 1)
    s = _lib.stream_ptr()
    for _ in range(5):
        lib.fr_conv_nhwc_f16(ctypes.byref(a), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.fr_conv_nhwc_f16(ctypes.byref(a), s)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    print(f"{tag} B={B} {H}x{W} {Cin}->{Cout} k{k}s{stride}: {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tag = os.environ.get("FR_CONV_DBG", "0") + "/" + os.environ.get("FR_CONV_KERNEL", "2")
    run(B, 14, 14, 256, 256, tag=tag)
    run(B, 28, 28, 128, 128, tag=tag)
    run(B, 56, 56, 64, 64, tag=tag)
    run(B, 7, 7, 512, 512, tag=tag)
    if os.environ.get("EXTRA"):
        run(B, 14, 14, 256, 256, k=1, pad=0, tag=tag + " 1x1")
        run(B, 14, 14, 512, 256, k=3, tag=tag + " cin512")
