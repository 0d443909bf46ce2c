"""Dev tool: embed net alone (r100, 256 faces): ms per forward, f16 and fp8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from facerecognition_infrenceengine_amd import weights
from facerecognition_infrenceengine_amd.iresnet import IResNetHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
net = IResNetHIP(weights.synth_iresnet_state("r100"), "r100", "cuda:0")
x = (torch.rand((B, 112, 112, 8), device="cuda") * 2 - 1).half(); x[..., 3:] = 0
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16", "fp8"]
for mode in MODES:
    if mode == "fp8":
        net.enable_fp8(x[:32].contiguous())
    net.use_stage28 = mode != "f16_layers28"       # A/B: the 28x28 run layer by layer
    for _ in range(3): net.forward(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): net.forward(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{mode}: {ms:.3f} ms / {B} faces = {B/ms*1e3:.0f} faces/s, {net.flops_per_face*B/ms/1e9:.0f} TFLOP/s", flush=True)
    net.profile = []
    net.forward(x); torch.cuda.synchronize()
    per = {}
    for v, fl, a, b in net.profile:
        d = per.setdefault(v, [0, 0.0, 0.0]); d[0] += 1; d[1] += fl; d[2] += a.elapsed_time(b)
    net.profile = None
    for v, (n, fl, t) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        print(f"   {v:62s} x{n:3d} {t:7.3f} ms  {fl/t/1e9:7.0f} TFLOP/s")
    if len(sys.argv) > 3 and sys.argv[3] == "detail":          # every launch in network order
        net.profile = []
        net.forward(x); torch.cuda.synchronize()
        for v, fl, a, b in net.profile:
            t = a.elapsed_time(b)
            print(f"      {v[:58]:58s} {t * 1e3:8.1f} us {fl / 1e9:8.1f} GFLOP {fl / t / 1e9:7.0f} TFLOP/s")
        net.profile = None
